#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for c in 1 2 4 8; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chains $c > gpurun_out/bench_c$c.log 2>&1
  tail -1 gpurun_out/bench_c$c.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('chains $c value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1),'| self us',round(r['avg_launch_us'],2),'GB/s',round(r['achieved']),'| 1-chain decode ms',round(r['decode_ms_single_chain'],1))
"
done
