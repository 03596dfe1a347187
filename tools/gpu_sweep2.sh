#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_kernels.py -k attention -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
for w in 2 3 4; do
  MT3_DEC_ATTN_WAVES=$w timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_w$w.log 2>&1
  tail -1 gpurun_out/bench_w$w.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('waves $w value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1),'| self us',round(r['avg_launch_us'],2),'GB/s',round(r['achieved']),'| cross us',round(r['cross_attn']['avg_launch_us'],2),'| 1-chain decode ms',round(r['decode_ms_single_chain'],1))
"
done
