#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for b in 64 128 512 1024; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b > gpurun_out/bench_b$b.log 2>&1
  tail -1 gpurun_out/bench_b$b.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('batch $b value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1),'| self us',round(r['avg_launch_us'],2),'GB/s',round(r['achieved']),'| cross us',round(r['cross_attn']['avg_launch_us'],2))
"
done
