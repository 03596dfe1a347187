#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 0 1 2; do
  MT3_KV_UNCACHED=$v timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_u$v.log 2>&1
  tail -1 gpurun_out/bench_u$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('uncached $v value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1),'| self us',round(r['avg_launch_us'],2),'GB/s',round(r['achieved']),'| cross us',round(r['cross_attn']['avg_launch_us'],2),'| noself',round(r['decode_ms_without_self_attn'],1))
" || tail -3 gpurun_out/bench_u$v.log
done
