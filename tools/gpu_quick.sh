#!/bin/bash
# quick iteration: engine + gemm parity tests, then the full-size bench (no rocprof)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
echo "exit $? : bench"; tail -1 gpurun_out/bench_quick.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1),'| self us',round(r['avg_launch_us'],2),'GB/s',round(r['achieved']),'| cross us',round(r['cross_attn']['avg_launch_us'],2),'| 1-chain decode ms',round(r['decode_ms_single_chain'],1),'noself',round(r['decode_ms_without_self_attn'],1),'nocross',round(r['decode_ms_without_cross_attn'],1))
"
