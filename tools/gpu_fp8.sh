#!/bin/bash
# fp8 K/V cache: parity tests, then the bench line with e4m3 caches for 2/3/4 waves per decode-attention workgroup
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_deep.py tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -s -k "fp8 or base" > gpurun_out/pytest_fp8.log 2>&1
echo "exit $? : fp8 tests"; grep -v "^/opt\|^$" gpurun_out/pytest_fp8.log | tail -25
for w in 2 3 4; do
  MT3_DEC_ATTN_FP8_WAVES=$w timeout 300 python bench.py --kv-dtype fp8_e4m3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_fp8_w$w.log 2>&1
  echo "waves $w exit $?"
  tail -1 gpurun_out/bench_fp8_w$w.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('  value %.1f  ms/step %.1f  self-attn %.2f us %.0f GB/s  cross %.2f us %.0f GB/s  decode %.1f ms' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['achieved'], r['cross_attn']['avg_launch_us'], r['cross_attn']['achieved'], r['decode_ms_single_chain']))
" 2>&1 | tail -2
done
