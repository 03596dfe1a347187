#!/bin/bash
# generic A/B: parity tests (PYTEST_ARGS), then the headline bench without extras under each env setting of AB_ENVS
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest ${PYTEST_ARGS:-tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_parity_deep.py} -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_ab.log 2>&1
echo "exit $? : tests"; grep -v "^/opt\|^$" gpurun_out/pytest_ab.log | tail -6
for e in ${AB_ENVS:-X=1}; do
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/bench_ab_$e.log 2>&1
  echo "bench $e exit $?"
  tail -1 gpurun_out/bench_ab_$e.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('  value %.1f  ms/step %.1f  self %.2f us  cross %.2f us  decode %.1f ms' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['cross_attn']['avg_launch_us'], r['decode_ms_single_chain']))
" 2>&1 | tail -2
done
