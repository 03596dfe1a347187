#!/bin/bash
# parity tests of the kernel / engine groups, then the full bench line (no CPU baseline), default vs an env switch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_try.log 2>&1
echo "exit $? : tests"; tail -8 gpurun_out/pytest_try.log
for e in "X=1" "${AB_ENV:-X=2}" "X=1" "${AB_ENV:-X=2}"; do
env $e timeout 300 python bench.py --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/try_bench.log 2>&1
echo "bench $e exit $?"
tail -1 gpurun_out/try_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  value %.1f  ms/step %.1f  roofline %s  1-chain decode %.1f ms' % (d['value'], d['ms_per_step'], {k: round(d['roofline'][k], 3) for k in ('achieved', 'frac')}, d['roofline']['decode_ms_single_chain']))
"
done
