#!/bin/bash
# Round-2 GPU check: every -m gpu test (per file, separate processes), smoke(), the default bench line (with the
# f32 line, stage extras and the CPU baseline), and the configs[3] corpus mode at N = 1.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
for f in test_gpu_kernels.py test_gpu_engine.py test_gpu_parity_deep.py test_gpu_distributed.py test_external_fixtures.py; do
  timeout 900 python -m pytest tests/$f -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_$f.log 2>&1
  echo "exit $? : $f" >> gpurun_out/summary.log
  tail -4 gpurun_out/pytest_$f.log >> gpurun_out/summary.log
done
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "exit $? : smoke" >> gpurun_out/summary.log
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py ${BENCH_ARGS} > gpurun_out/bench_r2.log 2>&1
  echo "exit $? : bench" >> gpurun_out/summary.log
  tail -1 gpurun_out/bench_r2.log | cut -c1-6000 >> gpurun_out/summary.log
fi
if [ -n "$CORPUS" ]; then
  timeout 600 python bench.py --corpus $CORPUS --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/bench_corpus.log 2>&1
  echo "exit $? : corpus" >> gpurun_out/summary.log
  tail -1 gpurun_out/bench_corpus.log | cut -c1-3000 >> gpurun_out/summary.log
fi
cat gpurun_out/summary.log
