#!/bin/bash
# engine tests (incl. beam-1 parity) + full-size bench with both token-selection rules
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_engine.log 2>&1
echo "exit $? : engine tests"; tail -15 gpurun_out/pytest_engine.log
for d in greedy beam1; do
  timeout 300 python bench.py --decoding $d --no-cpu-baseline > gpurun_out/bench_$d.log 2>&1
  echo "exit $? : bench $d"; tail -1 gpurun_out/bench_$d.log | cut -c1-400
done
