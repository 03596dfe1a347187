#!/bin/bash
# What the driver runs at round end, in its own order: ONE pytest process over every -m gpu test, smoke(), then the
# bench line with the driver's flags.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/pytest_all_gpu.log 2>&1
echo "exit $? : pytest -m gpu (one process)"; grep -v "^/opt\|^$" gpurun_out/pytest_all_gpu.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "exit $? : smoke"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_like.log 2>&1
echo "exit $? : bench"; tail -1 gpurun_out/bench_driver_like.log | cut -c1-1200
