#!/bin/bash
# the GPU tests tools/gpu_quick.sh leaves out, then the bench line with the driver's flags (extras on, CPU leg off)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity_deep.py tests/test_gpu_mx8.py tests/test_external_fixtures.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/pytest_rest.log 2>&1; echo "exit $? : pytest"
tail -3 gpurun_out/pytest_rest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final.log 2>&1; echo "exit $? : bench"
tail -1 gpurun_out/bench_final.log | cut -c1-400
