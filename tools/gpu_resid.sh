#!/bin/bash
# after a change to the encoder GEMM epilogues: GEMM / encoder parity tests, then the encoder alone, bf16 and MXFP8
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_mx8.py tests/test_gpu_kernels.py -x -q -m gpu -k "mx8 or gemm or glds or residual" > gpurun_out/pytest_resid.log 2>&1; echo "exit $? : pytest"
tail -4 gpurun_out/pytest_resid.log
for i in 1 2; do
timeout 120 python tools/enc_probe.py 2>&1 | tail -1
PROBE_DENSE=fp8_e4m3 timeout 120 python tools/enc_probe.py 2>&1 | tail -1
done
