#!/bin/bash
# MXFP8 encoder: parity tests, then the encoder alone (bf16 vs MXFP8 at B = 256) and under rocprofv3 --stats.
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_mx8.py -x -q -s -m gpu > gpurun_out/pytest_mx8.log 2>&1; echo "exit $? : pytest mx8"
tail -25 gpurun_out/pytest_mx8.log
timeout 120 python tools/enc_probe.py > gpurun_out/enc_bf16.log 2>&1; tail -2 gpurun_out/enc_bf16.log
PROBE_DENSE=fp8_e4m3 timeout 120 python tools/enc_probe.py > gpurun_out/enc_mx8.log 2>&1; tail -3 gpurun_out/enc_mx8.log
cd /tmp && export TMPDIR=/tmp
PROBE_DENSE=fp8_e4m3 timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_mx8 -o mx8 -- python $GRAFT_REPO_ROOT/tools/enc_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_mx8.log 2>&1
echo "exit $? : rocprofv3"
find $GRAFT_REPO_ROOT/gpurun_out/prof_mx8 -name "*kernel_stats.csv" | head -1 | xargs -I{} head -14 {}
