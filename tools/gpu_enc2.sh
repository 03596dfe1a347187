#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
PROBE_SPLIT=1 timeout 300 python tools/enc_probe.py 2>&1 | grep -v "^/opt" | tail -8
MT3_NO_GLDS=1 timeout 300 python tools/enc_probe.py 2>&1 | grep -v "^/opt" | tail -3
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/pytest_enc2.log 2>&1
echo "exit $? : tests"; grep -v "^/opt\|^$" gpurun_out/pytest_enc2.log | tail -8
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_enc" -o enc -- python "$R/tools/enc_probe.py" > "$R/gpurun_out/prof_enc.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-200
find gpurun_out/prof_enc -name "*kernel_trace.csv" -size +4M -delete; find gpurun_out/prof_enc -name "*.db" -delete
