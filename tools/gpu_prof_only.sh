#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r1 -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/bench_prof.log" 2>&1
echo "exit $? : rocprof"
cd "$R"
python tools/trace_digest.py gpurun_out/prof > gpurun_out/trace_digest.txt 2>&1
grep dec_attn gpurun_out/trace_digest.txt | cut -c1-300
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
