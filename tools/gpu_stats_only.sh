#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command alone (no PMC passes, no second bench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r2 -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$R/gpurun_out/bench_prof.log" 2>&1
echo "exit $? : rocprof bench"
cd "$R"
f=$(find gpurun_out/prof -name "r2_kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-200
python tools/trace_digest.py gpurun_out/prof > gpurun_out/trace_digest.txt 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/prof -name "*.db" -delete
tail -1 gpurun_out/bench_prof.log | cut -c1-300
