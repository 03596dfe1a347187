#!/bin/bash
# Full-size bench + rocprofv3 kernel trace of the same command.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_full.log 2>&1
echo "exit $? : bench_full"; tail -1 gpurun_out/bench_full.log | cut -c1-2500
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log" 2>&1
echo "exit $? : rocprof"
cd "$GRAFT_REPO_ROOT"
grep -v "^[WEI]2026" gpurun_out/bench_prof.log | tail -1 | cut -c1-2500
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-250
# the raw per-dispatch trace is large: keep a compact per-step digest instead
python tools/trace_digest.py gpurun_out/prof > gpurun_out/trace_digest.txt 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/prof -name "*.db" -delete
