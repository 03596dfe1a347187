#!/bin/bash
# Round-2 evidence: PMC traffic passes (decode attention + frontend), rocprofv3 kernel stats of the bench command,
# then the default bench line.  Everything lands under gpurun_out/; copy the summaries into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1
  echo "exit $? : pmc $c"
done
cd "$R"
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc r2 > gpurun_out/pmc/summary.log 2>&1; tail -30 gpurun_out/pmc/summary.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r2 -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$R/gpurun_out/bench_prof.log" 2>&1
echo "exit $? : rocprof bench"
cd "$R"
f=$(find gpurun_out/prof -name "r2_kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-220
python tools/trace_digest.py gpurun_out/prof > gpurun_out/trace_digest.txt 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/prof -name "*.db" -delete
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py > gpurun_out/bench_r2_final.log 2>&1
  echo "exit $? : bench"; tail -1 gpurun_out/bench_r2_final.log | cut -c1-7000
fi
