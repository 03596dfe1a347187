#!/bin/bash
# round 5, GPU call G: the whole GPU suite + smoke; rocprofv3 stats of a refilled pass at 256 slots; PMC passes on this round's sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_g_pytest.log 2>&1
echo "exit $? : pytest -m gpu after $(( $(date +%s) - t0 )) s"; tail -4 gpurun_out/r5_g_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r5b" -o refill256 -- python "$R/tools/eos_corpus.py" --slots 256 --segments 1280 --dtype float32 --mode refill > "$R/gpurun_out/r5_g_prof256.log" 2>&1
echo "exit $? : rocprof refill 256"
cd "$R"
f=$(find gpurun_out/prof_r5b -name "refill256_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r5_refill_f32_256_slots_kernel_stats.csv && head -12 "$f" | cut -c1-160
find gpurun_out/prof_r5b -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r5b -name "*.db" -delete
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1
  echo "exit $? : pmc $c"
done
cd "$R"
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc r5 > gpurun_out/pmc/summary.log 2>&1; tail -12 gpurun_out/pmc/summary.log
find gpurun_out/pmc -name "*.db" -delete; find gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete
