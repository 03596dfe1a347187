#!/bin/bash
# round 5, final GPU call: the whole GPU suite + smoke, rocprofv3 stats + trace digest of the f32 bench command, the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5_final_pytest.log 2>&1
echo "exit $? : pytest -m gpu after $(( $(date +%s) - t0 )) s"; tail -4 gpurun_out/r5_final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r5_f32 -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$R/gpurun_out/r5_bench_prof.log" 2>&1
echo "exit $? : rocprof bench"
cd "$R"
f=$(find gpurun_out/prof -name "r5_f32_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r5_f32_kernel_stats.csv && head -8 "$f" | cut -c1-180
mkdir -p gpurun_out/prof_r5_f32 && find gpurun_out/prof -name "r5_f32_kernel_trace.csv" -exec cp {} gpurun_out/prof_r5_f32/ \;
python tools/trace_digest.py gpurun_out/prof_r5_f32 > gpurun_out/r5_f32_trace_digest.txt 2>&1; tail -12 gpurun_out/r5_f32_trace_digest.txt
grep '^{' gpurun_out/r5_bench_prof.log | tail -1 > gpurun_out/r5_f32_bench_under_rocprof.json
find gpurun_out/prof gpurun_out/prof_r5_f32 -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*.db" -delete
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_like.log 2>&1
echo "exit $? : bench after $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r5_bench_driver_like.log | tail -1 > gpurun_out/r5_bench_driver_like.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_driver_like.json"))
print("value", d["value"], d["dtype"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("parity", json.dumps(d["cpu_baseline"].get("parity"))[:400])
x = d["extra"]
print("eos corpus", x["eos_schedule_corpus"]["f32"]["value"], x["eos_schedule_corpus"]["f32"]["batch_synchronous"]["value"], "single file", x["single_file"].get("speedup"), x["single_file"].get("notes_identical"))
print("errors:", {k: v.get("error") for k, v in x.items() if isinstance(v, dict) and v.get("error")})
PY
