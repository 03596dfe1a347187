#!/bin/bash
# round 5, GPU call B: tests + smoke, sleeping vs spinning waits (CPU seconds), rocprofv3 stats of a refilled corpus pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_transcribe.py tests/test_gpu_retire.py tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r5_b_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > gpurun_out/r5_b_smoke.log
L=gpurun_out/r5_b_waits.jsonl; : > $L
for spin in "" "--spin-waits"; do
  timeout 200 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --decode-probe $spin 2>&1 | grep '^{' >> $L
  timeout 300 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe $spin 2>&1 | grep '^{' >> $L
done
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r5" -o refill_f32 -- python "$R/tools/eos_corpus.py" --slots 1250 --segments 3750 --dtype float32 --mode refill > "$R/gpurun_out/r5_b_prof.log" 2>&1
echo "exit $? : rocprof refill"
cd "$R"
f=$(find gpurun_out/prof_r5 -name "refill_f32_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r5_refill_f32_kernel_stats.csv && head -20 "$f" | cut -c1-200
find gpurun_out/prof_r5 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r5 -name "*.db" -delete
cat gpurun_out/r5_b_tests.log gpurun_out/r5_b_smoke.log; tail -3 gpurun_out/r5_b_prof.log | cut -c1-600
