#!/bin/bash
# One parameterised GPU-session driver (run through `gpurun -- bash tools/gpurun.sh STAGE...`); every stage writes under
# gpurun_out/ with the round tag $TAG (default r6).  Stages:
#   test [pytest args]   pytest -m gpu (TESTS = files / -k expression, default the whole suite)
#   smoke                __graft_entry__.smoke()
#   bench                the driver's bench command (BENCH_ARGS, default --gpus 1 --steps 20 --warmup 5) + digest
#   prof                 rocprofv3 --kernel-trace --stats of the f32 bench command + trace digest
#   tol                  tools/note_tolerance.py (TOL_ARGS)
#   run                  RUN_CMD verbatim (one-off probes)
# Environment: TAG, TESTS, BENCH_ARGS, TOL_ARGS, RUN_CMD.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
TAG="${TAG:-r6}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for stage in "$@"; do
  t0=$(date +%s)
  case "$stage" in
    test)
      timeout 1800 python -m pytest ${TESTS:-tests} -m gpu -q -x > "gpurun_out/${TAG}_pytest.log" 2>&1
      echo "exit $? : pytest -m gpu ${TESTS:-tests} after $(( $(date +%s) - t0 )) s"; tail -6 "gpurun_out/${TAG}_pytest.log" ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      timeout 1500 python bench.py ${BENCH_ARGS:---gpus 1 --steps 20 --warmup 5} > "gpurun_out/${TAG}_bench.log" 2>&1
      echo "exit $? : bench after $(( $(date +%s) - t0 )) s"
      grep '^{' "gpurun_out/${TAG}_bench.log" | tail -1 > "gpurun_out/${TAG}_bench.json"
      python tools/bench_digest.py < "gpurun_out/${TAG}_bench.json" 2>&1 | tail -40 ;;
    prof)
      rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o "${TAG}_f32" -- \
          python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$R/gpurun_out/${TAG}_bench_prof.log" 2>&1 )
      echo "exit $? : rocprof bench after $(( $(date +%s) - t0 )) s"
      f=$(find gpurun_out/prof -name "${TAG}_f32_kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp "$f" "gpurun_out/${TAG}_f32_kernel_stats.csv" && head -8 "$f" | cut -c1-180
      mkdir -p "gpurun_out/prof_${TAG}_f32" && find gpurun_out/prof -name "${TAG}_f32_kernel_trace.csv" -exec cp {} "gpurun_out/prof_${TAG}_f32/" \;
      python tools/trace_digest.py "gpurun_out/prof_${TAG}_f32" > "gpurun_out/${TAG}_f32_trace_digest.txt" 2>&1; tail -12 "gpurun_out/${TAG}_f32_trace_digest.txt"
      grep '^{' "gpurun_out/${TAG}_bench_prof.log" | tail -1 > "gpurun_out/${TAG}_f32_bench_under_rocprof.json"
      find gpurun_out/prof "gpurun_out/prof_${TAG}_f32" -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*.db" -delete ;;
    tol)
      timeout 900 python tools/note_tolerance.py ${TOL_ARGS} > "gpurun_out/${TAG}_note_tolerance_$(echo ${TOL_ARGS} | tr -c 'a-zA-Z0-9\n' '_').log" 2>&1
      echo "exit $? : note_tolerance ${TOL_ARGS} after $(( $(date +%s) - t0 )) s"
      grep -h '^NOTE_TOLERANCE' gpurun_out/${TAG}_note_tolerance_*.log | tail -1 | cut -c1-3000 ;;
    run)
      bash -c "$RUN_CMD"; echo "exit $? : run after $(( $(date +%s) - t0 )) s" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
