#!/bin/bash
# Round-4 GPU session driver (run through gpurun): STAGES is a space-separated subset of
#   test   pytest -m gpu (TESTS = extra pytest args, e.g. a file or a -k filter)
#   bench  python bench.py $BENCH_ARGS -> gpurun_out/bench_r4.log (+ the JSON line as gpurun_out/bench_r4.json)
#   prof   rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras` for the f32
#          headline and for --dtype bfloat16 -> gpurun_out/prof/{r4,r4_bf16}_kernel_stats.csv + trace digests
#   pmc    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_attn.py (bf16 / f32 / fp8 decode attention +
#          frontend) -> gpurun_out/pmc/r4_pmc_summary.json (+ the two counter csv files)
#   pmcenc tools/gpu_pmc_enc.sh: SQ / MFMA counter passes over the encoder -> gpurun_out/pmc_enc/summary.json
#   ab     python tools/ab_r4.py $AB_ARGS -> gpurun_out/ab_r4.log
#   eosprof rocprofv3 --kernel-trace --stats of tools/eos_profile.py (the decode loop under the synthetic EOS schedule)
#   smoke  __graft_entry__.smoke()
#   corpus bench.py --corpus 10000 (BASELINE configs[3] at N = 1)
# Everything lands under gpurun_out/; copy what should be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
STAGES="${STAGES:-test bench}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in $STAGES; do
  case $st in
    test)
      timeout ${TEST_TIMEOUT:-1200} python -m pytest ${TESTS:-tests} -m gpu -x -q -s > gpurun_out/pytest_r4.log 2>&1
      echo "exit $? : pytest -m gpu $TESTS"; tail -5 gpurun_out/pytest_r4.log
      grep -h "Error\|assert \|FAILED" gpurun_out/pytest_r4.log | cut -c1-300 | tail -25
      ;;
    bench)
      timeout ${BENCH_TIMEOUT:-900} python bench.py $BENCH_ARGS > gpurun_out/bench_r4.log 2>&1
      echo "exit $? : bench $BENCH_ARGS"
      grep -h '^{"metric"' gpurun_out/bench_r4.log | tail -1 > gpurun_out/bench_r4${BENCH_TAG}.json
      python tools/bench_digest.py < gpurun_out/bench_r4${BENCH_TAG}.json
      grep -v "^/opt\|Warning\|^{" gpurun_out/bench_r4.log | tail -5
      ;;
    prof)
      [ -z "$PROF_VARIANTS" ] && rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
      for v in ${PROF_VARIANTS:-"r4:" "r4_bf16:--dtype=bfloat16"}; do
        name="${v%%:*}"; flags="${v#*:}"
        cd /tmp
        timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o $name -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras $flags > "$R/gpurun_out/bench_prof_$name.log" 2>&1
        echo "exit $? : rocprof bench $flags"
        cd "$R"
        f=$(find gpurun_out/prof -name "${name}_kernel_stats.csv" | head -1)
        [ -n "$f" ] && cp "$f" gpurun_out/${name}_kernel_stats.csv && head -14 "$f" | cut -c1-200
        mkdir -p gpurun_out/prof_$name && find gpurun_out/prof -name "${name}_kernel_trace.csv" -exec cp {} gpurun_out/prof_$name/ \;
        python tools/trace_digest.py gpurun_out/prof_$name > gpurun_out/${name}_trace_digest.txt 2>&1
        rm -rf gpurun_out/prof_$name
        grep -h '^{"metric"' "$R/gpurun_out/bench_prof_$name.log" | tail -1 > gpurun_out/${name}_bench_under_rocprof.json
        cut -c1-300 gpurun_out/${name}_bench_under_rocprof.json
      done
      find gpurun_out/prof -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*.db" -delete
      ;;
    pmc)
      rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1
        echo "exit $? : pmc $c"
      done
      cd "$R"
      python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc r4 > gpurun_out/pmc/summary.log 2>&1; tail -12 gpurun_out/pmc/summary.log
      cp gpurun_out/pmc/r4_pmc_summary.json profiles/ 2>/dev/null   # a later "bench" stage of the same call reads it
      find gpurun_out/pmc -name "*.db" -delete; find gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete
      ;;
    pmcenc)
      bash tools/gpu_pmc_enc.sh > gpurun_out/pmc_enc.log 2>&1
      echo "exit $? : pmc_enc"; grep "mfma_util" gpurun_out/pmc_enc.log | cut -c1-230 | tail -16
      ;;
    ab)
      timeout ${AB_TIMEOUT:-600} python tools/ab_r4.py $AB_ARGS > gpurun_out/ab_r4.log 2>&1
      echo "exit $? : ab_r4 $AB_ARGS"; grep -v "^/opt\|Warning" gpurun_out/ab_r4.log | tail -40
      ;;
    eosprof)
      rm -rf gpurun_out/prof_eos; mkdir -p gpurun_out/prof_eos
      for dt in ${EOS_DTYPES:-float32 bfloat16}; do
        cd /tmp
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_eos" -o eos_$dt -- python "$R/tools/eos_profile.py" $dt 2 > "$R/gpurun_out/eos_profile_$dt.log" 2>&1
        echo "exit $? : rocprof eos_profile $dt"; grep "eos-schedule" "$R/gpurun_out/eos_profile_$dt.log"
        cd "$R"
        f=$(find gpurun_out/prof_eos -name "eos_${dt}_kernel_stats.csv" | head -1)
        [ -n "$f" ] && cp "$f" gpurun_out/r4_eos_${dt}_kernel_stats.csv && head -16 "$f" | cut -c1-220
      done
      find gpurun_out/prof_eos -name "*kernel_trace.csv" -delete; find gpurun_out/prof_eos -name "*.db" -delete
      ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r4.log 2>&1
      echo "exit $? : smoke"; tail -1 gpurun_out/smoke_r4.log | cut -c1-200
      ;;
    corpus)
      timeout 600 python bench.py --corpus 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-extras $CORPUS_ARGS > gpurun_out/bench_corpus_r4.log 2>&1
      echo "exit $? : bench --corpus 10000"; grep -h '^{"metric"' gpurun_out/bench_corpus_r4.log | tail -1 | tee gpurun_out/bench_corpus_r4.json | python tools/bench_digest.py
      ;;
    *) echo "unknown stage $st";;
  esac
done
