#!/bin/bash
# Round-3 GPU session driver (run through gpurun): STAGES is a space-separated subset of
#   test   pytest -m gpu (TESTS = extra pytest args, e.g. a -k filter)
#   pmc    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_attn.py (bf16 / f32 / fp8 decode attention +
#          frontend) -> gpurun_out/pmc/r3_pmc_summary.json (+ the two counter csv files)
#   prof   rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras` for the bf16
#          headline and for --dtype float32 -> gpurun_out/prof/{r3,r3_f32}_kernel_stats.csv + trace digests
#   bench  python bench.py $BENCH_ARGS -> gpurun_out/bench_r3.log
#   pmcenc tools/gpu_pmc_enc.sh: SQ / MFMA counter passes over the encoder -> gpurun_out/pmc_enc/summary.json
#   ab     python tools/ab_decode.py $AB_ARGS (decode-loop variants, one process) -> gpurun_out/ab_decode.log
# Everything lands under gpurun_out/; copy what should be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
STAGES="${STAGES:-test bench}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in $STAGES; do
  case $st in
    test)
      timeout ${TEST_TIMEOUT:-900} python -m pytest ${TESTS:-tests} -m gpu -x -q -s > gpurun_out/pytest_r3.log 2>&1
      echo "exit $? : pytest -m gpu $TESTS"; tail -5 gpurun_out/pytest_r3.log; grep -h "rel-L2\|token-exact\|agreement\|benched\|vs separate\|vs f32 oracle\|vs the r2" gpurun_out/pytest_r3.log | cut -c1-260 | tail -40
      ;;
    pmc)
      rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1
        echo "exit $? : pmc $c"
      done
      cd "$R"
      python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc r3 > gpurun_out/pmc/summary.log 2>&1; tail -12 gpurun_out/pmc/summary.log
      cp gpurun_out/pmc/r3_pmc_summary.json profiles/ 2>/dev/null   # a later "bench" stage of the same call reads it
      python - <<'PY'
import json
try:
    s = json.load(open("gpurun_out/pmc/r3_pmc_summary.json"))
    for k, v in s.items():
        if k.startswith("dec_attn_self"):
            print(k, {n: round(e["traffic_over_algorithmic"], 4) for n, e in v.items()})
        elif k.startswith(("dec_attn_cross", "logmel")):
            print(k, round(v["traffic_over_algorithmic"], 4))
except Exception as e:
    print("no pmc summary:", e)
PY
      find gpurun_out/pmc -name "*.db" -delete; find gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete
      ;;
    prof)
      [ -z "$PROF_VARIANTS" ] && rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
      for v in ${PROF_VARIANTS:-"r3:" "r3_f32:--dtype=float32"}; do
        name="${v%%:*}"; flags="${v#*:}"
        cd /tmp
        timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o $name -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras $flags > "$R/gpurun_out/bench_prof_$name.log" 2>&1
        echo "exit $? : rocprof bench $flags"
        cd "$R"
        f=$(find gpurun_out/prof -name "${name}_kernel_stats.csv" | head -1)
        [ -n "$f" ] && head -12 "$f" | cut -c1-200
        mkdir -p gpurun_out/prof_$name && find gpurun_out/prof -name "${name}_kernel_trace.csv" -exec cp {} gpurun_out/prof_$name/ \;
        python tools/trace_digest.py gpurun_out/prof_$name > gpurun_out/${name}_trace_digest.txt 2>&1
        rm -rf gpurun_out/prof_$name
        tail -1 "$R/gpurun_out/bench_prof_$name.log" | cut -c1-400
      done
      find gpurun_out/prof -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*.db" -delete
      ;;
    bench)
      timeout ${BENCH_TIMEOUT:-900} python bench.py $BENCH_ARGS > gpurun_out/bench_r3.log 2>&1
      echo "exit $? : bench $BENCH_ARGS"; tail -1 gpurun_out/bench_r3.log | python tools/bench_digest.py
      ;;
    ab)
      timeout ${AB_TIMEOUT:-600} python tools/ab_decode.py $AB_ARGS > gpurun_out/ab_decode.log 2>&1
      echo "exit $? : ab_decode $AB_ARGS"; grep -v "^/opt\|Warning" gpurun_out/ab_decode.log | tail -20
      ;;
    enc)
      timeout 300 python tools/ab_encoder.py > gpurun_out/ab_encoder.log 2>&1
      echo "exit $? : ab_encoder"; grep -v "^/opt\|Warning" gpurun_out/ab_encoder.log | tail -8
      rm -rf gpurun_out/prof_enc; mkdir -p gpurun_out/prof_enc
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_enc" -o enc -- python "$R/tools/ab_encoder.py" > "$R/gpurun_out/ab_encoder_prof.log" 2>&1
      echo "exit $? : rocprof ab_encoder"
      cd "$R"
      f=$(find gpurun_out/prof_enc -name "enc_kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp "$f" gpurun_out/r3_encoder_kernel_stats.csv && head -24 "$f" | cut -c1-170
      find gpurun_out/prof_enc -name "*kernel_trace.csv" -delete; find gpurun_out/prof_enc -name "*.db" -delete
      ;;
    pmcenc)
      bash tools/gpu_pmc_enc.sh > gpurun_out/pmc_enc.log 2>&1
      echo "exit $? : pmc_enc"; grep "mfma_util" gpurun_out/pmc_enc.log | cut -c1-230 | tail -16
      ;;
    fe)
      timeout 200 python tools/ab_frontend.py > gpurun_out/ab_frontend.log 2>&1
      echo "exit $? : ab_frontend"; grep -v "^/opt\|Warning" gpurun_out/ab_frontend.log | tail -8
      ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r3.log 2>&1
      echo "exit $? : smoke"; tail -1 gpurun_out/smoke_r3.log | cut -c1-200
      ;;
    corpus)
      timeout 600 python bench.py --corpus 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/bench_corpus_r3.log 2>&1
      echo "exit $? : bench --corpus 10000"; grep -h '^{"metric"' gpurun_out/bench_corpus_r3.log | tail -1 | python tools/bench_digest.py
      ;;
    split)
      timeout ${SPLIT_TIMEOUT:-240} python tools/ab_split.py > gpurun_out/ab_split.log 2>&1
      echo "exit $? : ab_split"; grep -v "^/opt\|Warning" gpurun_out/ab_split.log | tail -14
      ;;
    fepmc)
      bash tools/gpu_pmc_fe.sh > gpurun_out/pmc_fe.log 2>&1
      echo "exit $? : pmc_fe"; grep "exit\|_per_frame\|share\|over_busy\|per_wave_cycle\|kernel_us" gpurun_out/pmc_fe.log | cut -c1-200 | tail -30
      ;;
    *) echo "unknown stage $st";;
  esac
done
