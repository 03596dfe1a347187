#!/bin/bash
# One parameterised GPU-session driver (run through `gpurun -- bash tools/gpurun.sh STAGE...`); every stage writes under
# gpurun_out/ with the round tag $TAG (default r6).  Stages:
#   test [pytest args]   pytest -m gpu (TESTS = files / -k expression, default the whole suite)
#   smoke                __graft_entry__.smoke()
#   bench                the driver's bench command (BENCH_ARGS, default --gpus 1 --steps 20 --warmup 5) + digest
#   prof                 rocprofv3 --kernel-trace --stats of the f32 bench command + trace digest
#   tol                  tools/note_tolerance.py (TOL_ARGS)
#   pmcdec               rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_decode.py -> per-kernel HBM bytes (PMC_STEPS)
#   pmcattn              the same two passes over tools/pmc_attn.py -> <TAG>_pmc_summary.json (roofline.traffic)
#   pmcenc               SQ / MFMA counter passes over one 256-segment encoder pass (tools/pmc_encoder.py) -> <TAG>_pmc_encoder_summary.json
#   pmcfe                SQ counter passes over the log-mel kernel (tools/pmc_frontend.py) -> <TAG>_pmc_frontend_counters.json
#   micro                tools/micro/<MICRO> (default cu_split_groups)
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
    pmcdec)
      # FETCH_SIZE / WRITE_SIZE passes over the PRODUCT decode schedule, reduced per kernel name (VERDICT r5 #3a)
      rm -rf gpurun_out/pmcdec; mkdir -p gpurun_out/pmcdec
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcdec" -o $c -- \
            python "$R/tools/pmc_decode.py" > "$R/gpurun_out/pmcdec/$c.log" 2>&1 )
        echo "exit $? : pmc decode $c after $(( $(date +%s) - t0 )) s"
      done
      python tools/pmc_decode_summary.py gpurun_out/pmcdec "gpurun_out/${TAG}_pmc_decode_per_kernel.json" ${PMC_STEPS:-32} 2>&1 | tail -30
      find gpurun_out/pmcdec -name "*.db" -delete; find gpurun_out/pmcdec -name "*.csv" -size +2M -delete ;;
    pmcattn)
      # the standalone decode-attention / frontend passes bench.py's roofline.traffic reads (tools/pmc_attn.py)
      rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- \
            python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1 )
        echo "exit $? : pmc attn $c"
      done
      python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc "${TAG}" > gpurun_out/pmc/summary.log 2>&1; tail -12 gpurun_out/pmc/summary.log
      find gpurun_out/pmc -name "*.db" -delete; find gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete ;;
    pmcenc)
      # where the encoder waves' cycles go (sq) and matrix-pipe busy cycles / MFMA op counts (mfma), bf16 + MXFP8 + f32 engines
      rm -rf gpurun_out/pmc_enc; mkdir -p gpurun_out/pmc_enc
      ( cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
          --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_enc" -o sq -- python "$R/tools/pmc_encoder.py" > "$R/gpurun_out/pmc_enc/sq.log" 2>&1 )
      echo "exit $? : encoder sq pass"
      ( cd /tmp && timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES \
          --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_enc" -o mfma -- python "$R/tools/pmc_encoder.py" > "$R/gpurun_out/pmc_enc/mfma.log" 2>&1 )
      echo "exit $? : encoder mfma pass"
      python tools/pmc_encoder_summary.py gpurun_out/pmc_enc "gpurun_out/${TAG}_pmc_encoder_summary.json" 2>&1 | tail -30
      find gpurun_out/pmc_enc -name "*.db" -delete; find gpurun_out/pmc_enc -name "*kernel_trace.csv" -size +4M -delete ;;
    pmcfe)
      rm -rf gpurun_out/pmc_fe; mkdir -p gpurun_out/pmc_fe
      i=0
      for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CU_CYCLES" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
                 "SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
        ( cd /tmp && timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fe" -o p$i -- python "$R/tools/pmc_frontend.py" > "$R/gpurun_out/pmc_fe/p$i.log" 2>&1 )
        echo "exit $? : frontend counter pass $i"
        i=$((i + 1))
      done
      python tools/pmc_frontend_summary.py gpurun_out/pmc_fe "gpurun_out/${TAG}_pmc_frontend_counters.json" 2>&1 | tail -12
      find gpurun_out/pmc_fe -name "*.db" -delete ;;
    micro)
      # stand-alone micro-benchmarks under tools/micro (MICRO = binary [args]); built here if the binary did not travel
      b=$(echo ${MICRO:-cu_split_groups} | cut -d' ' -f1)
      [ -x "tools/micro/$b" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "tools/micro/$b.hip" -o "tools/micro/$b" -lpthread
      ( cd tools/micro && timeout 600 ./${MICRO:-cu_split_groups} ) 2>&1 | tee "gpurun_out/${TAG}_micro_${b}.txt" | tail -20 ;;
    run)
      bash -c "$RUN_CMD"; echo "exit $? : run after $(( $(date +%s) - t0 )) s" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
