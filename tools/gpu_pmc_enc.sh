#!/bin/bash
# Two counter passes over one encoder pass at B=256 (tools/pmc_encoder.py):
#   sq   : where the waves' cycles go (parked / issue-stalled / issuing), LDS activity and conflicts
#   mfma : matrix-pipe busy cycles and MFMA op counts -> MFMA utilisation of the encoder kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/pmc_enc; mkdir -p gpurun_out/pmc_enc
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_enc" -o sq -- python "$R/tools/pmc_encoder.py" > "$R/gpurun_out/pmc_enc/sq.log" 2>&1
echo "exit $? : sq pass"
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_enc" -o mfma -- python "$R/tools/pmc_encoder.py" > "$R/gpurun_out/pmc_enc/mfma.log" 2>&1
echo "exit $? : mfma pass"
cd "$R"
python tools/pmc_encoder_summary.py gpurun_out/pmc_enc gpurun_out/pmc_enc/summary.json
find gpurun_out/pmc_enc -name "*.db" -delete; find gpurun_out/pmc_enc -name "*kernel_trace.csv" -size +4M -delete
