#!/bin/bash
# Counter passes over the log-mel kernel at 256 segments (tools/pmc_frontend.py): instruction counts and busy cycles of
# the VALU and of the LDS, wave-cycle breakdown -> gpurun_out/pmc_fe/summary.json (tools/pmc_frontend_summary.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/pmc_fe; mkdir -p gpurun_out/pmc_fe
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fe" -o p$i -- python "$R/tools/pmc_frontend.py" > "$R/gpurun_out/pmc_fe/p$i.log" 2>&1
  echo "exit $? : frontend counter pass $i ($set)"
  i=$((i + 1))
done
cd "$R"
python tools/pmc_frontend_summary.py gpurun_out/pmc_fe gpurun_out/pmc_fe/summary.json
find gpurun_out/pmc_fe -name "*.db" -delete
