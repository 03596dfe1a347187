cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 4 7 8 7 8; do
  ( cd tools/micro && timeout 90 ./cu_split_groups $v ) 2>&1 | tee -a gpurun_out/r6_micro_cu_split_groups_b.txt | tail -2
done
timeout 300 python tools/attn_width_probe.py 2>&1 | tee gpurun_out/r6_attn_width_probe.txt | tail -6
timeout 1500 python tools/ab_r6.py run 0 1 2 3 4 6 0 2>&1 | tee gpurun_out/r6_ab_decode_gemm_variants.txt | tail -12
