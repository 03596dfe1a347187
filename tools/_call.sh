cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 4 12 11 10; do
  ( cd tools/micro && timeout 120 ./cu_split_groups $v ) 2>&1 | tee -a gpurun_out/r6_micro_any_order_e.txt | tail -2
done
