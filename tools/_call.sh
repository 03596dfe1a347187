cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 4 13 14 5 13 14; do
  ( cd tools/micro && timeout 120 ./cu_split_groups $v ) 2>&1 | tee -a gpurun_out/r6_micro_persistent_nofence.txt | tail -3
done
