cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in 0 4 5 6 7 8 0 5 6; do
  ( cd tools/micro && timeout 90 ./cu_split_groups $v ) 2>&1 | tee -a gpurun_out/r6_micro_cu_split_groups.txt | tail -3
done
timeout 1200 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_note_tolerance.py -m gpu -q --durations=12 2>&1 | tail -25
