#!/bin/bash
# One gpurun call: per-group GPU parity tests (separate processes, so a faulting kernel cannot hide
# the other results), smoke(), then a short bench.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== box"; rocm-smi --showproductname 2>/dev/null | head -8
  python -c "import torch;print('torch',torch.__version__,'gpus',torch.cuda.device_count(), torch.cuda.get_device_name(0))"
  nproc
} > gpurun_out/box.log 2>&1
for grp in "test_gpu_kernels.py -k gemm" "test_gpu_kernels.py -k attention" "test_gpu_kernels.py -k ids_to_tokens" \
           "test_gpu_kernels.py -k frontend" "test_gpu_engine.py"; do
  name=$(echo "$grp" | tr ' ./' '___')
  timeout 420 python -m pytest tests/$grp -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$name.log 2>&1
  echo "exit $? : $grp" >> gpurun_out/summary.log
  tail -3 gpurun_out/pytest_$name.log >> gpurun_out/summary.log
done
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "exit $? : smoke" >> gpurun_out/summary.log
timeout 400 python bench.py ${BENCH_ARGS:---batch 64 --decode-steps 256 --steps 1 --warmup 1 --no-cpu-baseline} > gpurun_out/bench_small.log 2>&1
echo "exit $? : bench_small" >> gpurun_out/summary.log
cat gpurun_out/summary.log
tail -5 gpurun_out/bench_small.log
