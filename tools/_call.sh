cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tools/ab_r6.py run 0 64 0 64 2>&1 | tee gpurun_out/r6_ab_kernarg_pin.txt | tail -6
PH_STEPS=256 timeout 600 python tools/ab_r6.py phases 2>&1 | tee gpurun_out/r6_gemm_phases_in_situ_pinned.txt | grep -A5 "^1\.\|^4\." | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_parity_r5.py -m gpu -q -x 2>&1 | tail -4
