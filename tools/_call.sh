cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tools/ab_r6.py run 0 128 384 896 0 2>&1 | tee gpurun_out/r6_ab_narrow_tiles.txt | tail -6
PH_STEPS=256 timeout 600 python tools/ab_r6.py phases 928 2>&1 | tee gpurun_out/r6_gemm_phases_narrow.txt | grep -A5 "^1\.\|^4\." | cut -c1-30,88-330
