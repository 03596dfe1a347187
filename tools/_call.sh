cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 2400 python tools/train_synthetic.py --steps 60000 --minutes 30 --out gpurun_out/mt3_synthetic_ckpt_v2.npz > gpurun_out/r6_train_v2.log 2>&1
echo "exit $? train"; grep -E "held_out" gpurun_out/r6_train_v2.log | cut -c1-250 | tail -40; tail -2 gpurun_out/r6_train_v2.log | cut -c1-900
TOL_ARGS="--weights gpurun_out/mt3_synthetic_ckpt_v2.npz" bash tools/gpurun.sh tol
TOL_ARGS="--weights gpurun_out/mt3_synthetic_ckpt_v2.npz --minutes 5" bash tools/gpurun.sh tol
TOL_ARGS="--weights tests/golden/mt3_synthetic_ckpt.npz --minutes 5" bash tools/gpurun.sh tol
