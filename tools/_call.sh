cd "${GRAFT_REPO_ROOT:-/root/repo}"
nproc; free -g | head -2
timeout 1500 python tools/train_synthetic.py --steps 30000 --minutes 14 --out gpurun_out/mt3_synthetic_ckpt.npz > gpurun_out/r6_train.log 2>&1
echo "exit $? train"; grep -E "TRAIN" gpurun_out/r6_train.log | awk 'NR%10==1' | cut -c1-250 | tail -25; tail -3 gpurun_out/r6_train.log | cut -c1-900
TOL_ARGS="--weights gpurun_out/mt3_synthetic_ckpt.npz" bash tools/gpurun.sh tol
