cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 400 python tools/ids_hash.py 2>&1 | tail -1
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.1f audio-s/s %.1f ms/step decode %.1f direct %.1f' % (d['value'], d['ms_per_step'], r['decode_ms_product_schedule'], r['decode_ms_direct_launches']))"; done
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_parity_r5.py tests/test_gpu_parity_deep.py tests/test_gpu_transcribe.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpurun.sh prof 2>&1 | grep -E "argmax|exit"
