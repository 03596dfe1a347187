#!/bin/bash
# encoder GEMM work: kernel + engine parity, then the stage extras (encoder TF/s, configs[1]) for A/B env switches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_parity_deep.py -m gpu -q --tb=short -p no:cacheprovider -x ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_enc.log 2>&1
echo "exit $? : tests"; grep -v "^/opt\|^$" gpurun_out/pytest_enc.log | tail -${TAILN:-15}
for e in "X=1" ${AB_ENVS}; do
  env $e timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --decode-steps 64 > gpurun_out/bench_enc_$e.log 2>&1
  echo "bench $e exit $?"
  tail -1 gpurun_out/bench_enc_$e.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
x = d['extra']
print('  encoder %.3f ms %.0f TF/s (%.3f of peak) | configs1 %.0f seg/s %.0f TF/s | frontend %.0f GB/s' % (x['encoder']['ms'], x['encoder']['achieved'], x['encoder']['frac'], x['configs1']['segments_per_s'], x['configs1']['achieved'], x['frontend']['achieved']))
" 2>&1 | tail -2
done
