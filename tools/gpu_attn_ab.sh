#!/bin/bash
# decode-attention check: kernel + engine parity tests, then the full bench workload per waves-per-group setting
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or engine or decode or chains or graph or full or inference or beam" > gpurun_out/pytest_attn.log 2>&1
echo "exit $? : tests"; tail -12 gpurun_out/pytest_attn.log
for w in 3 4 2; do
  MT3_DEC_ATTN_WAVES=$w timeout 300 python bench.py --no-cpu-baseline > gpurun_out/ab_w$w.log 2>&1
  echo "waves=$w exit $?"
  tail -1 gpurun_out/ab_w$w.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  value %.1f  ms/step %.1f  roofline %s' % (d['value'], d['ms_per_step'], {k: d['roofline'][k] for k in ('achieved', 'frac')}))
"
done
