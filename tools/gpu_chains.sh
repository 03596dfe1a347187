#!/bin/bash
# bench at 1..4 decode chains (graph branches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --chains $c > gpurun_out/chains_$c.log 2>&1
  echo "chains=$c exit $?"
  tail -1 gpurun_out/chains_$c.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  value %.1f  ms/step %.1f  roofline %s' % (d['value'], d['ms_per_step'], {k: d['roofline'][k] for k in ('achieved', 'frac')}))
"
done
