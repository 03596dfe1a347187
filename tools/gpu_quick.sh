#!/bin/bash
# quick check after a decode-path change: engine + kernel parity tests, then the headline bench twice (no extras, no CPU leg)
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/pytest_quick.log 2>&1; echo "exit $? : pytest"
tail -3 gpurun_out/pytest_quick.log
for i in 1 2; do
  timeout 200 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/bench_quick_$i.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_quick_$i.log").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["decode_ms_single_chain"], d["roofline"]["decode_ms_without_self_attn"], d["roofline"]["avg_launch_us"])
PY
done
