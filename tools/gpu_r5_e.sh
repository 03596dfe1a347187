#!/bin/bash
# round 5, GPU call E: the driver's own bench command, timed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_like.log 2>&1
echo "exit $? after $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r5_bench_driver_like.log | tail -1 > gpurun_out/r5_bench_driver_like.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_driver_like.json"))
print("value", d["value"], d["dtype"], "ms/step", d["ms_per_step"])
print("roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
cb = d.get("cpu_baseline", {})
print("cpu_baseline", cb.get("value"), cb.get("cores"), "parity", json.dumps(cb.get("parity"))[:1500])
x = d["extra"]
print("eos_schedule", {k: (v.get("value"), v.get("decode_ms")) for k, v in x.get("eos_schedule", {}).items()})
print("eos_schedule_corpus", json.dumps(x.get("eos_schedule_corpus"))[:1800])
print("single_file", json.dumps(x.get("single_file"))[:1500])
print("errors:", {k: v.get("error") for k, v in x.items() if isinstance(v, dict) and v.get("error")}, d.get("cpu_parity_error"))
PY
tail -3 gpurun_out/r5_bench_driver_like.log | cut -c1-300
