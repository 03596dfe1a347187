#!/bin/bash
# round 5, last GPU call: the whole GPU suite + smoke on the final sources; BASELINE configs[3] canonical on one GPU; the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5_last_pytest.log 2>&1
echo "exit $? : pytest -m gpu after $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/r5_last_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
timeout 900 python bench.py --corpus 10000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5_corpus.log 2>&1
echo "exit $? : bench --corpus 10000 after $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r5_corpus.log | tail -1 > gpurun_out/r5_bench_corpus10000_n1_f32.json
python -c "
import json; d = json.load(open('gpurun_out/r5_bench_corpus10000_n1_f32.json')); r = d['roofline']
print('corpus canonical', d['value'], d['ms_per_step'], 'self-attn frac', r['frac'], 'whole step', r['whole_step_hbm_frac_product_schedule'])"
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_like.log 2>&1
echo "exit $? : bench after $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r5_bench_driver_like.log | tail -1 > gpurun_out/r5_bench_driver_like_2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_driver_like_2.json"))
print("value", d["value"], d["dtype"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("parity rows", d["cpu_baseline"]["parity"]["token_exact_rows"], d["cpu_baseline"]["parity"]["notes_equal"])
x = d["extra"]
print("eos corpus", x["eos_schedule_corpus"]["f32"]["value"], "single file", x["single_file"].get("speedup"), x["single_file"].get("notes_identical"))
print("errors:", {k: v.get("error") for k, v in x.items() if isinstance(v, dict) and v.get("error")})
PY
