cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { # name args...
  n=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-extras "$@" > "gpurun_out/r6_bench_$n.log" 2>&1
  grep '^{' "gpurun_out/r6_bench_$n.log" | tail -1 > "gpurun_out/r6_bench_$n.json"
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6_bench_%s.json" % n))
    r = d.get("roofline") or {}
    print("%-22s %8.1f %s  %9.1f ms/step  dtype %s  attn frac %s  whole decode %s" % (n, d["value"], d["unit"], d["ms_per_step"], d["dtype"], r.get("frac"), r.get("whole_step_hbm_frac_product_schedule")))
except Exception as e:
    print(n, "FAILED", e)
PY
}
run batch_128_f32 --batch 128 --steps 3 --warmup 1
run batch_512_f32 --batch 512 --steps 3 --warmup 1
run batch_1024_f32 --batch 1024 --steps 3 --warmup 1
run beam1_f32 --decoding beam1 --steps 3 --warmup 1
run bf16_headline --dtype bfloat16 --steps 8 --warmup 2
run corpus10000_n1_f32 --corpus 10000 --steps 1 --warmup 1
