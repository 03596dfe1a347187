cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
one() { # label, env...
  l=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > /tmp/b.log 2>&1
  python - "$l" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open("/tmp/b.log") if x.startswith("{")][-1])
    r = d["roofline"]
    print("%-34s %7.1f audio-s/s  %8.1f ms/step  decode graph %7.1f  direct %7.1f ms" % (sys.argv[1], d["value"], d["ms_per_step"], r["decode_ms_product_schedule"], r["decode_ms_direct_launches"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/b.log").read()[-600:])
PY
}
{
one "default" X=1
one "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
one "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
one "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
one "ROC_USE_FGS_KERNARG=0" ROC_USE_FGS_KERNARG=0
one "default" X=1
} 2>&1 | tee gpurun_out/r6_ab_runtime_kernarg_env.txt
