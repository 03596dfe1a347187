#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc_enc
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|SQ_[A-Z0-9_]+)\b" | sort -u > "$R/gpurun_out/pmc_enc/counter_names.txt"
wc -l "$R/gpurun_out/pmc_enc/counter_names.txt"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_enc" -o sq -- python "$R/tools/pmc_encoder.py" > "$R/gpurun_out/pmc_enc/sq.log" 2>&1
echo "exit $? : sq pass"
cd "$R"
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_enc/**/sq_counter_collection.csv", recursive=True)
if not fs:
    print("no csv"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "gemm_kernel" not in n and "enc_attn" not in n: continue
    key = n[:95]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
for k, v in agg.items():
    w = v.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   " + "  ".join("%s=%.3f" % (c.replace("SQ_", ""), v[c] / w) for c in sorted(v) if c != "SQ_WAVE_CYCLES"), " dispatches", cnt[(k, "SQ_WAVE_CYCLES")])
PY
