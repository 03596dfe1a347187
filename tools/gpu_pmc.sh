#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc" -o $c -- python "$R/tools/pmc_attn.py" > "$R/gpurun_out/pmc/$c.log" 2>&1
  echo "exit $? : $c"
done
cd "$R"
ls gpurun_out/pmc
python - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmc/**/%s_counter_collection.csv" % c, recursive=True)
    if not fs:
        print("no csv for", c); continue
    rows = list(csv.DictReader(open(fs[0])))
    print(c, "columns:", list(rows[0].keys()))
    for r in rows:
        n = r.get("Kernel_Name", "")
        if "dec_attn" in n or "copy" in n.lower() or "elementwise" in n.lower():
            print(c, n[:70], r.get("Counter_Name"), r.get("Counter_Value"), "grid", r.get("Grid_Size"))
PY
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc > gpurun_out/pmc/summary.log 2>&1; tail -45 gpurun_out/pmc/summary.log
