"""Reduce the counter passes of tools/gpurun.sh stage pmcfe to one JSON for the log-mel kernel.
Usage: python tools/pmc_frontend_summary.py <dir with p*_counter_collection.csv / p*_kernel_trace.csv> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

src, out_path = sys.argv[1], sys.argv[2]
raw = collections.defaultdict(float)
disp = collections.Counter()
for f in sorted(glob.glob(os.path.join(src, "**", "p*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "logmel" not in r["Kernel_Name"]:
            continue
        raw[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]] += 1
dur = []
for f in sorted(glob.glob(os.path.join(src, "**", "p*_kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "logmel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
n = max(disp.values()) if disp else 0
per = {k: v / disp[k] for k, v in raw.items()}              # per dispatch (256 segments = 65,536 frames)
frames = 256 * 256
res = {"source": "rocprofv3 --pmc (tools/gpurun.sh stage pmcfe) over logmel_kernel<4> launches of 256 segments, MI355X",
       "dispatches_per_pass": n, "per_dispatch": per,
       "kernel_us_under_counters": sorted(dur)[len(dur) // 2] if dur else None}
g = per.get
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    # instruction counters count per WAVE; 16 frames per 4-wave workgroup = 4 frames per wave
    res["valu_insts_per_frame"] = g("SQ_INSTS_VALU") / frames
    res["lds_insts_per_frame"] = g("SQ_INSTS_LDS", 0.0) / frames
    res["salu_insts_per_frame"] = g("SQ_INSTS_SALU", 0.0) / frames
    res["waves"] = g("SQ_WAVES")
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
              "SQ_LDS_BANK_CONFLICT"):
        if g(k) is not None:
            res[k.lower() + "_per_wave_cycle"] = g(k) / g("SQ_WAVE_CYCLES")
if g("SQ_BUSY_CU_CYCLES"):
    # 4 cycles of a SIMD per wave64 VALU instruction; SQ_BUSY_CU_CYCLES adds up the busy cycles of the CUs.  The ratio
    # is the share of the SIMDs' issue cycles that VALU work of this kernel needs (1.0 = every SIMD of every busy CU
    # issues VALU back to back)
    if g("SQ_INSTS_VALU"):
        res["valu_issue_share"] = 4.0 * g("SQ_INSTS_VALU") / (4.0 * g("SQ_BUSY_CU_CYCLES"))
    if g("SQ_ACTIVE_INST_VALU") is not None:
        res["active_inst_valu_over_busy_cu_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CU_CYCLES")
    if g("SQ_ACTIVE_INST_LDS") is not None:
        res["active_inst_lds_over_busy_cu_cycles"] = g("SQ_ACTIVE_INST_LDS") / g("SQ_BUSY_CU_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE") is not None:
        res["lds_idx_active_over_busy_cu_cycles"] = g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CU_CYCLES")
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
