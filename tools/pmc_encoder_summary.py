"""Reduce the counter passes of tools/gpurun.sh stage pmcenc to one JSON: per kernel of this library, the share of wave
cycles parked / stalled / issuing, LDS activity, and the matrix-pipe utilisation.
Usage: python tools/pmc_encoder_summary.py <dir with sq_/mfma_ csv> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out_path = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"gemm_kernelIDF16bLi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELb(\d)ELb(\d)ELi(\d)", name)
    if m:
        epi = ["STORE", "RESID", "GEGLU", "POS", "F32", "HEADS"][int(m.group(6))]
        return "gemm %sx%sx%s %s%s%s" % (m.group(1), m.group(2), m.group(3), "f32A " if m.group(4) == "1" else "",
                                         "norm " if m.group(5) == "1" else "", epi)
    m = re.search(r"gemm_(glds|mx8)_kernel<(\d+), (\d+)(?:, (\d+))?>", name) or \
        re.search(r"gemm_(glds|mx8)_kernelILi(\d+)ELi(\d+)(?:ELi(\d+))?", name)
    if m:
        epi = ["STORE", "RESID", "GEGLU", "POS", "F32", "HEADS"][int(m.group(2))]
        bm = m.group(4) or "128"
        return ("gemm LDS-DMA bf16 %sx128x32 " % bm if m.group(1) == "glds" else "gemm MXFP8 128x128x128 ") + epi
    m = re.search(r"gemm_x6_kernelILb(\d)ELi(\d)", name) or re.search(r"gemm_x6_kernel<(true|false|\d), (\d)", name)
    if m:
        epi = ["STORE", "RESID", "GEGLU", "POS", "F32", "HEADS"][int(m.group(2))]
        return "gemm f32 as 3 bf16 planes 128x128x32 " + ("norm " if m.group(1) in ("1", "true") else "") + epi
    if "mx8_quantize" in name:
        return "mx8_quantize"
    if "enc_attn" in name:
        return "enc_attn f32" if ("IfL" in name or "<float" in name) else "enc_attn"
    if "logmel" in name:
        return "logmel"
    return None


def load(prefix):
    fs = glob.glob(os.path.join(src, "**", "%s_counter_collection.csv" % prefix), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    if not fs:
        return agg, n
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            n[k] += 1
    return agg, n


sq, nsq = load("sq")
mf, nmf = load("mfma")
res = {"source": "rocprofv3 --pmc (two passes, tools/gpurun.sh stage pmcenc) over one encoder pass of 256 segments, MI355X",
       "note": "ratios of raw counters summed over all dispatches of the kernel; mfma_util = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): the busy counter adds up the 4 SIMDs of a CU (the "
               "gfx94x MfmaUtil formula); flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512",
       "kernels": {}}
for k in sorted(set(sq) | set(mf)):
    e = {"dispatches": int(max(nsq.get(k, 0), nmf.get(k, 0)))}
    if k in sq and sq[k].get("SQ_WAVE_CYCLES"):
        w = sq[k]["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
                  "SQ_LDS_BANK_CONFLICT"):
            e[c.replace("SQ_", "").lower() + "_per_wave_cycle"] = round(sq[k].get(c, 0.0) / w, 4)
    if k in mf:
        m = mf[k]
        if m.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_util"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_BUSY_CU_CYCLES"]), 4)
        e["mfma_mops_bf16"] = m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        e["mfma_flops"] = e["mfma_mops_bf16"] * 512.0
        e["mfma_insts"] = m.get("SQ_INSTS_MFMA", 0.0)
        e["raw"] = {c: m[c] for c in sorted(m)}
    res["kernels"][k] = e
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
for k, e in res["kernels"].items():
    print(k, {a: b for a, b in e.items() if a not in ("raw",)})
