"""Reduce the two rocprofv3 --pmc passes of tools/pmc_attn.py (tools/gpu_pmc.sh) to profiles/<round>_pmc_summary.json
(stamped with the hash of the kernel sources the counters were collected from; bench.py refuses a summary whose
hash differs from the sources it runs):
HBM traffic per decode-attention launch = FETCH_SIZE x 2 (gfx950 tallies 128-byte requests as 64: calibrated on
the 1 GiB copy of the same run) + WRITE_SIZE, against the algorithmic bytes of the launch.
Usage: python tools/pmc_summary.py gpurun_out/pmc profiles [round-prefix, default r3]"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FRONTEND_SOURCES, kernel_source_hash  # noqa: E402
B, H, D, CAP = 256, 6, 64, 1024


def rows(counter):
    fs = glob.glob(os.path.join(src, "**", "%s_counter_collection.csv" % counter), recursive=True)
    if not fs:
        raise SystemExit("no %s csv under %s" % (counter, src))
    shutil.copy(fs[0], os.path.join(dst, "%s_pmc_%s_counter_collection.csv" % (RND, counter)))
    out = []
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == counter:
            out.append((r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out


fetch, write = rows("FETCH_SIZE"), rows("WRITE_SIZE")


def pick(rs, pred):
    return [v for n, g, v in rs if pred(n, g)]


# calibration: the LAST 1 GiB f32 clone (a blit kernel: 1 GiB read + 1 GiB written)
is_copy = lambda n, g: "copyBuffer" in n and g >= 131072
cal_f, cal_w = pick(fetch, is_copy)[-1], pick(write, is_copy)[-1]
factor = (1 << 30) / (cal_f * 1024.0)

summary = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python tools/pmc_attn.py "
              "(tools/gpu_pmc.sh, reduced by tools/pmc_summary.py), MI355X, " + RND,
    "kernel_source_hash": kernel_source_hash(),
    "frontend_source_hash": kernel_source_hash(FRONTEND_SOURCES),
    "units": "counter values are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: "
             "MI355X_MICROARCH.md, HBM section), WRITE_SIZE is used as is",
    "calibration_1GiB_copy": {"FETCH_SIZE_KiB": cal_f, "WRITE_SIZE_KiB": cal_w, "read_bytes_true": 1 << 30,
                              "fetch_correction_factor": factor},
    "shape": {"B": B, "H": H, "head_dim": D, "cap": CAP},
}


def entry(f_kib, w_kib, n_keys, append, kind, H=H):
    # algorithmic: K and V rows of every cached key, the query, the output (+ the new K/V row read and written);
    # fp8: 64 e4m3 bytes per K / V row + the 8-byte scale pair of the position; q / out / new rows bf16
    es = 4 if kind == "f32" else 2
    n_cached = n_keys - (1 if append else 0)
    if kind == "fp8":
        alg = B * H * (n_cached * (2 * D + 8) + 2 * D * es + (2 * D * es + 2 * D + 8 if append else 0))
    else:
        alg = B * H * (2 * n_cached * D * es + D * es + D * es + (4 * D * es if append else 0))
    traffic = f_kib * 1024 * 2 + w_kib * 1024
    return {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "traffic_bytes": traffic, "algorithmic_bytes": alg,
            "traffic_over_algorithmic": traffic / alg}


# launch order of pmc_attn.py per cache format: (1024, 513, 129) x 2 rounds x 4 layers of self/append, then 8 cross
# (rocprofv3 prints the __bf16 instantiations either mangled -- ...dec_attn_kernelIDF16b... -- or mis-demangled as
# "dec_attn_kernel<bool _Accum, ...>")
H12 = (B * 12 * 192, B * 12 * 256)             # work-items of the 12-head fp8 launches (self: 3 waves, cross: 4)
KIND = {"bf16": lambda n, g: "dec_attn_kernel" in n and "dec_attn_kernel<float" not in n,
        "f32": lambda n, g: "dec_attn_kernel<float" in n,
        "fp8": lambda n, g: "dec_attn_fp8_kernel" in n and g not in H12,
        "fp8_h12": lambda n, g: "dec_attn_fp8_kernel" in n and g in H12}
for kind, pred in KIND.items():
    heads = 12 if kind.endswith("_h12") else H
    fmt = kind.split("_")[0]
    af, aw = pick(fetch, pred), pick(write, pred)
    if len(af) != 32 or len(aw) != 32:
        print("skipping %s: %d / %d dispatches (expected 32)" % (kind, len(af), len(aw)))
        continue
    sec = {}
    for i, n_keys in enumerate((1024, 513, 129)):
        sel = list(range(12 + i * 4, 12 + i * 4 + 4))        # second round: caches hold nothing of these layers
        sec["n_keys_%d" % n_keys] = entry(sum(af[j] for j in sel) / 4, sum(aw[j] for j in sel) / 4, n_keys, True, fmt, heads)
    summary["dec_attn_self_append_" + kind] = sec
    summary["dec_attn_cross_256_keys_" + kind] = entry(sum(af[24:]) / 8, sum(aw[24:]) / 8, 256, False, fmt, heads)
# log-mel frontend: 3 launches of 256 full segments
is_fe = lambda n, g: "logmel_kernel" in n
ff, fw = pick(fetch, is_fe), pick(write, is_fe)
if ff and fw:
    f_kib, w_kib = sum(ff[-3:]) / len(ff[-3:]), sum(fw[-3:]) / len(fw[-3:])
    alg = 256 * 655360
    tr = f_kib * 1024 * 2 + w_kib * 1024
    summary["logmel_kernel_256_segments"] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "traffic_bytes": tr,
                                             "algorithmic_bytes": alg, "traffic_over_algorithmic": tr / alg}
with open(os.path.join(dst, "%s_pmc_summary.json" % RND), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps({k: v for k, v in summary.items() if k.startswith(("dec_attn", "calib", "logmel", "kernel_source"))}, indent=1))
