"""Reduce the two rocprofv3 --pmc passes of tools/pmc_decode.py PER KERNEL NAME (VERDICT r5 #3a).
HBM bytes of a launch = FETCH_SIZE x correction (gfx950 tallies 128-byte requests as 64: calibrated on the run's own 1 GiB
copy) + WRITE_SIZE; counters are KiB.  For every kernel the LAST `steps x launches-per-step` dispatches are kept (the
decode of tools/pmc_decode.py first walks to cache depth DEPTH), and the dense launches are set against the bytes they
need: their f32 weight matrix (read once per 64-row group) + the group's activation rows in and out.
Usage: python tools/pmc_decode_summary.py <dir with FETCH_SIZE / WRITE_SIZE outputs> <out.json> [steps=32] [groups=4]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
GROUPS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ROWS = 256 // GROUPS
EMB, HD, MLP, V, NL = 512, 384, 1024, 1536, 8


def rows(counter):
    fs = glob.glob(os.path.join(src, "**", "%s_counter_collection.csv" % counter), recursive=True)
    if not fs:
        raise SystemExit("no %s csv under %s" % (counter, src))
    out = []
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == counter:
            out.append((r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"]), int(r.get("Dispatch_Id", 0) or 0)))
    out.sort(key=lambda t: t[3])
    return out


fetch, write = rows("FETCH_SIZE"), rows("WRITE_SIZE")
cal_f = [v for n, g, v, _ in fetch if "copyBuffer" in n and g >= 131072][-1]
factor = (1 << 30) / (cal_f * 1024.0)


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mt3k::", "").replace("void ", "").strip()


def per_kernel(rs):
    d = defaultdict(list)
    for n, g, v, _ in rs:
        d[(short(n), g)].append(v)
    return d


F, W = per_kernel(fetch), per_kernel(write)
# what a dense launch of one 64-row group must read / write, by (N, K) of its weight matrix [N][K] f32
def dense_need(n, k, out_cols, a_cols):
    return {"weights": n * k * 4, "activations": ROWS * (a_cols + out_cols) * 4}


need = {
    "fold (MLP out-projection + next layer's q|k|v|cross-q, or the logits)": dense_need(EMB + 4 * HD, MLP + EMB, EMB + 4 * HD, MLP + EMB),
    "GEGLU wi": dense_need(2 * MLP, EMB, MLP, EMB),
    "self out-projection (+ cross-q)": dense_need(EMB + HD, HD, EMB + HD, HD),
    "cross out-projection": dense_need(EMB, HD, EMB, HD),
}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python tools/pmc_decode.py: f32 engine, "
                 "B = 256, MT3 shape, %d row groups of %d rows, direct launches, the last %d steps of a decode that first "
                 "walks to the mean cache depth; reduced per kernel name by tools/pmc_decode_summary.py.  Under --pmc the "
                 "dispatches of the four groups' streams run ONE AT A TIME (counter collection serialises them), in the "
                 "interleaved order the groups issue them: between two launches of the same weight matrix lie the other "
                 "groups' attention launches (a mean 100 MB each of K/V per group launch)" % (GROUPS, ROWS, STEPS),
       "units": "bytes per launch; FETCH_SIZE x %.4f (calibrated on the run's 1 GiB copy), WRITE_SIZE as is" % factor,
       "calibration_1GiB_copy": {"FETCH_SIZE_KiB": cal_f, "fetch_correction_factor": factor},
       "dense_launch_needs_bytes": need, "kernels": []}
for key in sorted(F, key=lambda k: -sum(F[k])):
    name, grid = key
    n_per_step = None
    f, w = F[key], W.get(key, [])
    keep = None
    for per_step in (GROUPS * NL * 2, GROUPS * NL, GROUPS):                # launches of this kernel per step: 64, 32 or 4
        if len(f) >= STEPS * per_step and len(f) % per_step == 0:
            keep, n_per_step = STEPS * per_step, per_step
            break
    fl, wl = (f[-keep:], w[-keep:]) if keep else (f, w)
    mf = sum(fl) / len(fl) * 1024.0 * factor
    mw = sum(wl) / len(wl) * 1024.0 if wl else 0.0
    out["kernels"].append({"kernel": name, "grid_size": grid, "dispatches_total": len(f), "dispatches_reduced": len(fl),
                           "launches_per_step": n_per_step, "hbm_read_bytes_per_launch": mf, "hbm_write_bytes_per_launch": mw,
                           "min_read": min(fl) * 1024.0 * factor, "max_read": max(fl) * 1024.0 * factor})
json.dump(out, open(dst, "w"), indent=1)
for k in out["kernels"][:24]:
    print("%-110s grid %7d x%6d  read %10.0f B  write %9.0f B" % (k["kernel"][:110], k["grid_size"], k["dispatches_reduced"],
                                                                 k["hbm_read_bytes_per_launch"], k["hbm_write_bytes_per_launch"]))
