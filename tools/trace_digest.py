"""Digest a rocprofv3 --kernel-trace csv: per-kernel totals and, for the decode-attention kernels,
mean duration as a function of the decode step (launch order / 8 layers)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + root)
rows, queues = [], []
with open(files[0]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                     int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
        queues.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                       int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
tot = defaultdict(lambda: [0, 0.0])
for s, e, n, _, _ in rows:
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
allus = sum(v[1] for v in tot.values())
print("kernel,calls,total_us,avg_us,pct")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%s,%d,%.1f,%.3f,%.2f" % (n[:110].replace(",", ";"), c, us, us / c, 100 * us / allus))
# decode attention by launch geometry: the timed passes of bench.py may deal the batch to 2 chains
# (half-batch launches on parallel graph branches), the roofline passes always launch the full batch
# (B*H workgroups = the largest grid); the roofline's avg_launch_us must agree with the latter.
groups = defaultdict(list)
for s, e, n, grid, wg in rows:
    if "dec_attn_kernel" in n or "dec_attn_fp8_kernel" in n:
        # APPEND is the second template argument: mangled ...IDF16bLb0E... / demangled "<float, false, ..."
        kind = "cross(no append)" if ("Lb0ELi" in n or "<float, false" in n or "<__bf16, false" in n) else "self(append)"
        groups[(kind, grid // max(wg, 1))].append((e - s) / 1e3)
for (kind, wgs), d in sorted(groups.items()):
    line = "dec_attn %s, %d workgroups: %d launches, avg %.3f us" % (kind, wgs, len(d), sum(d) / len(d))
    if len(d) >= 8192 and len(d) % 8192 == 0:
        one = d[-8192:]
        per = [sum(one[i * 8:(i + 1) * 8]) / 8 for i in range(1024)]
        line += "; last pass by step t=0,15,63,127,255,511,767,1023: %s" % [round(per[i], 1) for i in
                                                                            (0, 15, 63, 127, 255, 511, 767, 1023)]
    print(line)

# ---- concurrency inside the row-group schedule: the decode-attention launches of the row groups have a smaller grid than
# the full-batch roofline passes; bursts of them (gaps < 5 ms) are the product-schedule decodes.  Inside those bursts: how
# much of the wall time has k attention kernels in flight (k = 0: the HBM stream is idle), and how much has no kernel at all.
attn = [(s, e, grid // max(wg, 1)) for s, e, n, grid, wg in rows if "dec_attn" in n]
if attn:
    full = max(w for _, _, w in attn)
    small = sorted((s, e) for s, e, w in attn if w < full)
    if small:
        bursts, cur = [], [small[0][0], small[0][1]]
        for s, e in small[1:]:
            if s - cur[1] > 5_000_000:
                bursts.append(cur)
                cur = [s, e]
            cur[1] = max(cur[1], e)
        bursts.append(cur)
        bursts = [b for b in bursts if b[1] - b[0] > 50_000_000]            # whole decodes only (> 50 ms)
        hist, idle, span = defaultdict(float), 0.0, 0.0
        for b0, b1 in bursts:
            ev = []
            for s, e, n, grid, wg in rows:
                if e <= b0 or s >= b1:
                    continue
                is_attn = 1 if "dec_attn" in n else 0
                ev.append((max(s, b0), 1, is_attn))
                ev.append((min(e, b1), -1, is_attn))
            ev.sort()
            k_all = k_attn = 0
            last = b0
            for t, d, a in ev:
                dt = t - last
                hist[k_attn] += dt
                if k_all == 0:
                    idle += dt
                last = t
                k_all += d
                k_attn += d * a
            span += b1 - b0
        if span:
            print("row-group decodes: %d bursts, %.1f ms; share of wall time with k decode-attention kernels in flight: %s; "
                  "no kernel at all in flight: %.3f" % (len(bursts), span / 1e6,
                                                         {k: round(v / span, 3) for k, v in sorted(hist.items())}, idle / span))


# ---- inter-kernel gaps per hardware queue inside the row-group decodes (round 6): end of a kernel -> start of the next one
# on the SAME queue, by what the two kernels are.  CAVEAT: under rocprofv3 --kernel-trace a row-group decode takes 2.6x its
# un-profiled time and a kernel's start stamp is the end stamp of its predecessor on the queue for most boundaries (median
# gap 0): these figures describe the profiler's serialisation, not the product's launch boundaries
def kind(n):
    if "dec_attn" in n:
        return "attention"
    if "argmax_step" in n:
        return "argmax"
    if "gemm_kernel" in n:
        return "dense"
    return "other"


if attn and small and bursts:
    byq = defaultdict(list)
    for s, e, n, q, grid in queues:
        if any(b0 <= s and e <= b1 for b0, b1 in bursts):
            byq[q].append((s, e, kind(n)))
    gaps = defaultdict(list)
    for q, ks in byq.items():
        ks.sort()
        for (s0, e0, k0), (s1, e1, k1) in zip(ks, ks[1:]):
            if s1 - e0 < 50_000 and k0 != "other" and k1 != "other":      # (a poll / refill pause is not a launch gap)
                gaps[(k0, k1)].append((s1 - e0) / 1e3)
    print("queues with decode work: %d" % len(byq))
    for (k0, k1), g in sorted(gaps.items()):
        g.sort()
        print("gap %-9s -> %-9s: %7d boundaries, mean %.2f us, median %.2f, p90 %.2f" % (k0, k1, len(g), sum(g) / len(g),
                                                                                         g[len(g) // 2], g[len(g) * 9 // 10]))
