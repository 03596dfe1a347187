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
rows = []
with open(files[0]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                     int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
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
