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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tot = defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
allus = sum(v[1] for v in tot.values())
print("kernel,calls,total_us,avg_us,pct")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%s,%d,%.1f,%.3f,%.2f" % (n[:110].replace(",", ";"), c, us, us / c, 100 * us / allus))
for key in ("Lb1EEEvNS_11DecAttnArgs", "Lb0EEEvNS_11DecAttnArgs", "dec_attn_kernel"):
    d = [(e - s) / 1e3 for s, e, n in rows if key in n]
    if len(d) >= 8192:
        one = d[:8192]
        per = [sum(one[i * 8:(i + 1) * 8]) / 8 for i in range(1024)]
        print(key, "first decode pass: avg us per launch at t=0,15,63,127,255,511,767,1023:",
              [round(per[i], 1) for i in (0, 15, 63, 127, 255, 511, 767, 1023)], "mean", round(sum(per) / 1024, 2))
