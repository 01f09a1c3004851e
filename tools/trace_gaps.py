#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (the last
graph replay of bench.py): where the step time goes outside the kernels' own begin..end."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "sow" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 73
rows = rows[-n:]
tot_k = tot_g = 0
agg = {}
for a, b in zip(rows, rows[1:] + [None]):
    d = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 if b else 0.0
    key = (a["Kernel_Name"].split("(")[0][-40:], int(a["Grid_Size_X"]) // max(int(a["Workgroup_Size_X"]), 1))
    e = agg.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1; e[1] += d; e[2] += g
    tot_k += d; tot_g += g
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:42s} grid {k[1]:5d} n={c:3d} avg dur {d / c:7.2f} us  avg gap after {g / c:5.2f} us")
print(f"kernels {tot_k / 1e3:.3f} ms + gaps {tot_g / 1e3:.3f} ms over {len(rows)} launches")
