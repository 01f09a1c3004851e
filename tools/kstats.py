#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) median / min / total."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
order = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "sow" not in name:
        continue
    k = (name[:70], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:72s} blocks={k[1]:5d} n={len(v):4d} med={v2[len(v2)//2]/1e3:7.1f}us min={v2[0]/1e3:7.1f} total={sum(v)/1e6:7.2f}ms")
if len(sys.argv) > 2:  # per-call sequence for one kernel substring
    sub = sys.argv[2]
    seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    print(sub, " ".join(f"{x:.1f}" for x in seq))
