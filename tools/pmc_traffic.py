#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE must be collected separately:
/opt/skills/guides/MI355X_MICROARCH.md, TCC has 4 counter slots).  Writes the JSON bench.py reads for `roofline.traffic`.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/fetch -- python3 bench.py --no-graph --steps 3 --warmup 1 --only-headline --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/write -- python3 bench.py ... (same)
    python tools/pmc_traffic.py <out>/fetch <out>/write <commit> profiles/r02_pmc_hbm_traffic.json

Units and corrections as the guide prescribes: both counters are reported in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide streaming reads at 64 bytes, i.e. reports exactly HALF of the bytes -> x2; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "sow::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def short(name):
    m = re.search(r"sow::([A-Za-z0-9_]+(?:<[a-z, ]+>)?)", name)
    return m.group(1) if m else name[:40]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"commit": sys.argv[3], "group": "per decoder block (default bench.py grouping)", "unit": "bytes per launch",
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x2 (gfx950)",
       "kernels": {}, "by_grid": {}}
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for key in sorted(set(fetch) | set(write)):
    fv, wv = fetch.get(key, []), write.get(key, [])
    rd = 2 * 1024 * sum(fv) / max(len(fv), 1)
    wr = 1024 * sum(wv) / max(len(wv), 1)
    n = max(len(fv), len(wv))
    out["by_grid"][f"{short(key[0])} grid_threads={key[1]}"] = {"launches": n, "read_bytes": round(rd), "written_bytes": round(wr)}
    t = tot[short(key[0])]
    t[0] += n
    t[1] += rd * n
    t[2] += wr * n
for k, (n, rd, wr) in tot.items():
    out["kernels"][k] = {"launches": n, "read_bytes_per_launch": round(rd / n), "written_bytes_per_launch": round(wr / n),
                         "hbm_bytes_per_launch": round((rd + wr) / n)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:36s} n={v['launches']:4d} read {v['read_bytes_per_launch'] / 1e6:8.2f} MB written {v['written_bytes_per_launch'] / 1e6:8.2f} MB per launch")
