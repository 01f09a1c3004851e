#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per (kernel, grid)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'sow' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:48], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
