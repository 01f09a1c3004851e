#!/bin/bash
# A/B of bench.py command-line variants: tools/bench_ab.sh "--tn-group group" "--group none" ...
for cfg in "" "$@"; do
  python bench.py --only-headline --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pl=d['per_launch']
print('[$cfg]', 'ms', round(d['ms_per_step'],3), 'fwd', round(d['kernel_groups_ms']['forward (chain kernels)'],3), {k.replace('chain','').replace(' (dX)','')[:26]: v['us'] for k, v in pl.items() if 'partials' in k})"
done
