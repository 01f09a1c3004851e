#!/bin/bash
# A/B of kernel-selection switches on the headline bench: tools/switch_ab.sh "NAME=1" "NAME2=1 NAME3=1" ...
for cfg in "" "$@"; do
  env $(for kv in $cfg; do echo SOW_AMD_$kv; done) python bench.py --only-headline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pl=d['per_launch']
print('[$cfg]', 'ms', round(d['ms_per_step'],3), {k.replace('chain','').replace(' (dX)','')[:26]: v['us'] for k, v in pl.items()})"
done
