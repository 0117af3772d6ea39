#!/bin/bash
# usage: sweep.sh "<bench args>" s1 s2 ...   -> one line per slice count
args=$1; shift
for s in "$@"; do
  python bench.py --steps 10 --warmup 3 --slices $s --no-cpu-baseline --no-backward $args 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$args', 'S=$s', 'Gedges/s', round(r['value']/1e9,2), 'ms', round(r['ms_per_step'],3), 'B_alg GB/s', round(r['roofline']['achieved'],1), 'gather TB/s', round(r['roofline']['gather_model_GBps']/1e3,2))"
done
