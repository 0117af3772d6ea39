#!/bin/bash
# slices x chunk x short-row sweep of the task plan on the headline workload (K=128 runs as two 64-column panels)
for S in ${SLICES:-7 8 9}; do for C in ${CHUNKS:-512 1024 2048}; do for SH in ${SHORTS:-64 128 192 256}; do
python bench.py --steps 10 --warmup 3 --slices $S --chunk $C --short $SH --no-cpu-baseline --no-backward 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('S=$S chunk=$C short=$SH ms', round(r['ms_per_step'],3))"
done; done; done
