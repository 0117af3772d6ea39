#!/bin/bash
# Experiment (GPU box): geometry of the 32-column stream kernel (rows per wave / batch registers / workgroups per CU) at K = 32.
for g in "64 2 3" "128 4 2" "128 2 2" "96 2 2" "128 3 2"; do
  set -- $g
  HYB=4:31 scripts/exp_variant.sh "-DISPLIB_STREAM_NV8=$1 -DISPLIB_STREAM_NBW8=$2 -DISPLIB_STREAM_WGS8=$3" python scripts/exp_hybrid.py 32 2>&1 | grep "variant\|stream form"
done
