#!/bin/bash
# Experiment (GPU box): cache-policy bits on the stream kernel's gathers.  usage: scripts/exp_aux.sh
for aux in 1 2 3 16 17; do
  HYB=4:31 scripts/exp_variant.sh "-DISPLIB_EXP_GATHER_AUX=$aux" python scripts/exp_hybrid.py 128 2>&1 | grep "variant\|stream form"
done
