#!/bin/bash
# Experiment helper (GPU box): rebuild libisplib_hip.so with extra defines for spmm_sweep.hip, run a command, restore.
# usage: scripts/exp_variant.sh "-DISPLIB_STREAM_NBW4=1" python scripts/exp_hybrid.py 128
set -e
extra="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cp "$root/isplib_amd/libisplib_hip.so" /tmp/libisplib_hip.keep
touch "$root/isplib_amd/csrc/spmm_sweep.hip"
make -s -C "$root/isplib_amd/csrc" EXTRA="$extra" "$root/isplib_amd/libisplib_hip.so" > /tmp/variant_build.log 2>&1 || { cat /tmp/variant_build.log; exit 1; }
echo "== variant $extra"
"$@" || true
cp /tmp/libisplib_hip.keep "$root/isplib_amd/libisplib_hip.so"
touch "$root/isplib_amd/csrc/spmm_sweep.hip"
