#!/bin/bash
# Experiment helper: rebuild the C library ON THE GPU BOX with extra -D flags (geometry macros of spmm_sweep.hip) and run a
# command per variant.  usage: exp_build_variants.sh "<command>" "<-Dflags of variant 1>" "<-Dflags of variant 2>" ...
set -e
cd "$(dirname "$0")/.."
cmd="$1"; shift
BASE="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
for v in "$@"; do
   echo "== $v"
   touch isplib_amd/csrc/spmm_sweep.hip
   make -s -C isplib_amd/csrc HIPFLAGS="$BASE $v" all > /dev/null
   bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
