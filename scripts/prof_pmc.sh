#!/bin/bash
# usage: prof_pmc.sh <outdir> <program> [args...]   -- separate --pmc passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
#
# <program> must be the real thing to profile: `python3 script.py ...`, `script.py ...` (python3 is put in front) or
# an ELF binary.  With --pmc the profiler's preloaded library initialises the GPU before the program starts, and
# on this pool a process that has initialised the GPU must never exec another program (it takes the machine down):
# env, taskset, numactl, bash -c / sh -c, torchrun and `#!/usr/bin/env` scripts run directly are all such an exec
# hop, so they are refused here.
out=$(realpath -m "$1"); shift
if [ $# -lt 1 ]; then echo "usage: prof_pmc.sh <outdir> <program> [args...]" >&2; exit 2; fi
case "$(basename "$1")" in
  env|taskset|numactl|bash|sh|dash|zsh|torchrun|nohup|timeout|stdbuf|time|xargs|sudo)
    echo "prof_pmc.sh: refusing '$1': a launcher between rocprofv3 and the program is an exec after GPU initialisation" >&2; exit 2;;
esac
if [[ "$1" == *.py ]]; then set -- "$(command -v python3)" "$@"; fi
prog=$(command -v "$1" || true)
if [ -z "$prog" ]; then echo "prof_pmc.sh: '$1' not found" >&2; exit 2; fi
# the ELF test reads the file behind the symlinks; the program is STARTED from the path it was found at (a venv's
# python3 is a symlink into the base installation: started from there it would not find its pyvenv.cfg / site-packages;
# executing a symlink to an ELF is not an extra exec hop)
real=$(readlink -f "$prog")
if [ "$(head -c 4 "$real" | od -An -c | tr -d ' ')" != '177ELF' ]; then
  echo "prof_pmc.sh: '$prog' is not an ELF binary (a script would be exec'd through its interpreter): name the interpreter" >&2; exit 2
fi
shift
args=("$prog")
for a in "$@"; do if [ -f "$a" ]; then args+=("$(realpath "$a")"); else args+=("$a"); fi; done   # we cd away below
set -- "${args[@]}"
mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
i=0
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "GRBM_GUI_ACTIVE GRBM_TA_BUSY")
if [ -n "$PMC_ONLY" ]; then IFS=';' read -r -a groups <<< "$PMC_ONLY"; fi     # e.g. PMC_ONLY="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;FETCH_SIZE"
for ctrs in "${groups[@]}"; do
  i=$((i+1))
  echo "pass $i: $ctrs"
  timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; }
done
find $out -name "*counter_collection.csv" | head -20
