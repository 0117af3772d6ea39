#!/bin/bash
# usage: prof_pmc.sh <outdir> <cmd...>   -- separate --pmc passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
out=$(realpath -m "$1"); shift
mkdir -p "$out"
args=()
for a in "$@"; do if [ -f "$a" ]; then args+=("$(realpath "$a")"); else args+=("$a"); fi; done   # we cd away below
set -- "${args[@]}"
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  echo "pass $i: $ctrs"
  timeout -k 10 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; }
done
# gather all counter csvs
find $out -name "*counter_collection.csv" | head -20
