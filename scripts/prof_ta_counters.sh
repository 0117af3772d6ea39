#!/bin/bash
# Address-path counters (TA / TCP / TD / SQ-VMEM) of the gather-only microbenchmark beside the headline kernel and the plain kernel on
# config 4: what the stream kernel's 26 clocks per 1-KiB gather are made of, against the microbenchmark's 19-20.  Separate --pmc passes,
# no trace domain.  usage: prof_ta_counters.sh [outdir]
out="${1:-gpurun_out/r5/ta}"
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/$out"
py="$(command -v python3)"
export PMC_TIMEOUT=100     # a pass whose counter set the hardware refuses aborts at once but rocprofv3 then sits until it is killed
# at most two counters of a block per pass: six TA or TCP counters together are "beyond the capabilities of the hardware to collect"
groups="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum;TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
groups="$groups;TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum;TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum;TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
tlb="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
groups="$groups;$tlb;TD_TD_BUSY_sum TD_TC_STALL_sum;SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD;SQ_BUSY_CYCLES SQ_WAVE_CYCLES;GRBM_GUI_ACTIVE GRBM_TA_BUSY"
run() {  # run <name> <program> [args...]
   local name="$1"; shift
   PMC_ONLY="$groups" "$root/scripts/prof_pmc.sh" "$root/$out/pmc_$name" "$@" > "$root/$out/pmc_$name.log" 2>&1
   "$py" "$root/scripts/pmc_summary.py" "$root/$out/pmc_$name" > "$root/$out/pmc_$name.summary.txt" 2>&1
   echo "pmc $name done"
}
run ubench "$root/scripts/ubench/gather_paths" 4096 2048
run headline "$py" "$root/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-backward --no-extra
groups="$tlb;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
run products-chunglu "$py" "$root/bench.py" --only products-chunglu-sum-k256-plain
ls "$root/$out"
