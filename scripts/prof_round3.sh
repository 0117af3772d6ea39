#!/bin/bash
# Round-3 profiling session on the GPU box: rocprofv3 kernel trace + separate PMC passes for the headline, for the other
# configurations of the `extra` array (bench.py --only <key>) and for the hybrid kernel.  usage: prof_round3.sh <part> [outdir]
# part: headline | reddit | products | hybrid.  Raw output under gpurun_out/ (scratch); the summaries are copied to profiles/.
part="$1"; out="${2:-gpurun_out/r3/prof}"
mkdir -p "$out"
root="$(cd "$(dirname "$0")/.." && pwd)"
export PMC_TIMEOUT=400
groups="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum;GRBM_GUI_ACTIVE GRBM_TA_BUSY"
py="$(command -v python3)"
kt() {   # kt <name> <args...>: kernel trace + stats of `python3 <args>`
   local name="$1"; shift
   ( cd /tmp && TMPDIR=/tmp timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/kt_$name" -- "$py" "$@" \
       > "$root/$out/kt_$name.out" 2> "$root/$out/kt_$name.err" )
   find "$root/$out/kt_$name" -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} "$root/$out/kt_$name.kernel_stats.csv"
   echo "kernel trace $name done"
}
pmc() {  # pmc <name> <counter groups> <args...>
   local name="$1" g="$2"; shift 2
   PMC_ONLY="$g" "$root/scripts/prof_pmc.sh" "$root/$out/pmc_$name" "$py" "$@" > "$root/$out/pmc_$name.log" 2>&1
   "$py" "$root/scripts/pmc_summary.py" "$root/$out/pmc_$name" > "$root/$out/pmc_$name.summary.txt" 2>&1
   echo "pmc $name done"
}
case "$part" in
  headline)
    kt bench "$root/bench.py" --no-cpu-baseline --no-extra
    pmc bench "$groups;SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "$root/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-backward --no-extra
    ;;
  reddit)
    for key in reddit-max-k64-weighted reddit-sum-k128-weighted reddit-sddmm-k128; do
      kt "$key" "$root/bench.py" --only "$key"
      pmc "$key" "$groups" "$root/bench.py" --only "$key"
    done
    ;;
  products)
    for key in products-chunglu-sum-k256-plain products-sbm-sum-k256-plain products-sbm-sum-k256-ordered; do
      kt "$key" "$root/bench.py" --only "$key"
      pmc "$key" "$groups" "$root/bench.py" --only "$key"
    done
    ;;
  hybrid)
    export HYB=4:31
    kt hybrid "$root/scripts/exp_hybrid.py" 128
    pmc hybrid "$groups;SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "$root/scripts/exp_hybrid.py" 128
    ;;
  *) echo "usage: prof_round3.sh headline|reddit|products|hybrid [outdir]" >&2; exit 2;;
esac
ls "$root/$out" | head -50
