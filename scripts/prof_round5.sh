#!/bin/bash
# Round-5 profiling session on the GPU box: rocprofv3 kernel trace + separate PMC passes (one counter group per pass, never
# mixed with a trace domain) for the headline and for what round 5 added or changed (bench.py --only <key>).
# usage: prof_round5.sh <part> [outdir]     part: headline | fusedmm | gcn | products | emulated
# Raw output under gpurun_out/ (scratch); scripts/make_traffic_json_r05.py + a copy put the summaries into profiles/.
part="$1"; out="${2:-gpurun_out/r5/prof}"
mkdir -p "$out"
root="$(cd "$(dirname "$0")/.." && pwd)"
export PMC_TIMEOUT=400
groups="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;GRBM_GUI_ACTIVE GRBM_TA_BUSY"
sq="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
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
    pmc bench "$groups;$sq" "$root/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-backward --no-extra
    ;;
  fusedmm)
    for key in reddit-fusedmm-sigmoid-k128 reddit-fusedmm-tdist-k128; do
      kt "$key" "$root/bench.py" --only "$key"
      g="$groups"; [ "$key" = reddit-fusedmm-sigmoid-k128 ] && g="$groups;$sq"
      pmc "$key" "$g" "$root/bench.py" --only "$key"
    done
    ;;
  gcn)
    kt gcn-epoch "$root/bench.py" --only gcn-epoch
    kt gcn-epoch-normalized "$root/scripts/gcn_epoch.py" --normalize --epochs 6
    ;;
  products)
    for key in ${KEYS:-products-chunglu-sum-k256-plain products-sbm-sum-k256-plain products-sbm-sum-k256-ordered}; do
      kt "$key" "$root/bench.py" --only "$key"
      pmc "$key" "$groups" "$root/bench.py" --only "$key"
    done
    ;;
  emulated)
    kt scaling-emulated-reddit "$root/bench.py" --only scaling-emulated-reddit
    ;;
  *) echo "usage: prof_round5.sh headline|fusedmm|gcn|products|emulated [outdir]" >&2; exit 2;;
esac
ls "$root/$out" | head -80
