#!/bin/bash
# Rehearsal of bench.py's N > 1 path on ONE GPU over gloo (ISPLIB_BENCH_BACKEND=gloo: the ranks share the card; numbers
# mean nothing, the code path and the failure containment are what it checks).  A scaled graph, so that the point-to-point
# `direct xB` schedules run inside bench.py too (over gloo a full-size shard takes seconds per peer).
#   1. clean run: every schedule validated, the line printed
#   2. a local kernel failure inside an optional schedule on one rank: dropped on every rank, the rest goes on
#   3. a rank that never comes back from an optional schedule: the measured result is printed, exit 0
#   4. a rank that raises before a schedule's first collective (its peers wait in it): the same
#   5. a rank that hangs before any result exists: non-zero exit inside the deadline, no line
# usage: rehearse_multirank.sh [ranks=4] [outdir=gpurun_out/r4]
ranks="${1:-4}"; out="${2:-gpurun_out/r4}"
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/$out"
export ISPLIB_BENCH_BACKEND=gloo ISPLIB_BENCH_NO_RETRY=1
common=(--gpus "$ranks" --scale 0.25 --steps 5 --warmup 2 --no-extra)
run() {   # run <name> <env assignments...>
   local name="$1"; shift
   local t0=$SECONDS
   env "$@" timeout -k 10 600 python3 "$root/bench.py" "${common[@]}" > "$root/$out/rehearsal_$name.json" 2> "$root/$out/rehearsal_$name.log"
   local rc=$?
   echo "== $name: exit $rc after $((SECONDS - t0)) s, $(grep -c '"metric"' "$root/$out/rehearsal_$name.json") JSON line(s)" | tee -a "$root/$out/rehearsal_summary.txt"
   grep -E "dropped|abandoned|deadline|-> |north_star|out of step|raised" "$root/$out/rehearsal_$name.log" | head -12 | tee -a "$root/$out/rehearsal_summary.txt"
   python3 - "$root/$out/rehearsal_$name.json" <<'PY' | tee -a "$root/$out/rehearsal_summary.txt"
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        r = json.loads(ln)
        print("   line:", {k: r.get(k) for k in ("n_gpus", "ms_per_step", "candidates_ms", "abandoned")}, r["config"]["partition"])
PY
}
: > "$root/$out/rehearsal_summary.txt"
run clean ISPLIB_BENCH_T_CANDIDATE=150 ISPLIB_BENCH_DEADLINE=900
run kernel_fault "ISPLIB_BENCH_INJECT=kernel:overlapped sliced:1" ISPLIB_BENCH_T_CANDIDATE=150 ISPLIB_BENCH_DEADLINE=900
run hang_optional "ISPLIB_BENCH_INJECT=hang:pipelined x2:2" ISPLIB_BENCH_T_CANDIDATE=40
run raise_optional "ISPLIB_BENCH_INJECT=raise:overlapped sliced:3" ISPLIB_BENCH_T_CANDIDATE=40
run hang_before_result "ISPLIB_BENCH_INJECT=hang:north_star:1" ISPLIB_BENCH_T_SAFE=60
