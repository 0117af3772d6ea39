"""Static check of a -save-temps gfx950 listing: per kernel, how many `s_waitcnt vmcnt(0)` sit inside loop blocks (a
vmcnt(0) in a gather loop is a drained pipeline per iteration), allocated VGPRs, scratch.  usage: isa_waits.py file.s [name filter]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
starts = [(i, ln.split(":")[0]) for i, ln in enumerate(lines) if ln.startswith("_ZN6isplib") and ":" in ln and flt in ln.split(":")[0]]
for idx, (i, name) in enumerate(starts):
    j = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    seg = lines[i:j]
    end = next((k for k, ln in enumerate(seg) if "s_endpgm" in ln), len(seg))
    in_loop, hits, cur = 0, [], False
    for k, ln in enumerate(seg[:end]):
        if ln.startswith(".LBB"):
            cur = "Loop" in ln or (k + 1 < end and "Loop" in seg[k + 1])
        elif ln.startswith("; %bb"):
            cur = "Loop" in ln
        if "s_waitcnt vmcnt(0)" in ln and cur:
            in_loop += 1
            hits.append(k)
    vg = [ln.split()[-1] for ln in seg if "next_free_vgpr" in ln]
    sc = [ln.split()[-1] for ln in seg if "private_segment_fixed_size" in ln]
    lds = [ln.split()[-1] for ln in seg if "group_segment_fixed_size" in ln]
    short = re.sub(r"EvNS_9SweepArgsE$", "", name).replace("_ZN6isplib", "")
    print(f"{short:62s} vmcnt(0) in loops: {in_loop:2d} {hits[:6]}  vgpr {vg[0] if vg else '?'} scratch {sc[0] if sc else '?'} lds {lds[0] if lds else '?'}")
