"""profiles/traffic.json entries of round 5 from the PMC summaries of scripts/prof_round5.sh (scripts/pmc_summary.py output):
fabric-side bytes per launch = sum over the launch's kernels of dispatches x (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE
counts 128-byte requests at 64 bytes on gfx950: the guide's correction), TCC hit rate, TA busy share.
usage: make_traffic_json.py <prof dir> <profiles dir>"""
import json
import os
import re
import sys

prof, dest = sys.argv[1], sys.argv[2]


def summary(path):
    out, cur = {}, None
    for ln in open(path):
        if not ln.startswith(" "):
            cur = ln.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+median=(\S+)", ln)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(3))
    return out


# key -> (summary file stem, kernels of one launch: name fragment -> dispatches per launch), round 4
launch = {
    "reddit-sum-k128-stream31-r05": ("pmc_bench", {"spmm_stream_kernel<16, false": 4, "sweep_hub_fold_kernel<0": 2}),
    "reddit-fusedmm-sigmoid-k128": ("pmc_reddit-fusedmm-sigmoid-k128", {"fusedmm_stream_kernel<1, 32": 8, "sweep_hub_fold_kernel<0": 1}),
    "reddit-fusedmm-tdist-k128": ("pmc_reddit-fusedmm-tdist-k128", {"fusedmm_stream_kernel<2, 32": 8, "sweep_hub_fold_kernel<0": 1}),
    "products-chunglu-sum-k256-plain": ("pmc_products-chunglu-sum-k256-plain", {"spmm_csr_kernel": 1}),
    "products-sbm-sum-k256-plain": ("pmc_products-sbm-sum-k256-plain", {"spmm_csr_kernel": 1}),
    "products-sbm-sum-k256-ordered": ("pmc_products-sbm-sum-k256-ordered", {"spmm_csr_kernel": 1}),
}
ROUND = os.environ.get("PROF_ROUND", "r05")
tpath = os.path.join(dest, "traffic.json")
table = json.load(open(tpath))
for key, (name, kernels) in launch.items():
    path = os.path.join(prof, name + ".summary.txt")
    if not os.path.exists(path):
        print("missing", path)
        continue
    s = summary(path)
    total, rec, main = 0.0, {}, None
    for frag, count in kernels.items():
        hit = [k for k in s if frag in k]
        if not hit:
            continue
        c = s[hit[0]]
        b = count * (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        total += b
        rec[hit[0].replace("void isplib::", "") + "_per_launch"] = int(b)
        if main is None:
            main = c
    rec_out = {"fabric_bytes_per_launch": int(total), "dispatches_per_launch": sum(kernels.values())}
    rec_out.update(rec)
    if main:
        if main.get("TCC_REQ_sum"):
            rec_out["tcc_hit_rate"] = main["TCC_HIT_sum"] / (main["TCC_HIT_sum"] + main["TCC_MISS_sum"])
            rec_out["tcc_requests_128B_per_dispatch"] = main["TCC_REQ_sum"]
        if main.get("GRBM_GUI_ACTIVE"):
            rec_out["grbm_ta_busy_over_gui_active"] = main["GRBM_TA_BUSY"] / main["GRBM_GUI_ACTIVE"]
    rec_out["source"] = f"profiles/{ROUND}_{name}_summary.txt (scripts/prof_round5.sh: one rocprofv3 --pmc group per pass; FETCH_SIZE doubled for gfx950); bytes leaving the XCD L2s, Infinity-Cache hits included"
    if key == "reddit-sum-k128-stream31-r05":
        table["reddit-sum-k128-stream31"] = rec_out          # what bench.py's headline line looks up
    table[key] = rec_out
    print(key, f"{total / 1e9:.2f} GB", rec_out.get("tcc_hit_rate"))
json.dump(table, open(tpath, "w"), indent=1)
