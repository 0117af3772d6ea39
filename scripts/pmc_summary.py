"""Summarise rocprofv3 counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "?")
        short = name.split("(")[0][-60:]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, ctrs in acc.items():
    if not any(s in kname for s in ("spmm", "sddmm", "minmax", "combine", "slices", "finalize", "colptr", "hub_fold", "stream", "sweep", "gather_kernel")):
        continue
    print(kname)
    for c, v in sorted(ctrs.items()):
        # skip warmup dispatches: use the median
        v = sorted(v)
        print(f"   {c:40s} n={len(v):3d} median={v[len(v)//2]:.6g}")
