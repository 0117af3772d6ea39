"""Per-kernel VGPR / occupancy / spill table from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/tmp/ru.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
    flag = " <-- SPILL" if r.get("VGPRs Spill", 0) or r.get("ScratchSize [bytes/lane]", 0) else ""
    print(f"{dem:92s} vgpr {r.get('VGPRs',0):4d} occ {r.get('Occupancy [waves/SIMD]',0)} spill {r.get('VGPRs Spill',0)} scratch {r.get('ScratchSize [bytes/lane]',0)}{flag}")
