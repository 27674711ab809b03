"""Register / scratch / occupancy of the kernels in one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

usage: kernel_resources.py ssn_f32.hip [name-filter] [extra hipcc flags...]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+) \[", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r":\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        short = re.sub(r"\(.*", "", name)
        print("%-60s VGPR %3d AGPR %3d scratch %4d occ %d  spill v%d s%d LDS %d" % (
            short[:60], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1),
            r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("LDS Size", -1)))
