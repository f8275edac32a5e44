"""Dev tool: VGPRs / spills / scratch of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py bayesian-coresets_amd/csrc/proj.hip [filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "--offload-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
bad = 0
for mangled, name in zip(rows, names):
    if flt and flt not in name:
        continue
    r = rows[mangled]
    spill = r.get("VGPRs Spill", 0)
    scratch = r.get("ScratchSize [bytes/lane]", 0)
    bad += (spill > 0) + (scratch > 0)
    print("%-64s VGPRs %3d AGPRs %3d spill %3d scratch %4d B  SGPRs %3d  occupancy %d" % (
        name[:64], r.get("VGPRs", -1), r.get("AGPRs", 0), spill, scratch, r.get("SGPRs", -1), r.get("Occupancy [waves/SIMD]", -1)))
print("kernels with spills or scratch: %d" % bad)
