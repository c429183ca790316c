"""Register / scratch usage of every kernel of a HIP source (device-only compile with -Rpass-analysis=kernel-resource-usage).
Usage: python scripts/kernel_resources.py [filter-substring] [source] [extra hipcc flags ...]"""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
src = sys.argv[2] if len(sys.argv) > 2 else "towr_amd/csrc/kernels.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: \s*(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", name)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if flt in name:
        print("%-58s VGPR %3s AGPR %3s SGPR %3s  spill S %3s V %3s  scratch %4s B  occ %s" % (
            name[-58:], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
