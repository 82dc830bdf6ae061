"""Register / scratch / LDS usage per kernel of one HIP source: python scripts/reg_usage.py <file.hip> [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-c", "--cuda-device-only",
                      src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} " + " ".join(f"{a}={b}" for a, b in v.items()))
