"""Per-kernel register / LDS / scratch usage of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel.  usage (from the repo root): python tools/kernel_resources.py n-body_amd/csrc/direct_sym.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden",
       "-Iinclude", "-In-body_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
for r in rows:
    if flt and flt not in r["name"]:
        continue
    name = re.sub(r"\(.*", "", r["name"]).replace("nbh::", "").replace("void ", "")
    print(f"{name:60s} vgpr {r.get('VGPRs', 0):3d} agpr {r.get('AGPRs', 0):3d} sgpr {r.get('TotalSGPRs', 0):3d} "
          f"scratch {r.get('ScratchSize', 0):4d} lds {r.get('LDS Size', 0):6d} occ {r.get('Occupancy', 0)}")
