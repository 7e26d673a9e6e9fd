#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of the gfx950 code objects (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [substring ...]   (prints only kernels whose demangled name contains a substring)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "juliachem.jl_amd", "csrc", "jcdf_api.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"),
       src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[1:] if a.startswith("-D")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
want = [a for a in sys.argv[1:] if not a.startswith("-D")]
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-60s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "spillV", "SGPR", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("jcdf::", "").replace("void ", "")
    if want and not any(w in n for w in want):
        continue
    print("%-60s %5s %5s %6s %6s %4s %7s" % (n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("TotalSGPRs"),
                                           r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
