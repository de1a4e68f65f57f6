"""Compact per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
    python tools/kernel_resources.py [mw_dycore.hip] [filter-substring ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0] if args and args[0].endswith((".hip", ".cpp")) else "mw_dycore.hip"
filt = [a for a in args if not a.endswith((".hip", ".cpp"))]
path = os.path.join(root, "miniweatherml_amd", "csrc", src)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=on", "-I/opt/rocm/include",
       "-x", "hip", "-c", path, "-o", "/tmp/rr/kr.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mw::", "")
    except Exception:
        return n
print("%-44s %5s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "occ", "LDS"))
for r in rows:
    n = dem(r["name"])
    if filt and not any(f in n for f in filt):
        continue
    print("%-44s %5s %5s %5s %6s %6s %4s %7s" % (n[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                             r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
