#!/usr/bin/env python3
"""Per-kernel resource summary of the HIP library (VGPRs, AGPRs, spills, scratch, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage.  usage: tools/kres.py [name-filter] [extra hipcc flags ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", "-I", os.path.join(root, "include"),
       os.path.join(root, "t-svgp_amd/csrc/tsvgp_kernels.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.split("\n"):
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    txt = m.group(1)
    if txt.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", txt.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
    elif ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and flt in cur.get("name", ""):
            n = re.sub(r"\(anonymous namespace\)::", "", cur["name"])[:70]
            print(f"{n:70s} vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} spill {cur.get('VGPR Spill', '?'):>3} "
                  f"scratch {cur.get('ScratchSize [bytes/lane]'):>5} occ {cur.get('Occupancy [waves/SIMD]')} lds {cur.get('LDS Size [bytes/block]')}")
