#!/usr/bin/env python3
"""Compile the HIP sources with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re, subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = [os.path.join(REPO, "graph-and-sequential-recommendation-systems_amd/csrc", f) for f in os.listdir(os.path.join(REPO, "graph-and-sequential-recommendation-systems_amd/csrc")) if f.endswith(".hip")]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
extra = sys.argv[2:]
out = ""
for f in src:
    out += subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-c", "--cuda-device-only",
                           "-Rpass-analysis=kernel-resource-usage", *extra, f, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    if "error" in line: print(line)
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and flt in cur["name"]:
            print(f"{cur['name']:60s} sgpr {cur.get('SGPRs','?'):>4s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>3s} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?')} lds {v.strip()}")
