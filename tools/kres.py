#!/usr/bin/env python3
"""Developer tool (no GPU): compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one
line per kernel: VGPRs (AGPRs), SGPR / VGPR spills, scratch, LDS, occupancy.

    python tools/kres.py conv_fast [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I" + ROOT + "/include",
       "-I" + ROOT + "/serenade_amd/csrc", "-c", f"{ROOT}/serenade_amd/csrc/{src}.hip", "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\d+)", ln)
    if m and cur:
        cur[m.group(1).strip()] = int(m.group(2))
        if m.group(1).strip().startswith("LDS Size"):
            n = cur["name"].replace("(anonymous namespace)::", "")
            if flt in n:
                print(f"{n[:110]:110s} vgpr {cur.get('VGPRs', -1):3d} agpr {cur.get('AGPRs', -1):3d} sgpr "
                      f"{cur.get('TotalSGPRs', -1):3d} spill s/v {cur.get('SGPRs Spill', -1)}/{cur.get('VGPRs Spill', -1)} "
                      f"scratch {cur.get('ScratchSize', -1)} occ {cur.get('Occupancy', -1)} lds {cur.get('LDS Size', -1)}")
