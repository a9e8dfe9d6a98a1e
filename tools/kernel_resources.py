#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy / LDS per kernel from hipcc's -Rpass-analysis=kernel-resource-usage."""
import re
import subprocess
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(root, "ipp-rl_amd", "csrc"), "resource-usage"], capture_output=True, text=True)
txt = out.stdout + out.stderr
cur = {}
rows = []
for line in txt.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, val = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": val}
        rows.append(cur)
    else:
        cur["Spill" if k == "VGPRs Spill" else k.split(" ")[0]] = val
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:45s} VGPR={r.get('VGPRs'):>4s} AGPR={r.get('AGPRs'):>3s} SGPR={r.get('TotalSGPRs'):>4s} scratch={r.get('ScratchSize'):>5s} occ={r.get('Occupancy')} spill={r.get('Spill')} LDS={r.get('LDS')}")
