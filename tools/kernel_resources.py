#!/usr/bin/env python3
"""Compile lft_api.hip for gfx950 and print per-kernel register / spill / occupancy figures."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
                      os.path.join(ROOT, "lft_amd/csrc/lft_api.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur: rows[cur][key] = int(m.group(1))
print(f"{'kernel':58s} VGPR AGPR spill scratch occ")
for k, r in rows.items():
    if "k_" not in k: continue
    name = k[:56]
    print(f"{name:58s} {r.get('VGPRs',0):4d} {r.get('AGPRs',0):4d} {r.get('VGPRs Spill',0):5d} {r.get('ScratchSize [bytes/lane]',0):7d} {r.get('Occupancy [waves/SIMD]',0):3d}")
