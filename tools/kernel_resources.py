#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re
import subprocess
import sys

import glob

paths = [sys.argv[1]] if len(sys.argv) > 1 else sorted(glob.glob("atlite_amd/csrc/*.resource.txt"))
txt = "".join(open(p).read() for p in paths)
pat = re.compile(
    r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)"
    r".*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
for name, sg, v, a, s, o, l in pat.findall(txt):
    try:
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        d = name
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)
    print(f"{d:60s} sgpr={sg:>3} vgpr={v:>3} agpr={a} scratch={s} occ={o} lds={l}")
