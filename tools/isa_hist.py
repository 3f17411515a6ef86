#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps gfx950 .s file.
usage: isa_hist.py file.s <substring of mangled name> [topN]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for m in re.finditer(r"^(_Z\S+):.*\n", s, re.M):
    name = m.group(1)
    if key not in name:
        continue
    end = s.find(".end_amdhsa_kernel", m.end())
    endp = s.find("s_endpgm", m.end())
    body = s[m.end():endp]
    ins = []
    for l in body.split("\n"):
        l = l.strip()
        if not l or l[0] in ".;_" or l.endswith(":"):
            continue
        ins.append(l.split()[0])
    c = collections.Counter(ins)
    print(name, "total", len(ins))
    for k, v in c.most_common(top):
        print(f"  {k:30s}{v}")
