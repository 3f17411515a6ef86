#!/usr/bin/env python3
"""Basic-block breakdown of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only): per block the instruction
mix (VALU / SALU / global loads / v_readlane+v_writelane = SGPR spills / LDS / waits) and where it branches - enough to
find a kernel's hot loop and see what it is made of.   usage: isa_blocks.py file.s <substring of mangled name>"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r"^(_Z\S+):.*\n", s, re.M):
    if key not in m.group(1):
        continue
    body = s[m.end():s.find("s_endpgm", m.end())]
    blocks, cur = [], ("entry", [])
    for l in body.split("\n"):
        l = l.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):", l)
        if lab:
            blocks.append(cur)
            cur = (lab.group(1), [])
            continue
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        cur[1].append(l)
    blocks.append(cur)
    print(m.group(1))
    for name, ins in blocks:
        ops = [i.split()[0] for i in ins]
        c = lambda p: sum(o.startswith(p) for o in ops)  # noqa: E731
        br = [i.split()[-1] for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
        print(f"{name:12s} n={len(ins):4d} valu={c('v_'):4d} f64={sum(('f64' in o) for o in ops):4d} salu={c('s_'):4d} gload={c('global_load'):2d} "
              f"gstore={c('global_store'):2d} lanes={c('v_readlane') + c('v_writelane'):3d} ds={c('ds_'):2d} waits={ops.count('s_waitcnt')} -> {' '.join(br)}")
