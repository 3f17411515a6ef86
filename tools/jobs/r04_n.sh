#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_n
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $REPO/tools/bench_pitch.py 201 201 > $OUT/p.log 2>&1
python - > $OUT/summary.txt 2>&1 <<'PY'
import glob, sqlite3
con = sqlite3.connect(glob.glob("/tmp/prof/**/*.db", recursive=True)[0])
rows = con.execute("select name, duration from kernels order by start").fetchall()
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
run = []
for n, d in rows:
    n = short(n)
    if run and run[-1][0] == n: run[-1][1].append(d / 1e3)
    else: run.append([n, [d / 1e3]])
# collapse alternating fused/combine sequences
out = []
for n, v in run:
    key = n
    if out and out[-1][0] == key: out[-1][1] += v
    elif len(out) > 1 and out[-2][0] == key and len(v) == 1: out[-2][1] += v
    else: out.append([key, list(v)])
for n, v in out:
    v2 = sorted(v)
    print(f"{n:90s} n={len(v):3d} median {v2[len(v2)//2]:9.1f} us  min {v2[0]:9.1f}")
PY
cat $OUT/summary.txt | head -60
