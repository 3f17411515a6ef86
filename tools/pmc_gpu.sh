#!/bin/bash
# usage: tools/pmc_gpu.sh <tag> "<counters>" <python script + args>   (run on the GPU box)
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace -d $OUT -o run -- python $REPO/"$@" > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
f=glob.glob("$OUT/*.db")[0]
con=sqlite3.connect(f)
rows=con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
import collections
d=collections.defaultdict(dict)
for k,c,n,a in rows:
    k=k.replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    d[k][c]=a; d[k]["n"]=n
dur={k.replace("(anonymous namespace)::","").replace("void ","").split("(")[0]:a for k,a in con.execute("select name, avg(duration) from kernels group by name")}
for k,v in d.items():
    if k.startswith("k_synth") or k.startswith("__amd"): continue
    print(k, "n=%d"%v.pop("n"), "avg_us=%.1f"%(dur.get(k,0)/1e3), " ".join(f"{c}={x:.4g}" for c,x in sorted(v.items())))
PY
