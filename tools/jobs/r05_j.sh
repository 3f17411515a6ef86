#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_j
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for q in 4 16; do
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/q$q.log 2>&1
echo "GPU_MAX_HW_QUEUES=$q"; grep "pv from FILE (inflate on the DEVICE\|stage split" $OUT/q$q.log | cut -c1-230
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $REPO/tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/trace.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob
f = glob.glob("gpurun_out/r05_j/trace/**/*.db", recursive=True)
print(f)
if f:
    db = sqlite3.connect(f[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    st = [t for t in tabs if t.startswith("rocpd_string")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kt})")]
    print(cols)
    ks = [t for t in tabs if "kernel_symbol" in t]
    q = f"select k.start, k.end, k.queue_id, k.stream_id, s.kernel_name from {kt} k join {ks[0]} s on k.kernel_id = s.id order by k.start"
    rows = [r for r in db.execute(q) if "k_inflate" in r[4]]
    t0 = rows[0][0]
    for r in rows[-14:]:
        print(f"start {(r[0]-t0)/1e6:9.2f} ms  dur {(r[1]-r[0])/1e6:8.2f} ms  queue {r[2]} stream {r[3]}")
PY
