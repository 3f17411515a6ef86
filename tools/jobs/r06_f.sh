#!/bin/bash
# round 6: what saturates k_inflate at 8 waves per SIMD?  SQ counters of ONE unfed launch over 8 176 streams (T = 7008, all resident)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_f
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp ATLITE_HIP_INGEST_FED=0
F=/tmp/c7008.nc
timeout 300 python tools/bench_ingest.py --T 7008 --quick --no-host --keep $F > $OUT/plain.log 2>&1
grep "stage split" $OUT/plain.log | cut -c1-300
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_CYCLES SQ_ACTIVE_INST_VMEM"
P4="SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o run -- python $REPO/tools/bench_ingest.py --T 7008 --quick --no-host --keep $F > $OUT/p$i.log 2>&1)
done
rm -f $F
python - <<PY
import sqlite3, glob, collections
for i in (1,2,3,4):
    fs = glob.glob("$OUT/p%d/**/*.db" % i, recursive=True)
    if not fs:
        print("pass", i, "no db"); continue
    con = sqlite3.connect(fs[0])
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_inflate%' group by kernel_name, counter_name").fetchall()
    for k, c, n, a in rows:
        print("pass", i, c, "n=%d" % n, "avg=%.6g" % a)
    for k, a, n in con.execute("select name, avg(duration), count(*) from kernels where name like '%k_inflate%' group by name"):
        print("pass", i, "k_inflate avg_us=%.1f n=%d" % (a / 1e3, n))
PY
# ---- the overlap, seen by rocprofv3: one fed read of a C2 year, kernel + memory-copy trace (no counters)
unset ATLITE_HIP_INGEST_FED
F=/tmp/c8760.nc
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o run -- python $REPO/tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/trace.log 2>&1)
rm -f $F
grep "DEVICE\|stage split" $OUT/trace.log | cut -c1-300
python - <<PY
import sqlite3, glob
fs = glob.glob("$OUT/trace/**/*.db", recursive=True)
print(fs)
if fs:
    db = sqlite3.connect(fs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    print([t for t in tabs if 'copy' in t.lower() or 'kernel' in t.lower()][:20])
    ks = db.execute("select name, start, end from kernels where name like '%k_inflate%' order by start").fetchall()
    print("k_inflate launches:", len(ks))
    if ks:
        name, s0, e0 = ks[-1]
        print("last k_inflate: %.1f ms" % ((e0 - s0) / 1e6))
        try:
            cps = db.execute("select start, end, size, name from memory_copies where start >= ? and start <= ? order by start", (s0 - 5_000_000, e0)).fetchall()
        except Exception as ex:
            print("memory_copies view:", ex); cps = []
        big = [c for c in cps if c[2] and c[2] > (1 << 20)]
        print("DMAs of > 1 MiB inside the launch: %d, first starts %.1f ms after the kernel, last ends at %.1f ms (kernel ends at %.1f)" % (
            len(big), (big[0][0] - s0) / 1e6 if big else -1, (big[-1][1] - s0) / 1e6 if big else -1, (e0 - s0) / 1e6))
        for c in big[:3] + big[-3:]:
            print("   DMA %.1f .. %.1f ms, %.1f MB, %.1f GB/s  %s" % ((c[0] - s0) / 1e6, (c[1] - s0) / 1e6, c[2] / 1e6, c[2] / max(c[1] - c[0], 1), c[3]))
        others = db.execute("select name, start, end from kernels where start >= ? and start <= ? and name not like '%k_inflate%' order by start", (s0, e0 + 10_000_000)).fetchall()
        for n, s, e in others[:8]:
            print("   kernel %-60s %.1f .. %.1f ms" % (n[:60], (s - s0) / 1e6, (e - s0) / 1e6))
PY
