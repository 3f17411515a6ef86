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
