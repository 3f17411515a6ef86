#!/bin/bash
# round 6: kernel trace of one from-file read of atlite's own chunking ((100, y, x): 16 MB streams) through the segment scheme
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_y
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
F=/tmp/large.nc
timeout 900 python tools/bench_ingest.py --T ${1:-2000} --quick --no-host --chunks 100,200,200 --keep $F > $OUT/large.log 2>&1
grep "^wrote\|DEVICE\|launch" $OUT/large.log | cut -c1-330
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o split -- python $REPO/tools/bench_ingest.py --T ${1:-2000} --quick --no-host --chunks 100,200,200 --keep $F > $OUT/prof.log 2>&1
cd $REPO
S=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && head -12 $S | cut -c1-200
