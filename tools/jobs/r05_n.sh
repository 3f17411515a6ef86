#!/bin/bash
# k_inflate A/B on one box, one file: HEAD~ library (variants/libatlite_hip_base.so) vs the tree's (window words stored before
# the next window's load is issued; resolve inlined; short matches 8 bytes per round trip; branch-free scan tail), then the
# ingest tests on the tree's library
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_n
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
F=/tmp/c2920.nc
BASE=$REPO/atlite_amd/lib/variants/libatlite_hip_base.so
ATLITE_HIP_LIB=$BASE timeout 70 python tools/bench_ingest.py --T 2920 --quick --keep $F > $OUT/base1.log 2>&1
timeout 30 python tools/bench_ingest.py --T 2920 --quick --keep $F > $OUT/new1.log 2>&1
ATLITE_HIP_LIB=$BASE timeout 30 python tools/bench_ingest.py --T 2920 --quick --keep $F > $OUT/base2.log 2>&1
timeout 30 python tools/bench_ingest.py --T 2920 --quick --keep $F > $OUT/new2.log 2>&1
rm -f $F
for f in base1 new1 base2 new2; do echo "== $f"; grep "DEVICE\|stage split\|identical\|Error\|error" $OUT/$f.log | cut -c1-330; done
timeout 60 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"
