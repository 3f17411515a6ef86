#!/bin/bash
# round 6: what does residency buy?  ONE k_inflate launch over a whole C2 year (10 220 streams) at 8 / 6 / 5 / 4 waves per SIMD
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_b
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
F=/tmp/c8760.nc
V=$REPO/atlite_amd/lib/variants
export ATLITE_HIP_SLAB_BYTES=$((8<<30)) ATLITE_HIP_INGEST_JOB=100000
timeout 400 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w8.log 2>&1
for w in 6 5 4; do ATLITE_HIP_LIB=$V/lib_w$w.so timeout 100 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w$w.log 2>&1; done
timeout 100 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w8b.log 2>&1
rm -f $F
for f in $OUT/w*.log; do echo "== $(basename $f)"; grep "DEVICE\|stage split\|sha1\|Error\|error" $f | cut -c1-400; done
