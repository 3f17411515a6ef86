#!/bin/bash
# round 6: a stream's time against residency and register budget - ONE k_inflate launch of 4 088 streams (T = 3504: one round for
# every build) and of 8 176 (T = 7008: one round at 8 waves per SIMD, two below)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_c
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
V=$REPO/atlite_amd/lib/variants
export ATLITE_HIP_SLAB_BYTES=$((8<<30)) ATLITE_HIP_INGEST_JOB=100000
for T in 3504 7008; do
F=/tmp/c$T.nc
timeout 400 python tools/bench_ingest.py --T $T --quick --no-host --keep $F > $OUT/w8_$T.log 2>&1
for w in 6 5 4; do ATLITE_HIP_LIB=$V/lib_w$w.so timeout 100 python tools/bench_ingest.py --T $T --quick --no-host --keep $F > $OUT/w${w}_$T.log 2>&1; done
rm -f $F
done
for f in $OUT/w*.log; do echo "== $(basename $f)"; grep "stage split\|Error\|error" $f | cut -c60-330; done
