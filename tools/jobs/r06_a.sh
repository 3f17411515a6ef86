#!/bin/bash
# round 6, first job: 16-bit Huffman tables (4.7 kB of LDS per stream) + pipelined jobs: ingest tests, then k_inflate at
# 8 / 7 / 6 / 5 resident waves per SIMD (register budgets 64 / 72 / 80 / 96) on one box and one file, T = 2920 and 8760
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_a
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
hostname > $OUT/box.txt
timeout 300 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "ingest tests rc=$? $(tail -1 $OUT/tests.log)"
F=/tmp/c2920.nc
V=$REPO/atlite_amd/lib/variants
timeout 200 python tools/bench_ingest.py --T 2920 --quick --keep $F > $OUT/w8_2920.log 2>&1
for w in 7 6 5; do ATLITE_HIP_LIB=$V/lib_w$w.so timeout 60 python tools/bench_ingest.py --T 2920 --quick --no-host --keep $F > $OUT/w${w}_2920.log 2>&1; done
timeout 60 python tools/bench_ingest.py --T 2920 --quick --no-host --keep $F > $OUT/w8b_2920.log 2>&1
for j in 512 1024 4096 100000; do ATLITE_HIP_INGEST_JOB=$j timeout 60 python tools/bench_ingest.py --T 2920 --quick --no-host --keep $F > $OUT/w8_job${j}_2920.log 2>&1; done
rm -f $F
F=/tmp/c8760.nc
timeout 400 python tools/bench_ingest.py --T 8760 --quick --keep $F > $OUT/w8_8760.log 2>&1
for w in 7 6 5; do ATLITE_HIP_LIB=$V/lib_w$w.so timeout 100 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w${w}_8760.log 2>&1; done
for j in 1024 4096; do ATLITE_HIP_INGEST_JOB=$j timeout 100 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w8_job${j}_8760.log 2>&1; done
ATLITE_HIP_SLAB_BYTES=$((8<<30)) timeout 100 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/w8_oneslab_8760.log 2>&1
rm -f $F
for f in $OUT/w*.log; do echo "== $(basename $f)"; grep "DEVICE\|stage split\|identical\|sha1\|Error\|error\|host threads" $f | cut -c1-400; done
