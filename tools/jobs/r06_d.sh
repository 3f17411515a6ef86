#!/bin/bash
# round 6: the FED k_inflate launch (DMA batches feed a running kernel; Adler-32 + unpack fused into it): ingest tests, then
# fed vs unfed at T = 2920 / 8760, batch sizes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_d
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "ingest tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head
for T in 2920 8760; do
F=/tmp/c$T.nc
timeout 400 python tools/bench_ingest.py --T $T --quick --keep $F > $OUT/fed_$T.log 2>&1
ATLITE_HIP_INGEST_FED=0 timeout 100 python tools/bench_ingest.py --T $T --quick --no-host --keep $F > $OUT/unfed_$T.log 2>&1
for b in 32 64 256; do ATLITE_HIP_INGEST_BATCH=$b timeout 100 python tools/bench_ingest.py --T $T --quick --no-host --keep $F > $OUT/fed_b${b}_$T.log 2>&1; done
timeout 100 python tools/bench_ingest.py --T $T --quick --no-host --keep $F > $OUT/fed2_$T.log 2>&1
rm -f $F
done
for f in $OUT/*fed*.log; do echo "== $(basename $f)"; grep "DEVICE\|stage split\|sha1\|Error\|error\|host threads" $f | cut -c1-400; done
