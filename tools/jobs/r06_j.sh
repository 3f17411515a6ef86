#!/bin/bash
# round 6: four-elements-per-lane unpack inside k_inflate: ingest tests, a C2 year fed / unfed, a third of a year
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_j
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export ATLITE_HIP_INGEST_TIMEOUT_MS=3000 ATLITE_HIP_INGEST_DEBUG=1
timeout 300 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "ingest tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error|ingest\]" $OUT/tests.log | head
F=/tmp/c8760.nc
timeout 400 python tools/bench_ingest.py --T 8760 --quick --keep $F > $OUT/fed.log 2>&1
timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/fed2.log 2>&1
ATLITE_HIP_INGEST_FED=0 timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/unfed.log 2>&1
ATLITE_HIP_INGEST_BATCH=256 timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/fed_b256.log 2>&1
rm -f $F
F=/tmp/c2920.nc
timeout 200 python tools/bench_ingest.py --T 2920 --quick --no-host --keep $F > $OUT/fed_2920.log 2>&1
ATLITE_HIP_INGEST_FED=0 timeout 200 python tools/bench_ingest.py --T 2920 --quick --no-host --keep $F > $OUT/unfed_2920.log 2>&1
rm -f $F
for f in $OUT/fed*.log $OUT/unfed*.log; do echo "== $(basename $f)"; grep "DEVICE\|stage split\|sha1\|Error\|error\|host threads" $f | cut -c1-400; done
