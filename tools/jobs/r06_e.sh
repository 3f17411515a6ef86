#!/bin/bash
# round 6: fed k_inflate, second try (copy stream on its own priority; dynamic-LDS pad caps residency below the wave slots):
# ingest tests, then a C2 year with pad 1024 (28 streams per CU) / 0 (all 32 slots: does the copy stream's priority alone do?) / 2560
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_e
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export ATLITE_HIP_INGEST_TIMEOUT_MS=3000
timeout 300 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "ingest tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/tests.log | head
F=/tmp/c8760.nc
timeout 400 python tools/bench_ingest.py --T 8760 --quick --keep $F > $OUT/pad1024.log 2>&1
for p in 0 2560 512; do ATLITE_HIP_INGEST_LDS_PAD=$p timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/pad$p.log 2>&1; done
ATLITE_HIP_INGEST_FED=0 timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/unfed.log 2>&1
for b in 32 64; do ATLITE_HIP_INGEST_BATCH=$b timeout 150 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/pad1024_b$b.log 2>&1; done
rm -f $F
for f in $OUT/pad*.log $OUT/unfed.log; do echo "== $(basename $f)"; grep "DEVICE\|stage split\|sha1\|Error\|error\|host threads" $f | cut -c1-400; done
