#!/bin/bash
# round 6: long chunk streams decoded block by block (k_find_blocks, k_segments, k_resolve): the new test, the ingest tests with the
# scheme forced on every read, then atlite's own chunking - (100, y, x): 16 MB streams - through bench_ingest
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_x
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider -k "block_by_block or payload" > $OUT/t1.log 2>&1
echo "new tests rc=$? $(grep -E 'passed|failed' $OUT/t1.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/t1.log | head
ATLITE_HIP_INFLATE_SPLIT=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider -k "not fed_launch" > $OUT/t2.log 2>&1
echo "forced rc=$? $(grep -E 'passed|failed' $OUT/t2.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/t2.log | head
timeout 900 python tools/bench_ingest.py --T ${1:-2000} --quick --chunks 100,200,200 --keep /tmp/large.nc > $OUT/large.log 2>&1
grep "^wrote\|DEVICE\|launch\|host threads\|identical\|rror" $OUT/large.log | cut -c1-360 | tail -8
echo "== count + decode (ATLITE_HIP_SPLIT_PASSES=2)"; ATLITE_HIP_SPLIT_PASSES=2 timeout 600 python tools/bench_ingest.py --T ${1:-2000} --quick --no-host --chunks 100,200,200 --keep /tmp/large.nc 2>&1 | grep "DEVICE\|launch\|sha1" | cut -c1-360
echo "== one pass, debug"; ATLITE_HIP_INGEST_DEBUG=1 timeout 600 python tools/bench_ingest.py --T ${1:-2000} --quick --no-host --chunks 100,200,200 --keep /tmp/large.nc 2>&1 | grep "split:" | tail -3 | cut -c1-300
ATLITE_HIP_SPLIT_PASSES=2 ATLITE_HIP_INFLATE_SPLIT=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider -k "not fed_launch" > $OUT/t3.log 2>&1
echo "forced, two passes rc=$? $(grep -E 'passed|failed' $OUT/t3.log | tail -1)"
