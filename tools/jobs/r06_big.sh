#!/bin/bash
# round 6: an 800 x 800 grid in atlite's own chunking - (100, 800, 800): 256 MB chunks - through the segment scheme (chunks beyond 64 MiB
# stayed on the host threads until now)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_big
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
ATLITE_HIP_INGEST_DEBUG=1 timeout 1500 python tools/bench_ingest.py --T ${1:-300} --Y 800 --X 800 --chunks 100,800,800 --quick --default-policy > $OUT/big.log 2>&1
grep "^wrote\|DEVICE\|launch\|host threads\|identical\|rror\|split: streams\|pool" $OUT/big.log | cut -c1-330 | tail -12
