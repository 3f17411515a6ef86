#!/bin/bash
# round 6: a whole year from files of several chunkings, the library's own choice of path (segment scheme for few long streams)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_z
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for c in "100,200,200" "1263,29,29" "24,100,100"; do
  F=/tmp/c_$c.nc
  timeout 900 python tools/bench_ingest.py --T 8760 --quick --chunks $c --keep $F --default-policy > $OUT/chunks_$c.log 2>&1
  rm -f $F
  echo "== chunks $c"; grep "^wrote\|DEVICE\|launch\|host threads\|identical\|rror" $OUT/chunks_$c.log | cut -c1-360
done
