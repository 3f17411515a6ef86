#!/bin/bash
# round 6: from-file with netCDF-C's DEFAULT chunking of a (8760, 200, 200) float32 variable (~4 MiB chunks: 1263 x 29 x 29) and with 2.7 MB ones
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for c in "1263,29,29" "1095,25,25" "100,100,100"; do
  F=/tmp/c_$c.nc
  timeout 600 python tools/bench_ingest.py --T 8760 --quick --chunks $c --keep $F > $OUT/chunks_$c.log 2>&1
  rm -f $F
  echo "== chunks $c"; grep "^wrote\|DEVICE\|launch\|host threads\|identical\|rror" $OUT/chunks_$c.log | cut -c1-330
done
