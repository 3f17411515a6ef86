#!/bin/bash
# round 6: the batch resolver with a "bytes still owed" bitmap (short in-batch matches copied by their own lanes, all that are ready at once)
# against the ordered one-match-at-a-time loop (lib_resolve_old.so = the previous commit's atl_ingest.hip): tests, cycle split, A/B/A/B at T = 8760
# (kept as the record of how it was measured: the experiment was not faster and its source is not in the tree - profiles/r06_ingest.txt)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_u
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
export ATLITE_HIP_INFLATE=device
ATLITE_HIP_LIB=$REPO/atlite_amd/lib/variants/lib_infprof.so timeout 300 python tools/bench_ingest.py --T 720 --quick --no-host > $OUT/prof.log 2>&1
grep "k_inflate \|decode 0" $OUT/prof.log | tail -3 | cut -c1-420
F=/tmp/year.nc
for i in 1 2; do
  for v in new old; do
    L=$REPO/atlite_amd/lib/libatlite_hip.so; [ $v = old ] && L=$REPO/atlite_amd/lib/variants/lib_resolve_old.so
    ATLITE_HIP_LIB=$L timeout 600 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F > $OUT/year_${v}_$i.log 2>&1
    echo "== $v $i"; grep "DEVICE\|launch\|sha1" $OUT/year_${v}_$i.log | cut -c1-330
  done
done
for c in "1263,29,29"; do
  for v in new old; do
    L=$REPO/atlite_amd/lib/libatlite_hip.so; [ $v = old ] && L=$REPO/atlite_amd/lib/variants/lib_resolve_old.so
    ATLITE_HIP_LIB=$L timeout 600 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks $c --keep /tmp/c.nc > $OUT/chunks_${v}.log 2>&1
    echo "== chunks $c $v"; grep "DEVICE\|launch\|sha1" $OUT/chunks_${v}.log | cut -c1-330
  done
done
