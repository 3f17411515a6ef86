#!/bin/bash
# end of round 4: the GPU suite, rocprofv3 over every bench leg group (stats + FETCH_SIZE + WRITE_SIZE + SQ), one bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
grep -E "passed|failed|rror" $OUT/tests.log | tail -3
cp profiles/bench_profile_latest.json $OUT/
timeout 1500 python tools/profile_bench.py $OUT headline night_skip star configs c4 > $OUT/profile.log 2>&1
grep -E "^(headline|night_skip|star_polygons|c3_|c5_|c4_)" $OUT/profile.log
cp $OUT/bench_profile_latest.json profiles/bench_profile_latest.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
