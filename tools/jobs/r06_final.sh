#!/bin/bash
# round 6, one box for everything DESIGN.md tabulates: the GPU suite, rocprofv3 over every bench leg (stats + FETCH_SIZE + WRITE_SIZE + SQ
# passes, tools/profile_bench.py), the pv variants, the plain bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r06_final}
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export ATL_PROFILE_TAG=r06
hostname > $OUT/box.txt
date +%T
timeout 1500 stdbuf -o0 -e0 python -X faulthandler -m pytest tests -m gpu -x -q --capture=no -p no:cacheprovider > $OUT/tests.log 2>&1
echo "suite rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
date +%T
cp profiles/bench_profile_latest.json $OUT/ 2>/dev/null
ATL_PROFILE_SQ=1 timeout 2400 python tools/profile_bench.py $OUT headline night_skip star configs c2sp c4 odd > $OUT/profile.log 2>&1
grep -E "^(headline|night_skip|star_polygons|c3_|c5_|c4_|c2_sp|odd_caller)" $OUT/profile.log
cp $OUT/bench_profile_latest.json profiles/bench_profile_latest.json 2>/dev/null
date +%T
timeout 900 python tools/bench_pv_variants.py > $OUT/pv_variants.txt 2>&1
tail -5 $OUT/pv_variants.txt | cut -c1-200
date +%T
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 800 $OUT/bench.json; echo
date +%T
date +%T
