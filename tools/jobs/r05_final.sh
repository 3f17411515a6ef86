#!/bin/bash
# end of round 5: the GPU suite, the rest of the suite under the fenced allocator, rocprofv3 over the bench legs and the
# plain bench line IN THE SAME JOB (one box: frac and frac_from_profile are comparable), the from-file leg
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export ATL_PROFILE_TAG=r05
hostname > $OUT/box.txt
date +%T
timeout 600 stdbuf -o0 -e0 python -X faulthandler -m pytest tests -m gpu -x -v --capture=no -p no:cacheprovider > $OUT/tests.log 2>&1
echo "suite rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
date +%T
# the files the fenced runs of job b did not reach (it stopped at a hang of that job's own making in test_gpu_ingest)
ATLITE_HIP_FENCE=1 ATLITE_HIP_FENCE_SLACK=8 timeout 500 python -X faulthandler -m pytest tests/test_gpu_ingest.py tests/test_gpu_interleave.py \
  tests/test_gpu_math.py tests/test_gpu_multidevice.py tests/test_gpu_odd_grids.py tests/test_gpu_parity.py tests/test_gpu_post.py \
  tests/test_gpu_streaming.py tests/test_gpu_wind_speed.py tests/test_gpu_day_map.py -m gpu -v --capture=no -p no:cacheprovider \
  -k "not rccl and not collective_branch and not bench_prints" > $OUT/fenced_rest.log 2>&1   # (RCCL itself segfaults on hipMemCreate-mapped buffers)
echo "fenced rest rc=$? $(grep -E 'passed|failed' $OUT/fenced_rest.log | tail -1)"; grep -n -i "fault\|HW Exception\|Aborted\|^FAILED" $OUT/fenced_rest.log | head
date +%T
cp profiles/bench_profile_latest.json $OUT/ 2>/dev/null
ATL_PROFILE_SQ=0 timeout 900 python tools/profile_bench.py $OUT headline night_skip configs > $OUT/profile.log 2>&1
grep -E "^(headline|night_skip|star_polygons|c3_|c5_|c4_)" $OUT/profile.log
cp $OUT/bench_profile_latest.json profiles/bench_profile_latest.json 2>/dev/null
date +%T
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
date +%T
