#!/bin/bash
# round 5, job b - the device-side DEFLATE decoder's first run, then the fenced suites
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_b
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
export ATLITE_HIP_BACKTRACE=$OUT/backtrace.log
echo "== 1. ingest tests, host and device inflate"; date +%T
timeout 600 python -X faulthandler -m pytest tests/test_gpu_ingest.py -x -q > $OUT/ingest_tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/ingest_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/ingest_tests.log | head -20
echo "== 2. from-file bench"; date +%T
timeout 900 python tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/ingest_quick.log 2>&1
cut -c1-260 $OUT/ingest_quick.log
date +%T
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_ingest -o ingest -- python $REPO/tools/bench_ingest.py --T 1440 --quick --keep /tmp/c1440.nc > $OUT/ingest_prof.log 2>&1
cd $REPO
f=$(ls $OUT/prof_ingest/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
date +%T
echo "== 3. fenced suite, 8 bytes of slack (vectorised kernels on odd grids run)"
ATLITE_HIP_FENCE=1 ATLITE_HIP_FENCE_SLACK=8 AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 HIP_LAUNCH_BLOCKING=1 \
  timeout 1500 stdbuf -o0 -e0 python -X faulthandler -m pytest tests -m gpu -v --capture=no -p no:cacheprovider > $OUT/fenced8.log 2>&1
echo "fenced8 rc=$? $(grep -E ' passed| failed' $OUT/fenced8.log | tail -1)"
grep -n -i "fault\|HW Exception\|Aborted\|FAILED" $OUT/fenced8.log | head -20
date +%T
echo "== 4. fenced suite, no slack (blocks end on a page: the unvectorised kernels on odd grids)"
ATLITE_HIP_FENCE=1 ATLITE_HIP_FENCE_SLACK=0 AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 HIP_LAUNCH_BLOCKING=1 \
  timeout 1500 stdbuf -o0 -e0 python -X faulthandler -m pytest tests -m gpu -v --capture=no -p no:cacheprovider -k "not aligned_plans" > $OUT/fenced0.log 2>&1
echo "fenced0 rc=$? $(grep -E ' passed| failed' $OUT/fenced0.log | tail -1)"
grep -n -i "fault\|HW Exception\|Aborted\|FAILED" $OUT/fenced0.log | head -20
date +%T
