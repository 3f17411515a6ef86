#!/bin/bash
# round-4 GPU job D: what the in-kernel solar position / early-out kernels spend their VALU time on (ablations), after the
# root + reciprocal merge; GPU tests of the touched paths.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_d
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_golden.py tests/test_gpu_post.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
V=$REPO/atlite_amd/lib/variants
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_spnostage.so $V/lib_nolog.so $V/lib_nodiv.so $V/lib_nologdiv.so $V/lib_spnomath.so $REPO/atlite_amd/lib/libatlite_hip.so; do
  echo "== $(basename $lib)" >> $OUT/ablate.log
  ATLITE_HIP_LIB=$lib ATL_VARIANTS="in-kernel solar position|getter + night early-out|getter, scalar orientation" ATL_VARIANT_REPS=8 timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "ms " >> $OUT/ablate.log
done
timeout 300 python tools/profile_api.py > $OUT/profile_api.log 2>&1
cat $OUT/status; cat $OUT/ablate.log; grep "warm call" $OUT/profile_api.log
