#!/bin/bash
# round-4 GPU job E: what the wind kernels' arithmetic costs (ablations), API timing after the gateway changes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_e
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
V=$REPO/atlite_amd/lib/variants
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_wnolog.so $V/lib_wnointerp.so $V/lib_wnone.so $REPO/atlite_amd/lib/libatlite_hip.so; do
  echo "== $(basename $lib)" >> $OUT/wind_ablate.log
  ATLITE_HIP_LIB=$lib timeout 300 python tools/bench_configs.py C3 C3m C3a 2>/dev/null | grep -E "^C[0-9]" >> $OUT/wind_ablate.log
done
timeout 300 python tools/profile_api.py > $OUT/profile_api.log 2>&1
timeout 600 python -m pytest tests/test_gpu_api_golden.py tests/test_gpu_interleave.py -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
cat $OUT/status; cat $OUT/wind_ablate.log; grep "warm call" $OUT/profile_api.log
