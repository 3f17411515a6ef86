#!/bin/bash
# round-4 GPU job H: uncached partial rows of all-finite batches without guards - A/B on the early-out kernels and wind / runoff
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_h
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_properties.py -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
V=$REPO/atlite_amd/lib/variants
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_guarded.so $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_guarded.so; do
  echo "== $(basename $lib)" >> $OUT/ab.log
  ATLITE_HIP_LIB=$lib ATL_VARIANTS="getter + night early-out|in-kernel solar position + night|getter, scalar orientation" ATL_VARIANT_REPS=8 timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "ms " >> $OUT/ab.log
  ATLITE_HIP_LIB=$lib timeout 300 python tools/bench_configs.py C3a C5h C5r 2>/dev/null | grep -E "^C[0-9]" >> $OUT/ab.log
done
cat $OUT/status; cat $OUT/ab.log
