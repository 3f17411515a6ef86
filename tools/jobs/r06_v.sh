#!/bin/bash
# round 6: the influx head with an albedo variable; trackers x the remaining tails / per-cell orientations behind a run-time tracker
# switch (atl_kernels_pvka.hip): parity, then the variants' times
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_v
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_golden.py tests/test_gpu_ingest.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
timeout 900 python tools/bench_pv_variants.py 2>&1 | grep -v amdgpu.ids > $OUT/pv_variants.txt
grep -i "track\|bofinger\|albedo" $OUT/pv_variants.txt | cut -c1-230
