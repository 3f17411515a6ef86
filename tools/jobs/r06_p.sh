#!/bin/bash
# round 6: trackers x (Hay-Davies | irradiation | bofinger) as fused fast-family kernels (atl_kernels_pvkt.hip): parity, then the variants' times
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_p
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_golden.py tests/test_gpu_day_map.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
timeout 900 python tools/bench_pv_variants.py 2>&1 | grep -v amdgpu.ids > $OUT/pv_variants.txt
grep -i "track\|bofinger\|irradiation" $OUT/pv_variants.txt | cut -c1-200
