#!/bin/bash
# gpu suite + pv variants table (+ the same table with a variant library, lines of interest only)
mkdir -p gpurun_out/cells
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/cells/pytest.log | tail -6
timeout 400 python tools/bench_pv_variants.py > gpurun_out/cells/variants.txt 2>&1
grep -i "early-out\|Traceback\|Error" gpurun_out/cells/variants.txt
for v in trkfastdiv; do
  ATLITE_HIP_LIB=$PWD/atlite_amd/lib/variants/lib_$v.so timeout 400 python tools/bench_pv_variants.py 2>&1 | grep -i "tracking.*early-out" > gpurun_out/cells/variants_$v.txt
  cat gpurun_out/cells/variants_$v.txt
done
