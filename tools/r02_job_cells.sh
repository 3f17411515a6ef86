#!/bin/bash
# gpu suite + pv variants table (+ the same table with a variant library, lines of interest only)
mkdir -p gpurun_out/cells
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
grep -E "passed|failed|rc=" gpurun_out/cells/pytest.log | tail -3
timeout 300 python tools/bench_pv_variants.py > gpurun_out/cells/variants.txt 2>&1
tail -14 gpurun_out/cells/variants.txt
for v in boftrk2; do
  ATLITE_HIP_LIB=$PWD/atlite_amd/lib/variants/lib_$v.so timeout 300 python tools/bench_pv_variants.py 2>&1 | grep -i "bofinger" > gpurun_out/cells/variants_$v.txt
  cat gpurun_out/cells/variants_$v.txt
done
