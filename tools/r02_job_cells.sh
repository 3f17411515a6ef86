#!/bin/bash
# gpu suite + pv variants table
mkdir -p gpurun_out/cells
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
grep -E "passed|failed|rc=" gpurun_out/cells/pytest.log | tail -3
timeout 300 python tools/bench_pv_variants.py > gpurun_out/cells/variants.txt 2>&1
tail -12 gpurun_out/cells/variants.txt
