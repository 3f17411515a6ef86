#!/bin/bash
# gpu suite + per-cell lines of the variants table
mkdir -p gpurun_out/cells
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/cells/pytest.log | tail -8
timeout 400 python tools/bench_pv_variants.py > gpurun_out/cells/variants.txt 2>&1
grep -i "per-cell\|Traceback\|Error" gpurun_out/cells/variants.txt
