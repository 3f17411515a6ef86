#!/bin/bash
mkdir -p gpurun_out/cells
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
grep -E "passed|failed|rc=|^E  |Error" gpurun_out/cells/pytest.log | tail -12
