#!/bin/bash
# per-cell night early-out: parity + variants table
mkdir -p gpurun_out/cells
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_golden.py tests/test_gpu_streaming.py tests/test_gpu_multidevice.py -m gpu -x -q > gpurun_out/cells/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/cells/pytest.log
tail -5 gpurun_out/cells/pytest.log
timeout 300 python tools/bench_pv_variants.py > gpurun_out/cells/variants.txt 2>&1
tail -8 gpurun_out/cells/variants.txt
