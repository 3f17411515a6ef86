#!/bin/bash
# long randomised differential runs (GPU vs oracle) through the public API - the night early-out is on by default
mkdir -p gpurun_out/fuzz
N1=${1:-600}; S1=${2:-2024}; N2=${3:-400}; S2=${4:-2025}
timeout 900 python tests/fuzz_pv_options.py $N1 $S1 > gpurun_out/fuzz/pv_options.log 2>&1; echo "rc=$?" >> gpurun_out/fuzz/pv_options.log
timeout 600 python tests/fuzz_gateway.py $N2 $S2 > gpurun_out/fuzz/gateway.log 2>&1; echo "rc=$?" >> gpurun_out/fuzz/gateway.log
tail -3 gpurun_out/fuzz/pv_options.log; tail -3 gpurun_out/fuzz/gateway.log
