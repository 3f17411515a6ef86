#!/bin/bash
# long randomised differential runs (GPU vs oracle) through the public API on the round-3 kernels: vectorised on odd grids,
# plain influx head, 16-slot dense tiles, general kernel with the run-time tracker switch; with and without ATLITE_HIP_NO_VEC
mkdir -p gpurun_out/r03_fuzz
N1=${1:-2500}; S1=${2:-3031}; N2=${3:-1500}; S2=${4:-3032}
timeout 1200 python tests/fuzz_pv_options.py $N1 $S1 > gpurun_out/r03_fuzz/pv_options.log 2>&1; echo "rc=$?" >> gpurun_out/r03_fuzz/pv_options.log
timeout 900 python tests/fuzz_gateway.py $N2 $S2 > gpurun_out/r03_fuzz/gateway.log 2>&1; echo "rc=$?" >> gpurun_out/r03_fuzz/gateway.log
ATLITE_HIP_NO_VEC=1 timeout 600 python tests/fuzz_pv_options.py 600 $((S1+10)) > gpurun_out/r03_fuzz/pv_options_novec.log 2>&1; echo "rc=$?" >> gpurun_out/r03_fuzz/pv_options_novec.log
ATLITE_HIP_NO_VEC=1 timeout 600 python tests/fuzz_gateway.py 400 $((S2+10)) > gpurun_out/r03_fuzz/gateway_novec.log 2>&1; echo "rc=$?" >> gpurun_out/r03_fuzz/gateway_novec.log
for f in pv_options gateway pv_options_novec gateway_novec; do echo "== $f"; tail -n 3 gpurun_out/r03_fuzz/$f.log; done
