#!/bin/bash
# Build the sanitizer variant of the library (host code under ASan + UBSan) and run the CPU suites that feed it
# files, streams, polygons and matrices - including the corruption / differential fuzzers - against it.
#   tools/run_asan_tests.sh [pytest args]      (no GPU needed)
set -e -o pipefail
ROOT=$(cd $(dirname $0)/.. && pwd)
make -C $ROOT/atlite_amd/csrc -j8 asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ATLITE_HIP_LIB=$ROOT/atlite_amd/lib/libatlite_hip_asan.so \
  python -m pytest tests/test_nc_reader.py tests/test_crs.py tests/test_host_logic.py tests/test_host_math.py tests/test_host_pv.py tests/test_host_wind.py -q -m "not gpu" -p no:cacheprovider "$@"
# short seeded stretches of the host fuzzers and the C-ABI misuse sweep against the same build
for t in "fuzz_api_misuse.py" "fuzz_plan.py 300 5" "fuzz_gis.py 200 5" "fuzz_interp.py 300 5" "fuzz_inflate.py 150 5"; do
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    ATLITE_HIP_LIB=$ROOT/atlite_amd/lib/libatlite_hip_asan.so python tools/$t | tail -1
done
