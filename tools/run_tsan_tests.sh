#!/bin/bash
# Build the ThreadSanitizer variant of the library (host code, incl. the chunk reader's worker pool) and run the CPU
# suites that exercise threads - the file reader (parallel chunk inflate) and the host logic - against it.
#   tools/run_tsan_tests.sh [pytest args]      (no GPU needed)
set -e -o pipefail
ROOT=$(cd $(dirname $0)/.. && pwd)
make -C $ROOT/atlite_amd/csrc -j8 tsan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.tsan-x86_64.so)
cd $ROOT
LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=1:abort_on_error=1:report_signal_unsafe=0:second_deadlock_stack=1 \
  ATLITE_HIP_LIB=$ROOT/atlite_amd/lib/libatlite_hip_tsan.so ATLITE_HIP_IO_THREADS=${ATLITE_HIP_IO_THREADS:-8} \
  python -m pytest tests/test_nc_reader.py tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider "$@"
