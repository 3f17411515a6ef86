#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh <name> "<extra hipcc flags>"   -> atlite_amd/lib/variants/lib_<name>.so
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/atlite_amd/lib/variants /tmp/atl_variant_$NAME
SRC=$ROOT/atlite_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ $EXTRA"
for f in atl_runtime.cpp atl_gis.cpp atl_comm.hip atl_h5.cpp atl_inflate.cpp atl_ingest.hip atl_kernels.hip; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f -o /tmp/atl_variant_$NAME/${f%.*}.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/atlite_amd/lib/variants/lib_$NAME.so /tmp/atl_variant_$NAME/*.o -ldl -lz
echo built $ROOT/atlite_amd/lib/variants/lib_$NAME.so
