#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh <name> "<extra hipcc flags>"   -> atlite_amd/lib/variants/lib_<name>.so
# Only atl_kernels.hip is recompiled (with the extra flags); the other objects are the product's
# (run `make -C atlite_amd/csrc` first).  -DATL_NO_PV / -DATL_NO_PVX leave kernel families out so that
# a wind-only variant compiles in ~15 s.
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/atlite_amd/lib/variants /tmp/atl_variant_$NAME
SRC=$ROOT/atlite_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ $EXTRA"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $SRC/atl_kernels.hip -o /tmp/atl_variant_$NAME/atl_kernels.o 2> /tmp/atl_variant_$NAME/resource.txt || (tail -30 /tmp/atl_variant_$NAME/resource.txt; false)
OBJS=""
for f in atl_runtime atl_gis atl_comm atl_h5 atl_inflate atl_ingest; do OBJS="$OBJS $SRC/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/atlite_amd/lib/variants/lib_$NAME.so /tmp/atl_variant_$NAME/atl_kernels.o $OBJS -ldl -lz
echo built $ROOT/atlite_amd/lib/variants/lib_$NAME.so
