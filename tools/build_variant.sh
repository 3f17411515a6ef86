#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh <name> "<extra hipcc flags>"   -> atlite_amd/lib/variants/lib_<name>.so
# Only the kernel units are recompiled (with the extra flags, in parallel); the other objects are the product's
# (run `make -C atlite_amd/csrc` first).
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/atlite_amd/lib/variants /tmp/atl_variant_$NAME
SRC=$ROOT/atlite_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ $EXTRA"
# UNITS="atl_kernels atl_kernels_pv" rebuilds only some kernel units; the others are taken from the product
UNITS=${UNITS:-"atl_kernels atl_kernels_wind atl_kernels_pv atl_kernels_pvt atl_kernels_pvk atl_kernels_pvi atl_kernels_pvx atl_kernels_pvxa atl_kernels_pvkt atl_kernels_pvka atl_kernels_pvkc"}
: > /tmp/atl_variant_$NAME/resource.txt
for u in $UNITS; do
  (/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $SRC/$u.hip -o /tmp/atl_variant_$NAME/$u.o 2> /tmp/atl_variant_$NAME/$u.resource.txt || (tail -30 /tmp/atl_variant_$NAME/$u.resource.txt; false)) &
done
wait
KO=""
for u in atl_kernels atl_kernels_wind atl_kernels_pv atl_kernels_pvt atl_kernels_pvk atl_kernels_pvi atl_kernels_pvx atl_kernels_pvxa atl_kernels_pvkt atl_kernels_pvka atl_kernels_pvkc; do
  if [ -f /tmp/atl_variant_$NAME/$u.o ] && echo "$UNITS" | grep -qw $u; then KO="$KO /tmp/atl_variant_$NAME/$u.o"; cat /tmp/atl_variant_$NAME/$u.resource.txt >> /tmp/atl_variant_$NAME/resource.txt; else KO="$KO $SRC/$u.o"; fi
done
OBJS=""
for f in atl_runtime atl_gis atl_gis_dev atl_comm atl_post atl_h5 atl_inflate atl_ingest; do OBJS="$OBJS $SRC/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/atlite_amd/lib/variants/lib_$NAME.so $KO $OBJS -ldl -lz
echo built $ROOT/atlite_amd/lib/variants/lib_$NAME.so
