#!/bin/bash
# atlite_amd/lib/variants/lib_infprof.so: the product library with k_inflate instrumented (-DATL_INF_PROFILE: cycle counters
# around the decoder's phases, printed by stream 0 of every launch).  Run `make -C atlite_amd/csrc` first, then e.g.
#   ATLITE_HIP_LIB=$PWD/atlite_amd/lib/variants/lib_infprof.so python tools/bench_ingest.py --T 1440 --quick
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/atlite_amd/csrc
mkdir -p /tmp/atl_variant_infprof $ROOT/atlite_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ \
  -DATL_INF_PROFILE=1 -c $SRC/atl_ingest.hip -o /tmp/atl_variant_infprof/atl_ingest.o
OBJS=""
for f in atl_runtime atl_gis atl_gis_dev atl_comm atl_post atl_h5 atl_inflate atl_kernels atl_kernels_wind atl_kernels_pv atl_kernels_pvt atl_kernels_pvk atl_kernels_pvi atl_kernels_pvx atl_kernels_pvxa atl_kernels_pvkt atl_kernels_pvka atl_kernels_pvkc; do OBJS="$OBJS $SRC/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/atlite_amd/lib/variants/lib_infprof.so $OBJS /tmp/atl_variant_infprof/atl_ingest.o -ldl -lz
echo built $ROOT/atlite_amd/lib/variants/lib_infprof.so
