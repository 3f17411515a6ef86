#!/bin/bash
# round 6: the staging area's size (output bytes per batch).  (1) k_segments_pool on 16 MB streams: ATL_POOL_STAGE 496 / 1024 / 1536 / 2048 / 4096 units
# (variants built by hand: see the ingest record); (2) k_inflate on the C2 year: ATL_STAGE 1024 (32 streams per CU) / 1536 / 2048 at 8 / 6 waves
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; export TMPDIR=/tmp
F=/tmp/year.nc
timeout 900 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F 2>&1 | grep "DEVICE" | cut -c60-200
for i in 1 2; do for v in libatlite_hip.so variants/lib_kstage1536_w8.so variants/lib_kstage1536_w6.so variants/lib_kstage2048_w8.so variants/lib_kstage2048_w6.so; do
echo "== $v"; ATLITE_HIP_LIB=$PWD/atlite_amd/lib/$v timeout 600 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F 2>&1 | grep "DEVICE\|sha1" | cut -c60-200
done; done
