#!/bin/bash
# round 6: a C2 year in atlite's own chunking, the final tree against the first one-pass version (variants/lib_d594269.so), A/B/A/B on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; export TMPDIR=/tmp
F=/tmp/yl.nc
timeout 1200 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks 100,200,200 --default-policy --keep $F 2>&1 | grep "DEVICE" | cut -c60-200
for i in 1 2; do for v in libatlite_hip.so variants/lib_d594269.so; do
echo "== $v"; ATLITE_HIP_LIB=$PWD/atlite_amd/lib/$v timeout 600 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks 100,200,200 --default-policy --keep $F 2>&1 | grep "DEVICE\|launch\|sha1" | cut -c60-280
done; done
