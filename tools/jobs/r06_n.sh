#!/bin/bash
# round 6: where the per-cell wind series' cubes lie (tools/probes/series_placement.py), three fresh processes on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_n
mkdir -p $OUT
cd $REPO
for i in 1 2 3; do echo "== process $i"; timeout 300 python tools/probes/series_placement.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/placement.log; done
