#!/bin/bash
# round 6: the 8-GPU rehearsal on one GPU (tools/scale_rehearsal.py) and where the cold / warm host time of cutout.pv() goes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_k
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1500 python tools/scale_rehearsal.py $OUT/scale_rehearsal.json > $OUT/rehearsal.log 2>&1
echo "rehearsal rc=$?"; tail -40 $OUT/rehearsal.log | cut -c1-200
timeout 300 python tools/profile_api.py > $OUT/profile_api.log 2>&1
echo "profile_api rc=$?"; grep -E "cold call|warm call" $OUT/profile_api.log
