#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the secondary
# configurations of tools/bench_configs.py (wind / heat / runoff / pv shard).
# usage: tools/profile_configs.sh <tag>
set -u
TAG=${1:-r01_configs}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o cfg -- python $REPO/tools/bench_configs.py > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o cfg -- python $REPO/tools/bench_configs.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o cfg -- python $REPO/tools/bench_configs.py > $OUT/pmc_write.log 2>&1
ls -la $OUT/*
