#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-parity $*"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o pv -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pv -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pv -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/*
