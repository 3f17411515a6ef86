#!/bin/bash
# round-4 GPU job G: the driver's round-end sequence on the final tree - GPU tests, smoke, bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_g
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/status
grep -v amdgpu $OUT/gputests.log | tail -2; grep -v amdgpu $OUT/smoke.log | tail -2; cat $OUT/status
