#!/bin/bash
# round-4 GPU job F: the record run - GPU tests, the bench line, rocprofv3 passes of its legs (stats, FETCH, WRITE, SQ),
# dense plans, the 1/8-shard overhead, the 1-rank RCCL path of the library's collective.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_f
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" > $OUT/status
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/status
timeout 600 python tools/bench_dense.py runoff wind pv > $OUT/dense.log 2>&1; echo "dense rc=$?" >> $OUT/status
timeout 300 python bench.py --emulate-shard 8 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $OUT/shard8.json 2>> $OUT/bench.err
timeout 300 python bench.py --debug-rccl-self --steps 50 --warmup 5 --T 1095 --no-cpu-baseline --no-extras > $OUT/rccl_self_lib.json 2>> $OUT/bench.err
timeout 1800 python tools/profile_bench.py $OUT/prof headline night_skip star configs c4 > $OUT/profile.log 2>&1; echo "profile rc=$?" >> $OUT/status
timeout 900 python bench.py > $OUT/bench2.json 2>> $OUT/bench.err; echo "bench2 rc=$?" >> $OUT/status
grep -v amdgpu $OUT/gputests.log | tail -2; cat $OUT/status
