#!/bin/bash
# round-4 GPU job C: GPU tests, dense plans with the resident operand image, pv option fuzz (new influx heads), the
# host profile of a warm Cutout.pv() call, the N = 2 control flow of bench.py on one GPU (gloo), the bench line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_c
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "start $(date +%s)" > $OUT/status
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$? $(date +%s)" >> $OUT/status
timeout 600 python tools/bench_dense.py runoff wind pv > $OUT/dense.log 2>&1; echo "dense rc=$? $(date +%s)" >> $OUT/status
timeout 900 python tests/fuzz_pv_options.py 1500 404 > $OUT/fuzz_pv.log 2>&1; echo "fuzz_pv rc=$? $(date +%s)" >> $OUT/status
timeout 600 python tests/fuzz_gateway.py 400 405 > $OUT/fuzz_gateway.log 2>&1; echo "fuzz_gateway rc=$? $(date +%s)" >> $OUT/status
timeout 300 python tools/profile_api.py > $OUT/profile_api.log 2>&1; echo "profile_api rc=$? $(date +%s)" >> $OUT/status
timeout 600 python bench.py --gpus 2 --debug-gloo-one-gpu --steps 10 --warmup 3 --T 2190 > $OUT/gloo2.json 2> $OUT/gloo2.err; echo "gloo2 rc=$? $(date +%s)" >> $OUT/status
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(date +%s)" >> $OUT/status
echo "end $(date +%s)" >> $OUT/status
grep -v amdgpu $OUT/gputests.log | tail -3; cat $OUT/status; tail -3 $OUT/fuzz_pv.log; tail -c 600 $OUT/gloo2.err
