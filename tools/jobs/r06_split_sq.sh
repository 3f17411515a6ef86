#!/bin/bash
# round 6: SQ counters of the segment scheme's kernels (its own pass: --pmc with --kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_split_sq
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
F=/tmp/large.nc
timeout 600 python $REPO/tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 --keep $F > $OUT/warm.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof -o sq -- python $REPO/tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 --keep $F > $OUT/sq.log 2>&1
ls $OUT/prof | head
