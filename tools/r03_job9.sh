#!/bin/bash
# padded slots (atl_set_slot_stride): GPU suite, then padded against contiguous cubes on real-world shaped grids
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job9; mkdir -p $O
( timeout 480 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -n 15 $O/pytest.log
timeout 600 python tools/bench_pitch.py > $O/pitch.log 2>&1
grep -v amdgpu.ids $O/pitch.log
