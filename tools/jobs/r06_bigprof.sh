#!/bin/bash
# round 6: kernel timeline of a read of 256 MB chunks (800 x 800 grid, T = 400)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=/tmp/big.nc
timeout 900 python $R/tools/bench_ingest.py --T 400 --Y 800 --X 800 --chunks 100,800,800 --quick --no-host --default-policy --keep $F > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/bigprof -o big -- python $R/tools/bench_ingest.py --T 400 --Y 800 --X 800 --chunks 100,800,800 --quick --no-host --default-policy --keep $F > $R/gpurun_out/bigprof.log 2>&1
ls $R/gpurun_out/bigprof
