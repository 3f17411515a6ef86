#!/bin/bash
# round 6: segments per wave slot (4 / 8 / 16 / all headers; the knob left with the experiment) on 256 MB chunks and on a C2 year
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=/tmp/big.nc; G=/tmp/yl.nc
timeout 900 python tools/bench_ingest.py --T 400 --Y 800 --X 800 --chunks 100,800,800 --quick --no-host --default-policy --keep $F > /dev/null 2>&1
timeout 900 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks 100,200,200 --default-policy --keep $G > /dev/null 2>&1
for ps in 4 8 16 64; do echo "== per slot $ps"; 
ATLITE_HIP_SPLIT_PER_SLOT=$ps timeout 900 python tools/bench_ingest.py --T 400 --Y 800 --X 800 --chunks 100,800,800 --quick --no-host --default-policy --keep $F 2>&1 | grep "DEVICE\|launch" | cut -c60-120,200-330
ATLITE_HIP_SPLIT_PER_SLOT=$ps timeout 900 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks 100,200,200 --default-policy --keep $G 2>&1 | grep "DEVICE\|launch" | cut -c60-120,200-330
done
