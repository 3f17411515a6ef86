#!/bin/bash
# round 6: the segment scheme's decode started when half of the DMA batches have landed ($ATLITE_HIP_SPLIT_EARLY) against one stage after
# the last DMA, A/B/A/B on one box (T = 2000 of the C2 grid in (100, y, x) chunks).  The knob left the library with the experiment: this is the
# record of how it was measured (git log -S ATLITE_HIP_SPLIT_EARLY).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=/tmp/large.nc
timeout 600 python tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 --keep $F 2>&1 | grep "DEVICE\|launch\|sha1" | cut -c1-300
for i in 1 2; do
echo "== early"; ATLITE_HIP_SPLIT_EARLY=1 timeout 600 python tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 --keep $F 2>&1 | grep "DEVICE\|launch\|sha1" | cut -c1-300
echo "== one stage"; timeout 600 python tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 --keep $F 2>&1 | grep "DEVICE\|launch\|sha1" | cut -c1-300
done
