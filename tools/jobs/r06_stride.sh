#!/bin/bash
# round 6: every stride-th block header starts a segment (~4 segments per wave slot): tests, then T = 2000 / a year of the C2 grid and T = 400 of 800 x 800
# in (100, y, x) chunks, with the debug line (headers, tasks, segments, pool)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu -k "block_by_block or payload" 2>&1 | tail -1
ATLITE_HIP_INGEST_DEBUG=1 timeout 900 python tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200 2>&1 | grep "DEVICE\|launch\|sha1\|one pass" | tail -4 | cut -c1-330
ATLITE_HIP_INGEST_DEBUG=1 timeout 1200 python tools/bench_ingest.py --T 8760 --quick --no-host --chunks 100,200,200 --default-policy 2>&1 | grep "DEVICE\|launch\|sha1\|one pass" | tail -4 | cut -c1-330
ATLITE_HIP_INGEST_DEBUG=1 timeout 1500 python tools/bench_ingest.py --T 400 --Y 800 --X 800 --chunks 100,800,800 --quick --no-host --default-policy 2>&1 | grep "DEVICE\|launch\|sha1\|one pass" | tail -4 | cut -c1-330
