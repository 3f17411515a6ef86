#!/bin/bash
# round 6: a read cut into several jobs (ATLITE_HIP_INGEST_JOB_GB) - the verdicts' copy queued behind every job's kernel (variants/lib_queued_verdicts.so:
# the commit before) against verdicts fetched when the slot is settled: does job k + 1's DMA run beside job k's kernel?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; export TMPDIR=/tmp
F=/tmp/year.nc
timeout 900 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F 2>&1 | grep "DEVICE" | cut -c1-200
for i in 1 2; do for v in new queued; do
  L=$REPO/atlite_amd/lib/libatlite_hip.so; [ $v = queued ] && L=$REPO/atlite_amd/lib/variants/lib_queued_verdicts.so
  for gb in 3 100; do echo "== $v, jobs of $gb GiB"; ATLITE_HIP_LIB=$L ATLITE_HIP_INGEST_JOB_GB=$gb timeout 600 python tools/bench_ingest.py --T 8760 --quick --no-host --keep $F 2>&1 | grep "DEVICE\|sha1" | cut -c60-200; done
done; done
