#!/bin/bash
# round 6: which host -> device paths work while a kernel's spinning waves fill every wave slot (tools/probes/feed_probe.hip)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_g
mkdir -p $OUT
cd $REPO
for cfg in "5120 10240" "5792 10240" "7328 10240" "5120 4096"; do
  echo "== LDS / grid: $cfg"
  timeout 120 tools/probes/feed_probe $cfg 2>&1 | tee -a $OUT/probe.log
done
