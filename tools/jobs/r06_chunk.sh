#!/bin/bash
# round 6: time slots per unit of the fused kernel ($ATLITE_HIP_CHUNK; the launcher picks <= 64): the one-cube converters (C5 heat demand, runoff) re-read
# their tiles' weight rows once per chunk - does a longer chunk buy the 4-5 % of traffic back?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; export TMPDIR=/tmp
for c in "" 128 256 512; do
  echo "== ATLITE_HIP_CHUNK=$c"
  ATLITE_HIP_CHUNK=$c timeout 900 python bench.py --legs c5_heat,c5_runoff,c3_aggregated --steps 6 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('headline %.3f ms' % d['roofline']['kernel_ms'], {k:(round(v['ms'],3), round(v['frac'],3)) for k,v in d['configs'].items()})"
done
