#!/bin/bash
# A/B the product library against variants in atlite_amd/lib/variants (run on the GPU box).
# usage: tools/ab_bench.sh [rounds] [bench args...]
R=${1:-2}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $R); do
 for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  ATLITE_HIP_LIB=$lib python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity "$@" 2>/dev/null | tail -1 | \
   python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-28s kernel_ms=%.3f step_ms=%.3f frac=%.3f' % ('$(basename $lib)', j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['frac']))"
 done
done
