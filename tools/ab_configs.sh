#!/bin/bash
# A/B product vs variant libraries on tools/bench_configs.py cases (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  echo "== $(basename $lib)"
  ATLITE_HIP_LIB=$lib python $REPO/tools/bench_configs.py "$@" 2>/dev/null | grep -E "^C[0-9]"
done
