REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
 for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  echo "== $(basename $lib)"
  ATLITE_HIP_LIB=$lib python $REPO/tools/bench_configs.py C3 2>&1 | grep "^C3 "
  ATLITE_HIP_LIB=$lib python $REPO/tools/bench_pv_variants.py 2>&1 | grep "per-cell series"
 done
done
