# interleaved A/B of the product library and the variants on the secondary configs (same box, two rounds)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
 for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  echo "== $(basename $lib)"
  ATLITE_HIP_LIB=$lib python $REPO/tools/bench_configs.py ${CFGS:-C3a C5h C5r C4s} 2>&1 | grep -v "^{" | grep "^C"
 done
done
