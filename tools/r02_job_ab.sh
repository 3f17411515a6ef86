# A/B of the product library against the variants in atlite_amd/lib/variants (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${AB_ARGS:-"--night-skip"}
for r in 1 2; do
 for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  for ns in $ARGS; do
  [ "$ns" = "none" ] && ns=""
  ATLITE_HIP_LIB=$lib python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras $ns 2>/dev/null | tail -1 | \
   python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-20s %-12s kernel_ms=%.3f min=%.3f step_ms=%.3f' % ('$(basename $lib)', '$ns', j['roofline']['kernel_ms'], j['roofline']['kernel_ms_min'], j['ms_per_step']))"
  done
 done
done
python -m pytest $REPO/tests/test_gpu_parity.py $REPO/tests/test_gpu_fullsize_properties.py -x -q -m gpu -k "night or skip or fullsize" 2>&1 | tail -2
