python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -8
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
 for lib in $REPO/atlite_amd/lib/libatlite_hip.so $REPO/atlite_amd/lib/variants/*.so; do
  for args in "" "--night-skip" "--shape-kind star" "--shape-kind star --night-skip"; do
  ATLITE_HIP_LIB=$lib python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras $args 2>/dev/null | tail -1 | \
   python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-20s %-32s kernel_ms=%.3f min=%.3f' % ('$(basename $lib)', '$args', j['roofline']['kernel_ms'], j['roofline']['kernel_ms_min']))"
  done
 done
done
