for lib in atlite_amd/lib/libatlite_hip.so atlite_amd/lib/variants/lib_rc6.so atlite_amd/lib/variants/lib_rc8.so; do
  echo "== $lib"
  ATLITE_HIP_LIB=$PWD/$lib python tools/bench_dense.py pv runoff 2>&1 | grep -v "^{" | grep "R=3 \|R=4 \|R=8 \|k=1 \|k=2 \|k=4 "
done
