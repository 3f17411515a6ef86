#!/bin/bash
# quick check: wind C3 kernels after the convert_batch fix; vectorised kernels forced onto odd grids (unaligned 16-byte accesses)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job4; mkdir -p $O
timeout 300 python tools/bench_configs.py C3 C3m C3a > $O/configs.log 2>&1; grep median $O/configs.log
for yx in "200 200" "201 201" "201 200" "189 157"; do set -- $yx
 for f in 0 1; do
  ( if [ $f = 1 ]; then export ATLITE_HIP_FORCE_VEC=1; fi
    python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('grid $1 x $2 force_vec=$f: kernel_ms=%.3f value=%.4g cell-steps/s parity=%s' % (j['roofline']['kernel_ms'], j['value'], j.get('parity',{}).get('max_rel_err')))" )
 done
done > $O/odd_grid.txt 2>&1
cat $O/odd_grid.txt
ATL_DENSE_R=16,32 timeout 300 python tools/bench_dense.py runoff wind > $O/dense.log 2>&1; grep "dense R" $O/dense.log
