#!/bin/bash
# vectorised kernels on any grid + alignment-aware tile choice: GPU suite, then real-world shaped grids
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job6; mkdir -p $O
( timeout 480 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -n 5 $O/pytest.log
for yx in "200 200" "201 200" "201 201" "189 157" "157 189" "241 321"; do set -- $yx
  python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('grid %3s x %3s: kernel_ms=%.3f  %.4g cell-steps/s  tile %s P=%s parity=%s' % ('$1','$2', j['roofline']['kernel_ms'], j['value'], j['config']['cell_tile'], j['config']['partial_rows'], j.get('parity',{}).get('max_rel_err')))"
done > $O/grids.txt 2>&1
cat $O/grids.txt
for yx in "189 157"; do set -- $yx
  python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --night-skip --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('night grid %3s x %3s: kernel_ms=%.3f  %.4g cell-steps/s  tile %s parity=%s' % ('$1','$2', j['roofline']['kernel_ms'], j['value'], j['config']['cell_tile'], j.get('parity',{}).get('max_rel_err')))"
  ATLITE_HIP_NO_VEC=1 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NO_VEC grid %3s x %3s: kernel_ms=%.3f  %.4g cell-steps/s  tile %s' % ('$1','$2', j['roofline']['kernel_ms'], j['value'], j['config']['cell_tile']))"
done >> $O/grids.txt 2>&1
tail -n 2 $O/grids.txt
