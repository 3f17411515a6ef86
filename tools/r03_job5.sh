#!/bin/bash
# slot rows that are not 128-byte aligned (S % 16 != 0): which tile shape suits them?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job5; mkdir -p $O
for yx in "201 200" "189 157" "200 200"; do set -- $yx
 for tile in 16x8 32x4 64x2 flat; do
  for f in 0 1; do
   if [ "$yx" = "201 200" -a $f = 1 ]; then continue; fi
   if [ "$yx" = "200 200" -a $f = 1 ]; then continue; fi
   ( if [ $f = 1 ]; then export ATLITE_HIP_FORCE_VEC=1; fi
     ATLITE_HIP_TILE=$tile python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('grid %3s x %3s tile %-5s force_vec=$f: kernel_ms=%.3f  %.4g cell-steps/s  P=%s parity=%s' % ('$1','$2','$tile', j['roofline']['kernel_ms'], j['value'], j['config']['partial_rows'], j.get('parity',{}).get('max_rel_err')))" )
  done
 done
done > $O/tiles_unaligned.txt 2>&1
cat $O/tiles_unaligned.txt
