#!/bin/bash
# Round 3, GPU call 2: the GPU suite on the shrunk instantiation matrix; the LDS-DMA fed fused kernel (parity subset
# + A/B against the register-fed one, warm clocks); pv family before / after the refactor.   bash tools/r03_job2.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job2
mkdir -p $O
( timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -n 4 $O/pytest.log
( ATLITE_HIP_GLDS=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_properties.py tests/test_gpu_api_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_glds.log 2>&1; echo "pytest glds rc=$?" >> $O/pytest_glds.log )
tail -n 4 $O/pytest_glds.log
B="--steps 30 --warmup 12 --no-cpu-baseline --no-extras"
run() { # label, env, lib, extra args
  ( export $2; ATLITE_HIP_LIB=$3 python bench.py $B $4 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%-34s kernel_ms=%.3f med=%.3f min=%.3f step_ms=%.3f frac=%.3f parity=%s' % ('$1', r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_min'], j['ms_per_step'], r['frac'], j.get('parity',{}).get('max_rel_err')))" )
}
L=$REPO/atlite_amd/lib/libatlite_hip.so
V=$REPO/atlite_amd/lib/variants
for r in 1 2; do
  run "regs (product)" ATLITE_HIP_GLDS=0 $L
  run "glds rows5 (3 blocks/CU)" ATLITE_HIP_GLDS=1 $L
  run "glds rows3 (4 blocks/CU)" ATLITE_HIP_GLDS=1 $V/lib_g3.so
  run "glds rows1 (4 blocks/CU)" ATLITE_HIP_GLDS=1 $V/lib_g1.so
  run "r02 kernels" ATLITE_HIP_GLDS=0 $V/lib_r02kern.so
done > $O/ab_glds.txt 2>&1
cat $O/ab_glds.txt
for r in 1; do
  run "star regs" ATLITE_HIP_GLDS=0 $L "--shape-kind star"
  run "star glds" ATLITE_HIP_GLDS=1 $L "--shape-kind star"
  run "c4-shard regs" ATLITE_HIP_GLDS=0 $L "--T 1095 --Y 800 --X 800 --shapes 500"
  run "c4-shard glds" ATLITE_HIP_GLDS=1 $L "--T 1095 --Y 800 --X 800 --shapes 500"
done > $O/ab_glds2.txt 2>&1
cat $O/ab_glds2.txt
ATLITE_HIP_DEBUG_OCCUPANCY=1 ATLITE_HIP_GLDS=1 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extras 2>&1 >/dev/null | grep "atlite-hip" | sort | uniq -c > $O/occupancy_glds.txt
cat $O/occupancy_glds.txt
# the pv family after the instantiation refactor vs the round-2 kernels
ATL_VARIANT_REPS=5 timeout 400 python tools/bench_pv_variants.py > $O/pv_variants_new.log 2>&1
ATLITE_HIP_LIB=$V/lib_r02kern.so ATL_VARIANT_REPS=5 timeout 400 python tools/bench_pv_variants.py > $O/pv_variants_r02kern.log 2>&1
paste -d'\n' <(cut -c1-150 $O/pv_variants_new.log | sed 's/^/new  /') <(cut -c1-150 $O/pv_variants_r02kern.log | sed 's/^/r02  /') | grep -E "ms " | awk '{print}' > $O/pv_variants_side_by_side.txt
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(.+?)\s+([0-9.]+) ms", l)
        if m: d[m.group(1).strip()[:70]]=float(m.group(2))
    return d
a=load("$O/pv_variants_new.log"); b=load("$O/pv_variants_r02kern.log")
for k in a:
    print("%-72s new %7.3f  r02 %7.3f  %+5.1f%%" % (k, a[k], b.get(k, float('nan')), 100*(a[k]/b.get(k, float('nan'))-1)))
PY
