#!/bin/bash
# Round 3, GPU call 3: (1) GPU tests on the 16-slot dense tiles and the influx head's plain pair evaluation, (2) timings
# with warm clocks, (3) rocprofv3 stats + FETCH_SIZE + WRITE_SIZE + SQ counter passes for every dominant kernel
# (tools/profile_all.py groups; bench.py legs for C2 main / early-out / star / star + early-out and C4 at full size).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job3
mkdir -p $O/summ
( timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -n 4 $O/pytest.log
V=$REPO/atlite_amd/lib/variants
# dense tiles: 16-slot batches against the round-2 kernels (8-slot batches)
ATL_DENSE_R=16,32 timeout 300 python tools/bench_dense.py runoff wind pv > $O/dense_new.log 2>&1
ATLITE_HIP_LIB=$V/lib_r02kern.so ATL_DENSE_R=16,32 timeout 300 python tools/bench_dense.py runoff wind pv > $O/dense_r02.log 2>&1
echo "== dense new"; grep -E "dense R" $O/dense_new.log; echo "== dense r02"; grep -E "dense R" $O/dense_r02.log
# influx head + a few family members, warm
ATL_VARIANTS="influx / outflux dataset|getter, scalar|getter + night|in-kernel solar" timeout 200 python tools/bench_pv_variants.py > $O/pv_variants.log 2>&1
ATLITE_HIP_LIB=$V/lib_r02kern.so ATL_VARIANTS="influx / outflux dataset" timeout 200 python tools/bench_pv_variants.py > $O/pv_variants_r02.log 2>&1
cut -c1-170 $O/pv_variants.log | grep " ms"; echo "r02:"; cut -c1-170 $O/pv_variants_r02.log | grep " ms"
# other configs, warm clocks
timeout 400 python tools/bench_configs.py > $O/configs.log 2>&1; grep -E "median" $O/configs.log
# odd grid (201 x 201: the unvectorised kernels) against 200 x 200
for yx in "200 200" "201 201" "201 200"; do set -- $yx
  python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --Y $1 --X $2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('grid $1 x $2: kernel_ms=%.3f value=%.4g cell-steps/s parity=%s' % (j['roofline']['kernel_ms'], j['value'], j.get('parity',{}).get('max_rel_err')))"
done > $O/odd_grid.txt 2>&1
cat $O/odd_grid.txt
# ---- profiles ------------------------------------------------------------------------------------------------------
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
prof() { # name, tag, sq2?, command...
  local name=$1 tag=$2 sq2=$3; shift 3
  local P=$O/prof_$name; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o x -- "$@" > $P/stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o x -- "$@" > $P/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o x -- "$@" > $P/pmc_write.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $P/pmc_sq -o x -- "$@" > $P/pmc_sq.log 2>&1
  if [ "$sq2" = "1" ]; then timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d $P/pmc_sq2 -o x -- "$@" > $P/pmc_sq2.log 2>&1; fi
  ( cd $REPO && python tools/rocpd_summary.py $P $O/summ/r03_$name $tag > /dev/null 2> $O/summ/r03_$name.err )
  cp $P/stats.log $O/summ/r03_$name.stdout.log 2>/dev/null
  rm -rf $P
  echo "profiled $name: $(grep -c 'read=' $O/summ/r03_$name.txt 2>/dev/null) kernels with traffic"
}
BA="--steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-extras"
prof pv_c2 pv_8760x200x200_100shapes_tessellation 0 python $REPO/bench.py $BA
prof pv_c2_nightskip pv_8760x200x200_100shapes_tessellation_nightskip 0 python $REPO/bench.py $BA --night-skip
prof pv_c2_star pv_8760x200x200_100shapes_star 0 python $REPO/bench.py $BA --shape-kind star
prof pv_c2_star_nightskip pv_8760x200x200_100shapes_star_nightskip 0 python $REPO/bench.py $BA --shape-kind star --night-skip
prof pvfam pvfam 0 python $REPO/tools/profile_all.py pvfam
prof cfg cfg 0 python $REPO/tools/profile_all.py cfg
prof dense dense 1 python $REPO/tools/profile_all.py dense
prof pv_c4_full pv_8760x800x800_500shapes_tessellation_sp 0 python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras
prof pv_c4_full_nightskip pv_8760x800x800_500shapes_tessellation_sp_nightskip 0 python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras --night-skip
ls -la $O/summ | head -40
