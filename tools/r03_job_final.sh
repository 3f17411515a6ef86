#!/bin/bash
# Round 3, last check of the shipped build: GPU suite, smoke, the driver's bench command, and the pv-family profile group again
# (the in-kernel solar position kernel moved to 4 waves per SIMD after tools/r03_job7.sh ran).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job7
mkdir -p $O/summ
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
ATL_VARIANT_REPS=5 timeout 400 python tools/bench_pv_variants.py > $O/pv_variants.log 2>&1
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
name=pvfam; P=$O/prof_$name; mkdir -p $P
CMD="python $REPO/tools/profile_all.py pvfam"
timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o x -- $CMD > $P/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o x -- $CMD > $P/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o x -- $CMD > $P/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $P/pmc_sq -o x -- $CMD > $P/pmc_sq.log 2>&1
( cd $REPO && python tools/rocpd_summary.py $P $O/summ/r03_$name $name > /dev/null 2> $O/summ/r03_$name.err )
grep -vE "simple_timer|rocprofv3\]|^$|amdgpu.ids" $P/stats.log | tail -n 60 > $O/summ/r03_$name.stdout.log 2>/dev/null
rm -rf $P
echo "profiled $name: $(grep -c 'read=' $O/summ/r03_$name.txt 2>/dev/null) kernels with traffic"
