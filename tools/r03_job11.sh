#!/bin/bash
# after the chunk-floor rule (weight rows vs cube bytes): the suite, the bench legs that depend on it and the dense profile group again
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job7
mkdir -p $O/summ
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python bench.py --emulate-shard 8 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > $O/shard8.json 2> $O/shard8.err; echo "shard8 rc=$?"
ATL_VARIANT_REPS=5 timeout 400 python tools/bench_pv_variants.py > $O/pv_variants.log 2>&1
timeout 400 python tools/bench_configs.py > $O/configs.log 2>&1
grep -E "median" $O/configs.log
timeout 400 python tools/bench_dense.py runoff wind pv > $O/dense.log 2>&1
grep -E "R=16|R=32|R=8 " $O/dense.log
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
name=dense; P=$O/prof_$name; mkdir -p $P
CMD="python $REPO/tools/profile_all.py dense"
timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o x -- $CMD > $P/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o x -- $CMD > $P/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o x -- $CMD > $P/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $P/pmc_sq -o x -- $CMD > $P/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d $P/pmc_sq2 -o x -- $CMD > $P/pmc_sq2.log 2>&1
( cd $REPO && python tools/rocpd_summary.py $P $O/summ/r03_$name $name > /dev/null 2> $O/summ/r03_$name.err )
grep -vE "simple_timer|rocprofv3\]|^$|amdgpu.ids" $P/stats.log | tail -n 60 > $O/summ/r03_$name.stdout.log 2>/dev/null
rm -rf $P
echo "profiled $name: $(grep -c 'read=' $O/summ/r03_$name.txt 2>/dev/null) kernels with traffic"
