#!/bin/bash
# Round 3, final measurement sweep on the shipped build: the driver's bench command, the 8-rank gloo flow, the 1-rank RCCL
# collective branch, then rocprofv3 stats + FETCH_SIZE + WRITE_SIZE + SQ passes for every dominant kernel.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=$REPO/gpurun_out/r03_job7
mkdir -p $O/summ
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -c 400 $O/bench_c2.json
python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4_n1.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
timeout 300 python bench.py --gpus 8 --debug-gloo-one-gpu --steps 5 --warmup 2 > $O/gloo8.json 2> $O/gloo8.err; echo "gloo8 rc=$?"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 300 python bench.py --debug-rccl-self --steps 10 --warmup 5 --no-cpu-baseline --no-extras > $O/rccl_self.json 2> $O/rccl_self.err; echo "rccl-self rc=$?"
python bench.py --emulate-shard 8 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > $O/shard8.json 2> $O/shard8.err; echo "shard8 rc=$?"
timeout 300 python tools/launch_spread.py 30 night,base,wind > $O/spread.log 2>&1
ATL_VARIANT_REPS=5 timeout 400 python tools/bench_pv_variants.py > $O/pv_variants.log 2>&1
timeout 400 python tools/bench_configs.py > $O/configs.log 2>&1
timeout 400 python tools/bench_dense.py runoff wind pv > $O/dense.log 2>&1
grep -E "median" $O/configs.log
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
  grep -vE "simple_timer|rocprofv3\]|^$|amdgpu.ids" $P/stats.log | tail -n 60 > $O/summ/r03_$name.stdout.log 2>/dev/null
  rm -rf $P
  echo "profiled $name: $(grep -c 'read=' $O/summ/r03_$name.txt 2>/dev/null) kernels with traffic"
}
BA="--steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-extras"
prof pv_c2 pv_8760x200x200_100shapes_tessellation 0 python $REPO/bench.py $BA
prof pv_c2_nightskip pv_8760x200x200_100shapes_tessellation_nightskip 0 python $REPO/bench.py $BA --night-skip
prof pv_c2_star pv_8760x200x200_100shapes_star 0 python $REPO/bench.py $BA --shape-kind star
prof pv_c2_star_nightskip pv_8760x200x200_100shapes_star_nightskip 0 python $REPO/bench.py $BA --shape-kind star --night-skip
prof pvfam pvfam 0 python $REPO/tools/profile_all.py pvfam
prof cfg cfg 0 python $REPO/tools/profile_all.py cfg
prof dense dense 1 python $REPO/tools/profile_all.py dense
prof pv_c4_full pv_8760x800x800_500shapes_tessellation_sp 0 python $REPO/bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-extras
prof pv_c4_full_nightskip pv_8760x800x800_500shapes_tessellation_sp_nightskip 0 python $REPO/bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-extras --night-skip
ls $O/summ | wc -l
