#!/bin/bash
# round-4 GPU job A (one gpurun call): GPU tests, the bench line with its new legs, A/B of the per-cell series kernels'
# store policy / batching / slot chunk (variant libraries built by tools/build_variant.sh), the pv family's new influx
# heads, the 1/8-shard overhead with and without a hipGraph, wind input layouts.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_a
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
echo "start $(date +%s)" > $OUT/status
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$? $(date +%s)" >> $OUT/status
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(date +%s)" >> $OUT/status
V=$REPO/atlite_amd/lib/variants
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_st1.so $V/lib_st2.so $V/lib_st3.so $V/lib_st4.so $V/lib_st5.so $V/lib_bs.so $V/lib_g8bs.so $V/lib_sl16.so $V/lib_sl64.so $REPO/atlite_amd/lib/libatlite_hip.so; do
  echo "== $(basename $lib)" >> $OUT/series_ab.log
  ATLITE_HIP_LIB=$lib timeout 300 python tools/bench_configs.py C3 C3m 2>/dev/null | grep -E "^C[0-9]" >> $OUT/series_ab.log
  ATLITE_HIP_LIB=$lib ATL_VARIANTS="per-cell series out (no matrix), no early-out|per-cell series out (no matrix) + night" timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "per-cell" >> $OUT/series_ab.log
done
echo "series_ab done $(date +%s)" >> $OUT/status
# the batch's converted values parked in LDS (kStageValues): in-kernel solar position with / without, the headline kernel at 4 waves
for lib in $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_spnostage.so $V/lib_pvstage.so $REPO/atlite_amd/lib/libatlite_hip.so $V/lib_spnostage.so $V/lib_pvstage.so; do
  echo "== $(basename $lib)" >> $OUT/stage_ab.log
  ATLITE_HIP_LIB=$lib ATL_VARIANTS="getter, scalar orientation|in-kernel solar position (5 cubes|getter, per-cell orientation|trigon_model='other') - fast" ATL_VARIANT_REPS=8 timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -E "ms " >> $OUT/stage_ab.log
done
echo "stage_ab done $(date +%s)" >> $OUT/status
ATL_VARIANTS="influx|getter, scalar orientation|trigon_model='other') - fast" timeout 600 python tools/bench_pv_variants.py > $OUT/pv_variants.log 2>&1; echo "pv_variants rc=$? $(date +%s)" >> $OUT/status
for layout in interleaved separate; do
  ATL_BENCH_WIND_LAYOUT=$layout timeout 600 python bench.py --legs c3_series,c3_cf_map,c3_aggregated --no-cpu-baseline --no-parity > $OUT/wind_$layout.json 2>> $OUT/bench.err
done
echo "wind layouts done $(date +%s)" >> $OUT/status
timeout 300 python bench.py --emulate-shard 8 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $OUT/shard8.json 2>> $OUT/bench.err
timeout 300 python bench.py --emulate-shard 8 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --graph > $OUT/shard8_graph.json 2>> $OUT/bench.err
timeout 300 python bench.py --debug-rccl-self --steps 50 --warmup 5 --T 1095 --no-cpu-baseline --no-extras > $OUT/rccl_self_lib.json 2>> $OUT/bench.err
timeout 300 python bench.py --debug-rccl-self --collective torch --steps 50 --warmup 5 --T 1095 --no-cpu-baseline --no-extras > $OUT/rccl_self_torch.json 2>> $OUT/bench.err
echo "end $(date +%s)" >> $OUT/status
tail -3 $OUT/gputests.log; cat $OUT/status; tail -c 600 $OUT/bench.err
