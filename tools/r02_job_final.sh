# round-2 record: gpu tests, default bench line, secondary configs, pv variants, dense sweep, 1/8 shard overhead,
# 2-rank dry run, 1-rank RCCL branch, C4 / C5 at full size, bench.py --config c4 at N = 1
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; mkdir -p gpurun_out/final
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 > gpurun_out/final/r02_gputests.txt
cat gpurun_out/final/r02_gputests.txt
python bench.py > gpurun_out/final/r02_bench_c2.json 2> gpurun_out/final/r02_bench_c2.err
python tools/bench_configs.py > gpurun_out/final/r02_configs.log 2>&1
python tools/bench_pv_variants.py > gpurun_out/final/r02_pv_variants.log 2>&1
python tools/bench_dense.py pv runoff wind > gpurun_out/final/r02_dense.log 2>&1
python tools/bench_indicator.py > gpurun_out/final/r02_indicator.log 2>&1
for P in 1 2; do python bench.py --emulate-shard 8 --pipeline $P --steps 50 --warmup 10 --no-parity --no-cpu-baseline --no-extras > gpurun_out/final/r02_strong_shard8_overhead_p$P.json 2>> gpurun_out/final/emul.err; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --debug-gloo-one-gpu > gpurun_out/final/r02_gloo2.json 2> gpurun_out/final/r02_gloo2.err
python bench.py --debug-rccl-self --pipeline 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > gpurun_out/final/r02_rccl_self.json
( timeout 600 python tools/check_c4_device_sp.py 2>&1 | tail -5; timeout 600 python tools/check_c5_fullsize.py 2>&1 | tail -3 ) > gpurun_out/final/r02_fullsize_c4_c5.log
timeout 900 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > gpurun_out/final/r02_bench_c4_n1.json
cat gpurun_out/final/r02_fullsize_c4_c5.log
bash tools/r02_job_cells_prof.sh > gpurun_out/final/r02_cells_prof.log 2>&1
cp gpurun_out/summ/r02_pv_c2_cells.* gpurun_out/final/ 2>/dev/null
tail -12 gpurun_out/final/r02_cells_prof.log
