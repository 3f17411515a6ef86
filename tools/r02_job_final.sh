# round-2 record: default bench line, secondary configs, pv variants, dense sweep, 1/8 shard overhead, 2-rank dry run
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/r02_bench_c2.json 2> gpurun_out/final/r02_bench_c2.err
python tools/bench_configs.py > gpurun_out/final/r02_configs.log 2>&1
python tools/bench_pv_variants.py > gpurun_out/final/r02_pv_variants.log 2>&1
python tools/bench_dense.py pv runoff wind > gpurun_out/final/r02_dense.log 2>&1
for P in 1 2; do python bench.py --emulate-shard 8 --pipeline $P --steps 50 --warmup 10 --no-parity --no-cpu-baseline --no-extras > gpurun_out/final/r02_strong_shard8_overhead_p$P.json 2>> gpurun_out/final/emul.err; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --debug-gloo-one-gpu > gpurun_out/final/r02_gloo2.json 2> gpurun_out/final/r02_gloo2.err
tail -2 gpurun_out/final/r02_gloo2.err
cat gpurun_out/final/r02_bench_c2.json
