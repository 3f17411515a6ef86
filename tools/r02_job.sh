set -x
python -m pytest tests/test_gpu_configs_at_size.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02_atsize.log
python bench.py > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
for P in 1 2 4; do python bench.py --emulate-shard 8 --pipeline $P --steps 50 --warmup 10 --no-parity --no-cpu-baseline --no-extras > gpurun_out/r02_emul8_p$P.json 2>> gpurun_out/r02_emul.err; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --debug-gloo-one-gpu > gpurun_out/r02_gloo2.json 2> gpurun_out/r02_gloo2.err
tail -3 gpurun_out/r02_gloo2.err
cat gpurun_out/r02_atsize.log
