# round-2 baseline job: gpu tests, default bench, rocprof stats + PMC of the same command, variants/configs tables
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r02_gputests.log 2>&1
tail -5 gpurun_out/r02_gputests.log
python bench.py > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
cat gpurun_out/r02_bench_c2.json
bash tools/profile_gpu.sh r02 --no-extras > gpurun_out/r02_profile.log 2>&1
cd $REPO
python tools/bench_pv_variants.py > gpurun_out/r02_pv_variants.log 2>&1
python tools/bench_configs.py > gpurun_out/r02_configs.log 2>&1
tail -30 gpurun_out/r02_pv_variants.log gpurun_out/r02_configs.log
