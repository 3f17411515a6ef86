# full gpu tests, A/B of the product (and variants, if any) with and without night skip, configs table
( python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 ) > gpurun_out/r02_gputests.log 2>&1; cat gpurun_out/r02_gputests.log
AB_ARGS="--night-skip none" bash tools/r02_job_ab.sh 2>/dev/null
python tools/bench_configs.py 2>&1 | grep -v "^{" | tail -6
