#!/bin/bash
mkdir -p gpurun_out/r03_job10; O=gpurun_out/r03_job10
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python - <<'P'
import json
j=json.loads(open('gpurun_out/r03_job10/bench_c2.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j.get('separate_cubes'), j['night_skip']['kernel_ms'], j['star_polygons']['kernel_ms'], j['api_e2e_ms']['warm'], j['parity']['max_rel_err'])
P
timeout 300 python tools/bench_configs.py 2>/dev/null | grep -v "^{" > $O/configs.log; cat $O/configs.log
timeout 300 python tools/bench_pv_variants.py 2>/dev/null | grep -v "^{" > $O/variants.log; head -12 $O/variants.log
