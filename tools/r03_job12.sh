#!/bin/bash
mkdir -p gpurun_out/r03_job12; O=gpurun_out/r03_job12
timeout 600 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_interleave.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3 4; do
  P=1; [ $i = 3 ] && P=0
  ATLITE_HIP_PLACE=$P ATLITE_HIP_DEBUG_PLACE=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity > $O/b$i.json 2> $O/b$i.err
  grep "placement" $O/b$i.err
  python -c "
import json;j=json.loads([l for l in open('$O/b$i.json').read().splitlines() if l.startswith('{')][-1]);print('place=$P kernel', round(j['roofline']['kernel_ms'],3), 'frac', round(j['roofline']['frac'],3))"
done
