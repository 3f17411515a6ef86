#!/bin/bash
# bench.py collective branch on a 1-rank RCCL group: step overlap on / off, plus the test that checks its parity
mkdir -p gpurun_out/overlap
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for f in "" "--no-step-overlap"; do
  python bench.py --debug-rccl-self --steps 30 --warmup 5 --T 1095 --no-cpu-baseline --no-extras $f 2>gpurun_out/overlap/err$f.txt | grep "^{" > gpurun_out/overlap/self_T1095$f.json
  python bench.py --debug-rccl-self --steps 20 --warmup 5 --no-cpu-baseline --no-extras $f 2>>gpurun_out/overlap/err$f.txt | grep "^{" > gpurun_out/overlap/self_full$f.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/overlap/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms_per_step=%.4f kernel_ms=%.4f parity=%s" % (j["ms_per_step"], j["roofline"]["kernel_ms"], j.get("parity",{}).get("ok")), j["config"].get("parallelism"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/overlap/err*.txt
timeout 600 python -m pytest tests/test_gpu_multidevice.py -q -m gpu 2>&1 | tail -3
