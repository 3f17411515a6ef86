#!/bin/bash
# HBM read traffic of the odd_caller bench leg, launch by launch: ordinary plan vs line-aligned plan on the same contiguous cubes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_u
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o run -- python $REPO/bench.py --legs odd_caller --no-cpu-baseline --no-parity --steps 6 --warmup 3 > $OUT/fetch.log 2>&1
python - > $OUT/summary.txt 2>&1 <<'PY'
import glob, sqlite3
con = sqlite3.connect(glob.glob("/tmp/prof_f/**/*.db", recursive=True)[0])
rows = con.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name='FETCH_SIZE' group by dispatch_id, kernel_name order by dispatch_id").fetchall()
vals = [v for _, k, v in rows if "k_fused_segred<" in k and "PvConvT<false, false, false, 0, 0, 0>, true, false" in k]
print("launches of the pv kernel:", len(vals))
gb = [2.0 * v * 1024 / 1e9 for v in vals]  # KiB, x2: the gfx950 wide-read correction
print("read GB per launch, in order:", " ".join(f"{g:.2f}" for g in gb))
alg = 8760 * 201 * 201 * 56 / 1e9
head, rest = gb[:9], gb[9:]
half = len(rest) // 2
import statistics as st
print(f"C2 headline (first 9): {st.mean(head):.3f} GB")
print(f"201 x 201 contiguous cubes, ordinary plan ({half} launches): {st.mean(rest[:half]):.3f} GB = {st.mean(rest[:half]) / alg:.3f} x the algorithmic {alg:.3f} GB")
print(f"201 x 201 contiguous cubes, line-aligned plan ({len(rest) - half} launches): {st.mean(rest[half:]):.3f} GB = {st.mean(rest[half:]) / alg:.3f} x")
PY
cat $OUT/summary.txt
