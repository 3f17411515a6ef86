#!/usr/bin/env python3
"""profiles/r06_split_kernels.txt from the rocprofv3 databases of tools/jobs/r06_y.sh (kernel trace) and tools/jobs/r06_split_sq.sh (SQ
counters) - the segment scheme's kernels (k_find_blocks, k_segments_pool, k_gather, k_gather_rest, k_adler_parts, k_unpack) on one read of a
C2-grid cutout in atlite's own chunking.  Run here after the two jobs: python tools/profile_split.py"""
import collections
import glob
import sqlite3
import statistics
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    con = sqlite3.connect(glob.glob(str(ROOT / "gpurun_out/r06_y/prof/**/*.db"), recursive=True)[0])
    rows = list(con.execute("select name, start, end, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x from kernels order by start"))
    agg = collections.OrderedDict()
    for n, s, e, v, a, sg, l, sc, w in rows:
        d = agg.setdefault(short(n), {"t": [], "res": (v, a, sg, l, sc, w)})
        d["t"].append((e - s) / 1e3)
    lines = ["== r06: rocprofv3 --kernel-trace -- python tools/bench_ingest.py --T 2000 --quick --no-host --chunks 100,200,200  (job tools/jobs/r06_y.sh; durations in us) ==",
             "a C2-grid cutout in atlite's own chunking (100, 200, 200): 140 streams of 16 MB, 1.26 GB stored, 2.24 GB inflated; 5 calls of Cutout(path).pv(matrix=M)",
             f"{'kernel':44s} {'calls':>6s} {'avg':>10s} {'median':>10s} {'min':>10s} {'max':>10s}   vgpr agpr sgpr    lds scratch   wg"]
    for k, d in sorted(agg.items(), key=lambda x: -sum(x[1]["t"])):
        t = d["t"]
        v, a, sg, l, sc, w = d["res"]
        lines.append(f"{k[:44]:44s} {len(t):6d} {sum(t) / len(t):10.1f} {statistics.median(t):10.1f} {min(t):10.1f} {max(t):10.1f}   {v:4d} {a:4d} {sg:4d} {l:6d} {sc:6d} {w:4d}")
    last = [i for i, r in enumerate(rows) if "k_gather" in r[0]][-1]
    i0 = last
    while i0 > 0 and "k_find_blocks" not in rows[i0][0]:
        i0 -= 1
    while i0 > 0 and "k_find_blocks" in rows[i0 - 1][0]:
        i0 -= 1
    t0 = rows[i0][1]
    lines += ["", "== the last call's launches (ms from its first finder kernel) =="]
    for n, s, e, *_ in rows[i0:last + 12]:
        lines.append(f"{(s - t0) / 1e6:9.2f} +{(e - s) / 1e6:8.2f}  {short(n)[:60]}")
    f = glob.glob(str(ROOT / "gpurun_out/r06_split_sq/prof/**/*.db"), recursive=True)
    if f:
        con = sqlite3.connect(f[0])
        q = ("select kernel_name, counter_name, count(*), avg(v) from (select dispatch_id, kernel_name, counter_name, sum(value) as v from counters_collection "
             "group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name")
        dur = {short(n): a for n, a in con.execute("select name, avg(duration) from kernels group by name")}
        sq = {}
        for k, cn, c, a in con.execute(q):
            sq.setdefault(short(k), {})[cn] = a
        lines += ["", "== rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace (its own pass, job",
                  "   tools/jobs/r06_split_sq.sh; derived as tools/profile_bench.py derives them: per launch, summed over the counter's instances, SQ wave counters in",
                  "   quad-cycles; a --pmc pass serialises dispatches) =="]
        for k, e in sorted(sq.items()):
            if not (k.startswith("k_segments") or k.startswith("k_gather") or k.startswith("k_find") or k.startswith("k_unpack") or k.startswith("k_adler_parts")):
                continue
            cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8.0
            if cyc <= 0 or not e.get("SQ_WAVE_CYCLES"):
                continue
            der = {"clock_GHz": cyc / dur[k], "valu_busy": 4 * e["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), "resident_waves_per_simd": 4 * e["SQ_WAVE_CYCLES"] / (cyc * 1024)}
            for cn, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt")):
                der[lab] = e[cn] / e["SQ_WAVE_CYCLES"]
            lines.append(f"{k:20s} avg {dur[k] / 1e3:9.1f} us  " + " ".join(f"{a}={b:.2f}" for a, b in der.items()))
    (ROOT / "profiles" / "r06_split_kernels.txt").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
