#!/usr/bin/env python3
"""
rocprofv3 over bench.py itself, leg by leg (run on the GPU box):

    python tools/profile_bench.py <out_dir> [group ...]        groups: headline night_skip star configs c4

For every group three separate passes of the SAME command (``python bench.py --legs ...``): --kernel-trace --stats,
--pmc FETCH_SIZE, --pmc WRITE_SIZE (never combined with other trace domains).  The dominant kernel of a leg is found by
NAME (bench.KERNELS), never by "largest total".  Writes <out_dir>/<group>.txt (per-kernel tables of the three passes)
and <out_dir>/bench_profile_latest.json:

    {"legs": {leg: {kernel, calls, avg_us, median_us, min_us, max_us, read_bytes, write_bytes, hbm_bytes_per_launch, source}}}

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
half of a wide (16 B per lane) streaming read, so the read side is doubled (re-derived with a known-size read in
profiles/r03_partial_line_probe.txt).  Copy the directory's .txt / .json files into profiles/ to commit them.
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

GROUPS = {
    # group -> (bench.py arguments, {leg: kernel name key in bench.KERNELS})
    "headline": (["--legs", "none"], ["headline"]),
    "night_skip": (["--legs", "night_skip"], ["night_skip"]),
    "star": (["--legs", "none", "--shape-kind", "star"], ["star_polygons"]),
    "configs": (["--legs", "c3_series,c3_cf_map,c3_aggregated,c5_heat,c5_runoff"],
                ["c3_series", "c3_cf_map", "c3_aggregated", "c5_heat", "c5_runoff"]),
    "c4": (["--legs", "c4_full_sp"], ["c4_full_sp"]),
    "c2sp": (["--legs", "c2_sp"], ["c2_sp"]),
    # the same kernel NAME as the headline: this leg's launches are told apart by their place in the pass (SLICES)
    "odd": (["--legs", "odd_caller"], ["odd_caller"]),
}
# legs whose kernel also runs for other workloads of the same pass: (first launch, launches) of the leg's TIMED launches in
# the pass's sequence of that kernel.  odd_caller: 9 headline launches, then 10 + 6 on the ordinary plan, then 10 warm-up + 6
# timed on the line-aligned plan
SLICES = {"odd_caller": (9 + 16 + 10, 6)}
COMMON = ["--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-parity"]
# untimed launches of a leg's kernel before bench.py's timed ones (its --warmup, the 12 warm-up launches of the side legs, the
# 10 of the configs legs): `avg_us` is over every launch of the pass, `avg_us_timed` over those after the warm-up - the
# figure bench.py's own HIP-event mean is comparable with (the first launches after an idle gap run at a colder clock)
WARM = {"headline": 3, "star_polygons": 3, "night_skip": 12}
WARM_CONFIGS = 10


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("> >", ">>")


def db(path):
    f = glob.glob(f"{path}/**/*.db", recursive=True)
    return sqlite3.connect(f[0]) if f else None


def run_pass(out, name, extra):
    d = f"{out}/{name}"
    cmd = ["rocprofv3", *extra, "-d", d, "-o", "run", "--", sys.executable, str(ROOT / "bench.py")]
    return d, cmd


def main():
    out = Path(sys.argv[1]).resolve()
    groups = sys.argv[2:] or list(GROUPS)
    out.mkdir(parents=True, exist_ok=True)
    from bench import KERNELS

    env = dict(os.environ, TMPDIR="/tmp")
    latest_f = out / "bench_profile_latest.json"
    latest = json.loads(latest_f.read_text()) if latest_f.exists() else {"note": __doc__.strip().split("\n\n")[0], "legs": {}}
    tag = os.environ.get("ATL_PROFILE_TAG", "r06")
    import socket

    from bench import box_id

    box = box_id()  # which box (host name + boot id) the record was measured on: bench.py quotes it beside a line from another one
    for g in groups:
        args, legs = GROUPS[g]
        lines, per_pass = [], {}
        passes = [("stats", ["--kernel-trace", "--stats"]), ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"]),
                  ("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"])]
        if os.environ.get("ATL_PROFILE_SQ", "1") != "0":  # a fourth pass: how busy the SIMDs are (quad-cycle counters)
            passes.append(("sq", ["--pmc", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                  "GRBM_GUI_ACTIVE", "--kernel-trace"]))
        for pname, extra in passes:
            d, cmd = run_pass(out / f"raw_{g}", pname, extra)
            with open(out / f"{g}.{pname}.log", "w") as log:
                subprocess.run(cmd + args + COMMON, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
            per_pass[pname] = db(d)
        stats = {}
        con = per_pass["stats"]
        if con:
            lines.append(f"== {tag} {g}: rocprofv3 --kernel-trace --stats -- python bench.py {' '.join(args + COMMON)}  (durations in us) ==")
            lines.append(f"{'kernel':72s} {'calls':>6s} {'total':>12s} {'avg':>10s} {'median':>10s} {'min':>10s} {'max':>10s} {'%':>6s}")
            per = {}
            for n, du in con.execute("select name, duration from kernels order by start"):
                per.setdefault(short(n), []).append(du / 1e3)
            tot = sum(sum(v) for v in per.values())
            for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                sv = sorted(v)
                stats[n] = dict(calls=len(v), avg_us=sum(v) / len(v), median_us=sv[len(sv) // 2], min_us=sv[0], max_us=sv[-1], in_order_us=v)
                lines.append(f"{n:72s} {len(v):6d} {sum(v):12.1f} {stats[n]['avg_us']:10.1f} {stats[n]['median_us']:10.1f} {sv[0]:10.1f} {sv[-1]:10.1f} {100 * sum(v) / tot:6.2f}")
            lines.append("")
            lines.append("== dispatch resources (as rocprofv3 reports them) ==")
            for n, v, av, sg, l, sc, gx, w in con.execute(
                    "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels group by name"):
                lines.append(f"{short(n):72s} vgpr={v} agpr={av} sgpr={sg} lds={l} scratch={sc} grid={gx} wg={w}")
        pmc, pmc_seq = {}, {}
        for pname, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            con = per_pass[pname]
            if not con:
                continue
            for k, v in con.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id, kernel_name "
                                    "order by dispatch_id", (cname,)):
                pmc_seq.setdefault(short(k), {}).setdefault(cname, []).append(v)
            lines.append("")
            lines.append(f"== rocprofv3 --pmc {cname} (its own pass): per-kernel averages, KiB per launch ==")
            for k, cn, c, a, mn, mx in con.execute("select kernel_name, counter_name, count(*), avg(v), min(v), max(v) from (select dispatch_id, "
                                                    "kernel_name, counter_name, sum(value) as v from counters_collection group by dispatch_id, "
                                                    "kernel_name, counter_name) group by kernel_name, counter_name"):
                pmc.setdefault(short(k), {})[cn] = a
                lines.append(f"{short(k):72s} {cn:11s} n={c:3d} avg={a:16.1f} min={mn:16.1f} max={mx:16.1f}")
        sq = {}
        con = per_pass.get("sq")
        if con:
            lines.append("")
            lines.append("== rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE (its own pass; per launch, summed over the counter's instances; SQ wave counters in "
                         "quad-cycles; dispatches of a --pmc pass are serialised, so clocks are colder than in the stats pass) ==")
            rows = con.execute("select kernel_name, counter_name, count(*), avg(v) from (select dispatch_id, kernel_name, counter_name, sum(value) as v "
                               "from counters_collection group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name").fetchall()
            dur = {short(n): a for n, a in con.execute("select name, avg(duration) from kernels group by name")}
            for k, cn, c, a in rows:
                sq.setdefault(short(k), {})[cn] = a
            for k, e in sorted(sq.items()):
                if not (k.startswith("k_fused") or k.startswith("k_cells")):
                    continue
                cyc = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # shader cycles per XCD
                der = {}
                if cyc > 0 and dur.get(k):
                    der["clock_GHz"] = cyc / dur[k]
                if cyc > 0 and "SQ_ACTIVE_INST_VALU" in e:
                    der["valu_busy"] = 4 * e["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)
                if cyc > 0 and "SQ_WAVE_CYCLES" in e:
                    der["resident_waves_per_simd"] = 4 * e["SQ_WAVE_CYCLES"] / (cyc * 1024)
                if e.get("SQ_WAVE_CYCLES"):
                    for cn, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt")):
                        if cn in e:
                            der[lab] = e[cn] / e["SQ_WAVE_CYCLES"]
                e["_derived"] = der
                lines.append(f"{k:72s} " + " ".join(f"{a}={b:.2f}" for a, b in der.items()))
        lines.append("")
        lines.append("== legs of this group (kernel chosen by NAME; read = 2 x FETCH_SIZE KiB [gfx950 wide-read correction], write = WRITE_SIZE KiB) ==")
        for leg in legs:
            k = KERNELS[leg]
            if k not in stats:
                lines.append(f"{leg}: kernel {k} did not run in this pass")
                continue
            st = dict(stats[k])
            seq = st.pop("in_order_us")
            w = WARM.get(leg, WARM_CONFIGS)
            n_timed = int(COMMON[COMMON.index("--steps") + 1])  # (launches after those belong to other legs: the API calls)
            if leg in SLICES:
                w, n_timed = SLICES[leg]
            timed = seq[w:w + n_timed] if len(seq) > w else seq
            e = dict(kernel=k, **st, warmup_launches=w if len(seq) > w else 0, timed_launches=len(timed),
                     avg_us_timed=sum(timed) / len(timed), source=f"profiles/{tag}_bench_{g}.txt", box=box)
            if leg in SLICES and k in pmc_seq and len(pmc_seq[k].get("FETCH_SIZE", [])) >= w + n_timed:
                # the counters of THIS leg's launches only (the kernel's other launches in the pass read other cubes)
                f = pmc_seq[k]["FETCH_SIZE"][w:w + n_timed]
                wr = pmc_seq[k].get("WRITE_SIZE", [])[w:w + n_timed]
                e["read_bytes"] = 2.0 * sum(f) / len(f) * 1024
                e["write_bytes"] = (sum(wr) / len(wr) * 1024) if wr else 0.0
                e["hbm_bytes_per_launch"] = e["read_bytes"] + e["write_bytes"]
                e["calls"] = len(timed)
                e["avg_us"] = e["avg_us_timed"]
            elif leg in SLICES:
                pass  # no trustworthy counters for this leg in this pass
            elif k in pmc and "FETCH_SIZE" in pmc[k]:
                e["read_bytes"] = 2.0 * pmc[k]["FETCH_SIZE"] * 1024
                e["write_bytes"] = pmc[k].get("WRITE_SIZE", 0.0) * 1024
                e["hbm_bytes_per_launch"] = e["read_bytes"] + e["write_bytes"]
            if k in sq and sq[k].get("_derived"):
                e["sq"] = sq[k]["_derived"]
            latest["legs"][leg] = e
            lines.append(f"{leg}: {k}  avg {e['avg_us']:.1f} us over {e['calls']} launches, {e['avg_us_timed']:.1f} us over the {len(timed)} after the warm-up" +
                         (f"  read {e['read_bytes'] / 1e9:.3f} GB  write {e['write_bytes'] / 1e9:.3f} GB" if "read_bytes" in e else ""))
        (out / f"{tag}_bench_{g}.txt").write_text("\n".join(lines) + "\n")
        print("\n".join(lines), flush=True)
        latest_f.write_text(json.dumps(latest, indent=1) + "\n")
        # the raw databases are large: keep the summaries only
        subprocess.run(["rm", "-rf", str(out / f"raw_{g}")], check=False)


if __name__ == "__main__":
    main()
