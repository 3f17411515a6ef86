#!/usr/bin/env python3
"""
Summarise rocprofv3 (rocpd sqlite) outputs into the small text/JSON files kept under profiles/.

  tools/rocpd_summary.py <prof_dir> <out_prefix> [workload-tag]

<prof_dir> holds stats/, pmc_fetch/, pmc_write/ as written by tools/profile_gpu.sh.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB, collected in separate passes; on gfx950 FETCH_SIZE reports exactly half
of a wide (16 B/lane) coalesced streaming read, so the read side is doubled.
"""
import glob
import json
import sqlite3
import sys


def db(path):
    f = glob.glob(f"{path}/*.db")
    return sqlite3.connect(f[0]) if f else None


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else None
    lines = []
    con = db(f"{prof}/stats")
    stats = {}
    if con:
        lines.append("== rocprofv3 --kernel-trace --stats : per-kernel summary (durations in us) ==")
        lines.append(f"{'kernel':48s} {'calls':>6s} {'total':>12s} {'avg':>10s} {'min':>10s} {'max':>10s} {'%':>6s}")
        rows = con.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
            "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        per = {}
        for n, st, du in con.execute("select name, start, duration from kernels order by start"):
            per.setdefault(n, []).append(du / 1e3)
        for n, c, t, a, mn, mx in rows:
            d = sorted(per[n])
            stats[short(n)] = dict(calls=c, avg_us=a, min_us=mn, max_us=mx, median_us=d[len(d) // 2])
            lines.append(f"{short(n):48s} {c:6d} {t:12.1f} {a:10.1f} {mn:10.1f} {mx:10.1f} {100*t/tot:6.2f}   median {d[len(d) // 2]:.1f}")
        r = con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, "
                        "workgroup_x from kernels group by name").fetchall()
        lines.append("")
        lines.append("== dispatch resources ==")
        for n, v, av, s, l, sc, g, w in r:
            lines.append(f"{short(n):48s} vgpr={v} agpr={av} sgpr={s} lds={l} scratch={sc} grid={g} wg={w}")
    pmc = {}
    for name in ("fetch", "write"):
        con = db(f"{prof}/pmc_{name}")
        if not con:
            continue
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                           "from counters_collection group by kernel_name, counter_name").fetchall()
        lines.append("")
        lines.append(f"== rocprofv3 --pmc ({name} pass): per-kernel counter averages (KiB per launch) ==")
        for k, cn, c, a, mn, mx in rows:
            pmc.setdefault(short(k), {})[cn] = a
            lines.append(f"{short(k):48s} {cn:12s} n={c:3d} avg={a:16.1f} min={mn:16.1f} max={mx:16.1f}")
    # SQ / GRBM passes (pmc_sq, pmc_sq2, ...): per-dispatch sums over the counter's instances, averaged over the launches
    sq = {}
    import glob as _glob
    for d in sorted(_glob.glob(f"{prof}/pmc_sq*")):
        con = db(d)
        if not con:
            continue
        rows = con.execute("select kernel_name, counter_name, count(*), avg(v) from (select dispatch_id, kernel_name, counter_name, "
                           "sum(value) as v from counters_collection group by dispatch_id, kernel_name, counter_name) "
                           "group by kernel_name, counter_name").fetchall()
        dur = {short(n): a for n, a in con.execute("select name, avg(duration) from kernels group by name")}
        for k, cn, c, a in rows:
            e = sq.setdefault((short(k), d.rsplit("/", 1)[-1]), {})
            e[cn] = a
            e["_n"] = c
            e["_dur_ns"] = dur.get(short(k), 0.0)
    if sq:
        lines.append("")
        lines.append("== SQ / GRBM counters (separate pass each; per launch; SQ_* wave counters in quad-cycles; the dispatches of a --pmc pass "
                     "are serialised with idle gaps, so clocks and durations here are colder than in the stats pass) ==")
        for (k, pas), e in sorted(sq.items()):
            if k.startswith("k_synth") or k.startswith("__amd") or k in ("k_combine", "k_chunk_reduce", "k_rows_timered"):
                continue
            g = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # shader cycles per XCD
            der = []
            if g > 0 and e["_dur_ns"] > 0:
                der.append(f"clock={g / e['_dur_ns']:.2f}GHz")
            if g > 0 and "SQ_ACTIVE_INST_VALU" in e:
                der.append(f"VALU_busy={100 * 4 * e['SQ_ACTIVE_INST_VALU'] / (g * 1024):.0f}%")
            if g > 0 and "SQ_WAVE_CYCLES" in e:
                der.append(f"resident_waves/SIMD={4 * e['SQ_WAVE_CYCLES'] / (g * 1024):.2f}")
            if e.get("SQ_WAVE_CYCLES"):
                w = e["SQ_WAVE_CYCLES"]
                for c, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt")):
                    if c in e:
                        der.append(f"{lab}={100 * e[c] / w:.0f}%")
            if g > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                der.append(f"MFMA_busy={100 * e['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024):.0f}%")
            raw = " ".join(f"{c}={v:.4g}" for c, v in sorted(e.items()) if not c.startswith("_"))
            lines.append(f"{k:60s} [{pas}] n={e['_n']} {e['_dur_ns'] / 1e3:9.1f} us  " + " ".join(der))
            lines.append(f"{'':60s}   {raw}")
    lines.append("")
    lines.append("== HBM traffic per launch (read = 2 x FETCH_SIZE KiB [gfx950 wide-read correction], write = WRITE_SIZE KiB) ==")
    summary = {}
    for k, d in pmc.items():
        rd = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024
        wr = d.get("WRITE_SIZE", 0.0) * 1024
        summary[k] = dict(read_bytes=rd, write_bytes=wr, hbm_bytes=rd + wr,
                          fetch_size_kib_raw=d.get("FETCH_SIZE"), write_size_kib_raw=d.get("WRITE_SIZE"))
        lines.append(f"{k:48s} read={rd/1e9:10.3f} GB write={wr/1e9:10.3f} GB total={(rd+wr)/1e9:10.3f} GB")
    open(out + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    allk = {k: dict(summary.get(k, {}), **stats.get(k, {})) for k in set(summary) | set(stats)
            if k.startswith("k_fused") or k.startswith("k_cells")}
    if allk:
        open(out + ".kernels.json", "w").write(json.dumps(allk, indent=1, sort_keys=True) + "\n")
    dom = max((k for k in summary if k.startswith("k_fused") or k.startswith("k_cells")),
              key=lambda k: summary[k]["hbm_bytes"], default=None)
    if dom and tag:
        j = dict(workload=tag, kernel=dom, hbm_bytes_per_launch=summary[dom]["hbm_bytes"], **summary[dom],
                 avg_kernel_us=stats.get(dom, {}).get("avg_us"))
        open(out + ".json", "w").write(json.dumps(j, indent=1) + "\n")


if __name__ == "__main__":
    main()
