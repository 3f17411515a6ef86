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
        for n, c, t, a, mn, mx in rows:
            stats[short(n)] = dict(calls=c, avg_us=a, min_us=mn, max_us=mx)
            lines.append(f"{short(n):48s} {c:6d} {t:12.1f} {a:10.1f} {mn:10.1f} {mx:10.1f} {100*t/tot:6.2f}")
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
    dom = max((k for k in summary if k.startswith("k_fused") or k.startswith("k_cells")),
              key=lambda k: summary[k]["hbm_bytes"], default=None)
    if dom and tag:
        j = dict(workload=tag, kernel=dom, hbm_bytes_per_launch=summary[dom]["hbm_bytes"], **summary[dom],
                 avg_kernel_us=stats.get(dom, {}).get("avg_us"))
        open(out + ".json", "w").write(json.dumps(j, indent=1) + "\n")


if __name__ == "__main__":
    main()
