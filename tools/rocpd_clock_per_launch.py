#!/usr/bin/env python3
"""Per-dispatch duration and effective shader clock from a  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace  database
(GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel duration; MI355X_MICROARCH.md "DVFS give-back").
  usage: rocpd_clock_per_launch.py <dir with the .db> [kernel-name substring ...]"""
import glob
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
    pats = sys.argv[2:] or ["k_fused"]
    con = sqlite3.connect(f)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kcols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    print("# counters_collection columns:", cols)
    print("# kernels columns:", kcols)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection order by dispatch_id").fetchall() \
        if "start" in cols else None
    if rows is None:
        rows = con.execute("select c.dispatch_id, c.kernel_name, c.counter_name, c.value, k.start, k.end from counters_collection c "
                           "join kernels k on k.dispatch_id = c.dispatch_id order by c.dispatch_id").fetchall()
    per = {}
    for did, kn, cn, val, s, e in rows:
        d = per.setdefault(did, dict(kernel=short(kn), start=s, end=e))
        d[cn] = d.get(cn, 0.0) + val  # one row per XCD / instance: sum
    print(f"{'dispatch':>8s} {'kernel':60s} {'ms':>8s} {'GRBM_GUI_ACTIVE':>16s} {'GHz (sum/8/dur)':>16s}")
    for did in sorted(per):
        d = per[did]
        if not any(p in d["kernel"] for p in pats):
            continue
        dur = (d["end"] - d["start"]) * 1e-9
        g = d.get("GRBM_GUI_ACTIVE", 0.0)
        print(f"{did:8d} {d['kernel'][:60]:60s} {dur * 1e3:8.3f} {g:16.4g} {g / 8 / dur / 1e9 if dur > 0 else 0:16.3f}")


if __name__ == "__main__":
    main()
