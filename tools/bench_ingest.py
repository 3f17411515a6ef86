#!/usr/bin/env python3
"""
End-to-end rate of the pv path fed from a cutout FILE, on the GPU box:

  python tools/bench_ingest.py [--T 720 --Y 200 --X 200 --chunks 24,100,100 --dtype f4]

1. writes an ERA5-shaped NetCDF-4 cutout with h5py under the conda interpreter (shuffle + deflate),
2. times ``Cutout(path).pv(shapes)`` from the file (cold-ish and warm page cache),
3. times the same conversion from pinned in-memory float32 and float64 arrays (PCIe-only baselines),
4. times the library's host reader and h5py itself (what the reference's xarray/netCDF4 stack does:
   single-threaded inflate + unshuffle on the CPU) reading the same 7 variables.
Prints one line per leg.
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONDA = "/opt/conda/bin/python3.9"
PV_VARS = ["influx_direct", "influx_diffuse", "influx_toa", "albedo", "temperature", "solar_altitude", "solar_azimuth"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=720)
    ap.add_argument("--Y", type=int, default=200)
    ap.add_argument("--X", type=int, default=200)
    ap.add_argument("--chunks", default="24,100,100")
    ap.add_argument("--dtype", default="f4")
    ap.add_argument("--shapes", type=int, default=100)
    ap.add_argument("--keep", default=None, help="write the file here and keep it")
    ap.add_argument("--threads", type=int, default=0, help="threads that deflate the chunks while the file is written (0: CPUs)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-inflate leg (A/B runs of the device decoder)")
    ap.add_argument("--default-policy", action="store_true", help="let the library choose host / device inflate (default: device forced)")
    ap.add_argument("--quick", action="store_true", help="only the from-file legs (device / host inflate) and the stage split")
    a = ap.parse_args()
    ct, cy, cx = (int(v) for v in a.chunks.split(","))
    tmp = tempfile.mkdtemp(prefix="atl_ingest_", dir="/tmp")
    path = a.keep or os.path.join(tmp, "cutout.nc")
    t0 = time.perf_counter()
    if not (a.keep and os.path.exists(path)):  # --keep: an existing file is reused
        subprocess.run([CONDA, os.path.join(ROOT, "tests/golden/make_nc_fixtures.py"), "--cutout", path, str(a.T), str(a.Y),
                        str(a.X), str(ct), str(cy), str(cx), a.dtype, "11", str(a.threads or len(os.sched_getaffinity(0)))], check=True)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB on disk in {time.perf_counter() - t0:.1f} s "
          f"(T={a.T} {a.Y}x{a.X}, chunks {a.chunks}, {a.dtype}, 11 cubes)", flush=True)

    import atlite_amd as aa
    from atlite_amd import _lib, gis, io
    from atlite_amd.device import default_context

    ctx = default_context()
    cells = a.T * a.Y * a.X
    t0 = time.perf_counter()
    cf = aa.Cutout(path)
    t_open = time.perf_counter() - t0
    x, y = cf.coords["x"], cf.coords["y"]
    M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(a.shapes, cf.bounds, seed=0))
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    disk = sum(cf.data.file.variables[v].stored_bytes for v in PV_VARS)
    raw = 7 * cells * np.dtype(a.dtype).itemsize
    print(f"open + parse: {t_open * 1e3:.1f} ms; pv inputs: {disk / 1e6:.1f} MB stored, {raw / 1e6:.1f} MB inflated, "
          f"{7 * cells * 8 / 1e6:.1f} MB as fp64", flush=True)

    def leg(label, fn, n=3):
        best, first = 1e30, None
        for i in range(n):
            t0 = time.perf_counter()
            r = fn()
            dt = time.perf_counter() - t0
            first = dt if first is None else first
            best = min(best, dt)
        print(f"{label:52s} first {first:7.3f} s  best {best:7.3f} s  {cells / best:10.3e} cell-steps/s  "
              f"{7 * cells * 8 / best / 1e9:7.2f} GB/s fp64-equivalent  {disk / best / 1e6:8.1f} MB/s of file", flush=True)
        return r

    import ctypes as C

    def times():
        ms = (C.c_double * 5)()
        cb, rb = C.c_int64(), C.c_int64()
        _lib.check(ctx.lib.atl_nc_ingest_times(ctx.handle, ms, C.byref(cb), C.byref(rb)))
        st = [C.c_int64() for _ in range(3)]
        _lib.check(ctx.lib.atl_nc_ingest_stats(ctx.handle, *[C.byref(x) for x in st]))
        return np.array(list(ms)), cb.value, rb.value, [x.value for x in st]

    if a.default_policy:
        os.environ.pop("ATLITE_HIP_INFLATE", None)
    else:
        os.environ["ATLITE_HIP_INFLATE"] = "device"
    ref = leg("pv from FILE (inflate on the DEVICE, one wave per chunk stream)", lambda: cf.pv(**kw).values, n=4)
    m0, c0, r0, s0 = times()
    cf.pv(**kw).values
    m1, c1, r1, s1 = times()
    dm = m1 - m0
    print(f"  one warm call, the stages of its ONE fed launch (they overlap): preads {dm[0]:.1f} ms | DMAs, first to last {dm[1]:.1f} ms "
          f"({(c1 - c0) / max(dm[1], 1e-9) / 1e6:.1f} GB/s) | k_inflate incl. its waits, Adler-32 and unpack {dm[2]:.1f} ms "
          f"({(r1 - r0) / max(dm[2], 1e-9) / 1e6:.1f} GB/s of output, {s1[0] - s0[0]} streams, {s1[2] - s0[2]} redone on the host) | "
          f"unpack of never-written chunks {dm[4]:.1f} ms | segments decoded side by side {int(dm[3])}", flush=True)
    if a.no_host:
        import hashlib
        print("result sha1", hashlib.sha1(np.ascontiguousarray(ref).tobytes()).hexdigest(), flush=True)
    else:
        os.environ["ATLITE_HIP_INFLATE"] = "host"
        ref_h = leg("pv from FILE (inflate on host threads, decode on GPU)", lambda: cf.pv(**kw).values)
        assert np.array_equal(ref, ref_h), "device-inflate and host-inflate results differ"
        import hashlib
        print("device-inflate result == host-inflate result: bit-identical; sha1", hashlib.sha1(np.ascontiguousarray(ref).tobytes()).hexdigest(), flush=True)
    if a.quick:
        if not a.keep:
            os.remove(path)
        return
    for nt in (1, 16, 32, 64, 128):
        os.environ["ATLITE_HIP_IO_THREADS"] = str(nt)
        leg(f"  same, ATLITE_HIP_IO_THREADS={nt}", lambda: cf.pv(**kw).values, n=2)
    os.environ.pop("ATLITE_HIP_IO_THREADS")
    os.environ.pop("ATLITE_HIP_INFLATE")

    f = cf.data.file
    t0 = time.perf_counter()
    host = {v: f.read(v) for v in PV_VARS}
    t_host = time.perf_counter() - t0
    print(f"{'library host reader (threads), 7 vars -> fp64':52s} {t_host:7.3f} s  {7 * cells * 8 / t_host / 1e9:7.2f} GB/s fp64-equivalent")
    coords = {k: cf.coords[k] for k in ("time", "y", "x")}
    os.environ["ATLITE_HIP_STREAM"] = "1"
    d32 = aa.Dataset({k: v.astype(np.float32) for k, v in host.items()}, coords, chunked=True).pin()
    r32 = leg("pv from pinned float32 arrays (widened on GPU)", lambda: aa.Cutout(d32).pv(**kw).values)
    del d32
    d64 = aa.Dataset(host, coords, chunked=True).pin()
    r64 = leg("pv from pinned float64 arrays (plain DMA)", lambda: aa.Cutout(d64).pv(**kw).values)
    assert np.array_equal(ref, r64) and (a.dtype != "f4" or np.array_equal(ref, r32)), "file / memory results differ"
    print("file-backed result == in-memory result: bit-identical")

    code = ("import h5py, time, sys\nt0=time.perf_counter()\nf=h5py.File(sys.argv[1],'r')\n"
            "n=0\nfor v in sys.argv[2:]:\n    a=f[v][...].astype('f8'); n+=a.nbytes\n"
            "print(time.perf_counter()-t0, n)")
    out = subprocess.run([CONDA, "-c", code, path] + PV_VARS, capture_output=True, text=True)
    if out.returncode == 0:
        dt, n = out.stdout.split()
        print(f"{'h5py (libhdf5, 1 thread), 7 vars -> fp64 on host':52s} {float(dt):7.3f} s  {float(n) / float(dt) / 1e9:7.2f} GB/s "
              f"fp64-equivalent  (the reference's reader stack; its NumPy conversion comes on top)")
    if not a.keep:
        os.remove(path)


if __name__ == "__main__":
    main()
