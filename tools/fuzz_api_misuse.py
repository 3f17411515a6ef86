#!/usr/bin/env python3
"""C-ABI misuse sweep on the host-callable entry points (no GPU): NULL pointers, negative sizes, unknown names, absurd
codes.  Every call must come back with a negative ATL_E_* code (or succeed harmlessly) - never crash.  Run against the
sanitizer build like tools/fuzz_reader.py."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from atlite_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    n = 0
    bad = []

    def call(name, *args, ok_allowed=False):
        nonlocal n
        rc = getattr(lib, name)(*args)
        n += 1
        if rc == 0 and not ok_allowed:
            bad.append((name, args))
        return rc

    d = np.zeros(8)
    i64 = C.c_int64()
    # file reader
    h = C.c_void_p()
    call("atl_nc_open", None, C.byref(h))
    call("atl_nc_open", b"/nonexistent/file.nc", C.byref(h))
    call("atl_nc_open", str(ROOT / "tests" / "golden" / "nc" / "cutout_nc4.nc").encode(), None)
    assert lib.atl_nc_open(str(ROOT / "tests" / "golden" / "nc" / "cutout_nc4.nc").encode(), C.byref(h)) == 0
    info = _lib.NcVar()
    call("atl_nc_inquire", h, b"no_such_variable", C.byref(info))
    call("atl_nc_inquire", h, None, C.byref(info))
    call("atl_nc_inquire", None, b"temperature", C.byref(info))
    call("atl_nc_inquire", h, b"temperature", None)
    call("atl_nc_list", None, None, 0, C.byref(i64))
    call("atl_nc_list", h, None, 0, None, ok_allowed=True)  # nothing asked for: a no-op
    call("atl_nc_dims", h, b"no_such_variable", None, 0, C.byref(i64))
    call("atl_nc_att_text", h, b"temperature", None, None, 0, C.byref(i64))
    call("atl_nc_att_double", h, b"temperature", b"units", None, 5, C.byref(i64))
    call("atl_nc_read_host", h, b"temperature", -1, 2, d.ctypes.data)
    call("atl_nc_read_host", h, b"temperature", 0, 10**9, d.ctypes.data)
    call("atl_nc_read_host", h, b"temperature", 0, 1, None)
    call("atl_nc_read_host", h, b"no_such_variable", 0, 1, d.ctypes.data)
    call("atl_nc_read_host", None, b"temperature", 0, 1, d.ctypes.data)
    assert lib.atl_nc_close(h) == 0
    call("atl_nc_close", None, ok_allowed=True)
    # inflate
    call("atl_inflate_probe", None, 10, d.ctypes.data, 8, 2, None)
    call("atl_inflate_probe", d.ctypes.data, 8, None, 8, 2, None)
    call("atl_inflate_probe", d.ctypes.data, 8, d.ctypes.data, 8, 99, None)
    # math / converter probes
    call("atl_math_probe_host", 99, d.ctypes.data, 4, d.ctypes.data)
    call("atl_math_probe_host", 0, None, 4, d.ctypes.data)
    call("atl_math_probe_host", 0, d.ctypes.data, -4, d.ctypes.data)
    call("atl_wind_interp_host", None, None, 3, d.ctypes.data, 4, d.ctypes.data)
    call("atl_wind_interp_host", d.ctypes.data, d.ctypes.data, 0, d.ctypes.data, 4, d.ctypes.data)
    call("atl_wind_interp_host", d.ctypes.data, d.ctypes.data, 10**6, d.ctypes.data, 4, d.ctypes.data)
    wp = _lib.WindParams(7, 80.0, 100.0, 2, d.ctypes.data_as(_lib.c_double_p), d.ctypes.data_as(_lib.c_double_p))
    call("atl_wind_probe_host", C.byref(wp), 4, d.ctypes.data, d.ctypes.data, d.ctypes.data)
    call("atl_wind_probe_host", None, 4, d.ctypes.data, d.ctypes.data, d.ctypes.data)
    pp = _lib.PvParams()
    call("atl_pv_probe_host", None, 0, 4, None, d.ctypes.data)
    call("atl_pv_probe_host", C.byref(pp), 7, 4, None, d.ctypes.data)
    # plan builder / polygons
    ip = np.array([0, 1], dtype=np.int64)
    ix = np.array([0], dtype=np.int32)
    call("atl_agg_check_host", 1, 4, 0, None, ix.ctypes.data, d.ctypes.data, None, None, C.byref(i64))
    call("atl_agg_check_host", -1, 4, 0, ip.ctypes.data, ix.ctypes.data, d.ctypes.data, None, None, C.byref(i64))
    call("atl_agg_check_host", 1, -4, 0, ip.ctypes.data, ix.ctypes.data, d.ctypes.data, None, None, C.byref(i64))
    call("atl_agg_check_host", 1, 4, 0, ip.ctypes.data, None, None, None, None, C.byref(i64))
    call("atl_agg_check_host", 70000, 4, 0, ip.ctypes.data, ix.ctypes.data, d.ctypes.data, None, None, C.byref(i64))
    call("atl_agg_selfcheck", 10, 3, 16, C.byref(i64), C.byref(i64), C.byref(i64))
    call("atl_agg_selfcheck", -10, 0, 128, C.byref(i64), C.byref(i64), C.byref(i64))
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for fn in ("atl_indicator_polygons", "atl_indicator_polygons_integral_host"):
        call(fn, 1, None, 1, None, None, None, 4, 4, 0.0, 1.0, 0.0, 1.0, C.byref(p1), C.byref(p2), C.byref(p3))
        call(fn, 1, ip.ctypes.data, 1, ip.ctypes.data, None, d.ctypes.data, 0, 4, 0.0, 1.0, 0.0, 1.0, C.byref(p1), C.byref(p2), C.byref(p3))
        call(fn, 1, ip.ctypes.data, 1, ip.ctypes.data, None, d.ctypes.data, 4, 4, 0.0, -1.0, 0.0, 1.0, C.byref(p1), C.byref(p2), C.byref(p3))
        call(fn, -1, ip.ctypes.data, 1, ip.ctypes.data, None, d.ctypes.data, 4, 4, 0.0, 1.0, 0.0, 1.0, C.byref(p1), C.byref(p2), C.byref(p3))
        call(fn, 1, ip.ctypes.data, 1, ip.ctypes.data, None, d.ctypes.data, 1 << 20, 1 << 20, 0.0, 1.0, 0.0, 1.0, C.byref(p1), C.byref(p2), C.byref(p3))
    # contexts without a device: creation must fail with an error, not crash; NULL contexts everywhere
    ctx = C.c_void_p()
    call("atl_create", 0, None, C.byref(ctx), ok_allowed=True)
    if ctx.value:
        lib.atl_destroy(ctx)
    call("atl_create", 0, None, None)
    call("atl_create", -5, None, C.byref(ctx))
    call("atl_destroy", None, ok_allowed=True)
    call("atl_spmm_csr", None, None, None, 1, 1, 0, None, 1)
    call("atl_pv_convert", None, None, None, 1, 1, 0, None)
    call("atl_wind_convert", None, None, None, 1, 1, 0, None)
    call("atl_runoff_convert", None, None, None, 1, 1, 0, None)
    call("atl_host_register", None, 16)
    call("atl_host_free", None, ok_allowed=True)
    assert not bad, bad
    print(f"{n} misuse calls: every one refused with an error code (or harmless), no crash")


if __name__ == "__main__":
    main()
