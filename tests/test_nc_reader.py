"""
Native NetCDF-4 / HDF5 container reader (atlite_amd/csrc/atl_h5.cpp, atl_nc_* in
include/atlite_hip.h) - host-only entry points, no GPU needed.  Fixtures: tests/golden/nc/*.nc
written by h5py under the conda interpreter (tests/golden/make_nc_fixtures.py) with the
expected fp64 values beside them; when that interpreter is present, random shapes / chunkings
are generated on the fly as well.  SURVEY.md section 8 row f-4.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pandas as pd
import pytest

from atlite_amd import _lib, io

NC = os.path.join(os.path.dirname(__file__), "golden", "nc")
CONDA = "/opt/conda/bin/python3.9"
MAKE = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")


def _have_h5py():
    try:
        return subprocess.run([CONDA, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        return False


@pytest.mark.parametrize("name", ["cutout_nc4", "cutout_earliest", "cutout_latest", "cutout_many", "userblock",
                                  "cutout_unlimited", "cutout_unlimited_y", "cutout_unlimited_ty"])
def test_fixture_values(name):
    """Every variable of every container flavour decodes to the values h5py wrote."""
    f = io.NcFile(f"{NC}/{name}.nc")
    exp = np.load(f"{NC}/{name}.npz")
    assert set(exp.files) == set(f.variables)
    for v in exp.files:
        got = f.read(v)
        assert got.shape == exp[v].shape, v
        assert np.array_equal(got, exp[v], equal_nan=True), v
    f.close()


def test_partial_rows_and_edge_chunks():
    f = io.NcFile(f"{NC}/cutout_nc4.nc")
    exp = np.load(f"{NC}/cutout_nc4.npz")
    T = exp["runoff"].shape[0]
    for name in ("runoff", "albedo", "soil_temperature", "roughness", "u16cube", "count_i32"):
        for t0, n in ((0, 1), (3, 9), (9, 2), (10, 10), (T - 1, 1), (5, T - 5), (7, 0)):
            got = f.read(name, t0, n)
            assert np.array_equal(got, exp[name][t0:t0 + n], equal_nan=True), (name, t0, n)
    with pytest.raises(ValueError, match="outside"):
        f.read("runoff", T - 1, 2)
    with pytest.raises(KeyError):
        f.read("nope")


def test_metadata():
    f = io.NcFile(f"{NC}/cutout_nc4.nc")
    v = f.variables["runoff"]
    assert v.dtype == "int16" and v.shape == (23, 7, 9) and v.chunks == (10, 4, 5)
    assert v.dims == ("time", "y", "x")  # resolved through DIMENSION_LIST object references
    assert v.layout == "chunked" and v.shuffle and v.fletcher32 and v.deflate == 4
    assert v.scale_factor == 1.5e-4 and v.add_offset == 4.25 and v.fill_value == -32767.0
    assert v.n_chunks == 3 * 2 * 2
    a = f.variables["albedo"]
    assert a.big_endian and a.dtype == "float32" and a.missing_value == float(np.float32(9.96921e36))
    assert f.variables["height"].layout == "contiguous" and f.variables["height"].dims == ("y", "x")
    assert f.variables["temperature"].deflate == 9
    # attributes: fixed strings, variable-length strings, scalars, arrays; dense storage (> 8 attributes)
    assert f.attr("time", "units") == "hours since 1900-01-01 00:00:00.0"
    assert f.attr(None, "module") == "era5"
    assert f.attr(None, "vlen_note") == "written as a variable-length string"
    assert f.attr(None, "dx") == 0.25
    assert f.attr("temperature", "extra_11") == 11 * 1.25
    assert np.array_equal(f.attr("temperature", "ints"), np.arange(5.0))
    assert f.attr("temperature", "absent") is None
    # old-style file: same answers through symbol tables / v1 headers
    g = io.NcFile(f"{NC}/cutout_earliest.nc")
    assert g.variables["runoff"].dims == ("time", "y", "x")
    assert g.attr("temperature", "extra_11") == 11 * 1.25


def test_many_links_two_level_index():
    f = io.NcFile(f"{NC}/cutout_many.nc")
    assert sum(n.startswith("aux_") for n in f.variables) == 160
    assert np.array_equal(f.read("aux_0159"), 159.0 + np.arange(3))


def test_unsupported_and_corrupt_inputs(tmp_path):
    f = io.NcFile(f"{NC}/unlimited_latest.nc")  # an extensible-array chunk index (refused until round 3)
    assert np.array_equal(f.read("influx"), np.ones((6, 4, 5)))
    assert np.array_equal(f.read("y"), np.arange(4.0))
    # a chunk index type this reader does not know (the index-type byte of the layout message patched to 9): refused
    raw = bytearray(open(f"{NC}/unlimited_latest.nc", "rb").read())
    i = raw.index(b"EAHD")  # the layout message points at it: version 4, class 2, flags, rank 4, 1-byte dims 2 4 5 4, index type 4
    hits = [k for k in range(len(raw) - 10) if bytes(raw[k:k + 2]) == b"\x04\x02" and raw[k + 3] == 4 and bytes(raw[k + 5:k + 9]) == b"\x02\x04\x05\x04"
            and raw[k + 9] == 4]
    assert len(hits) == 1 and i > 0
    raw[hits[0] + 9] = 9
    q = tmp_path / "unknown_index.nc"
    q.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match="chunk index"):
        io.NcFile(q).read("influx")
    with pytest.raises(ValueError, match="cannot open"):
        io.NcFile(tmp_path / "missing.nc")
    p = tmp_path / "text.nc"
    p.write_bytes(b"not an hdf5 file at all, but long enough to pass the size check........")
    with pytest.raises(ValueError, match="not an HDF5"):
        io.NcFile(p)
    import scipy.io

    p3 = tmp_path / "classic.nc"
    with scipy.io.netcdf_file(str(p3), "w") as nc3:
        nc3.createDimension("x", 4)
        nc3.createVariable("x", "d", ("x",))[:] = np.arange(4.0)
    with pytest.raises(NotImplementedError, match="NetCDF-3"):
        io.NcFile(p3)
    # truncations and bit flips must give errors (or still-valid data), never a crash
    raw = open(f"{NC}/cutout_nc4.nc", "rb").read()
    rng = np.random.default_rng(0)
    for k in range(40):
        b = bytearray(raw)
        if k % 2:
            b = b[: int(rng.integers(64, len(raw)))]
        else:
            for pos in rng.integers(0, min(len(raw), 20000), size=8):
                b[int(pos)] ^= 0xFF
        q = tmp_path / f"bad{k}.nc"
        q.write_bytes(bytes(b))
        try:
            g = io.NcFile(q)
            for name, var in list(g.variables.items())[:6]:
                if var.dtype and 1 <= var.ndim <= 3 and np.prod(var.shape) < 10**6:
                    try:
                        g.read(name)
                    except (ValueError, NotImplementedError, MemoryError):
                        pass
            g.close()
        except (ValueError, NotImplementedError, MemoryError):
            pass


def test_decode_time():
    t = io.decode_time([990552, 990553.5], "hours since 1900-01-01 00:00:00.0", "proleptic_gregorian")
    assert list(t) == [pd.Timestamp("2013-01-01 00:00"), pd.Timestamp("2013-01-01 01:30")]
    t = io.decode_time(np.arange(3), "days since 2013-01-01", None)
    assert list(t) == list(pd.date_range("2013-01-01", periods=3, freq="D"))
    t = io.decode_time([60], "minutes since 2013-01-01T00:00:00+00:00")
    assert t[0] == pd.Timestamp("2013-01-01 01:00")
    with pytest.raises(ValueError):
        io.decode_time([0], "fortnights since 2013-01-01")
    with pytest.raises(NotImplementedError):
        io.decode_time([0], "hours since 2013-01-01", "360_day")


def test_open_cutout_dataset():
    """Dataset view of a cutout file: coordinates, lazy cubes, eager static fields (no GPU involved)."""
    ds = io.open_cutout(f"{NC}/cutout_small_f32.nc")
    assert ds.chunked  # like the reference's dask-backed file cutouts (cutout.py:143)
    assert ds.sizes == {"time": 48, "y": 9, "x": 12}
    assert ds.coords["time"][0] == pd.Timestamp("2013-01-01 00:00") and ds.coords["time"][-1] == pd.Timestamp(
        "2013-01-02 23:00")
    assert np.allclose(ds.coords["x"], -5.0 + 0.25 * np.arange(12)) and np.array_equal(ds.coords["lon"], ds.coords["x"])
    assert "influx_direct" in ds and "soil temperature" in ds and "lon" not in ds
    la = ds["temperature"]
    assert la.dims == ("time", "y", "x") and la.shape == (48, 9, 12) and getattr(la.data, "is_file_array", False)
    assert isinstance(ds["height"].data, np.ndarray) and ds["height"].shape == (9, 12)
    v = la.values  # host materialisation goes through atl_nc_read_host
    assert v.dtype == np.float64 and 268 <= v.min() and v.max() <= 298
    assert np.array_equal(la.data[5:9], v[5:9])
    assert ds.attrs["module"] == "era5"
    from atlite_amd import Cutout

    c = Cutout(f"{NC}/cutout_small_f32.nc")
    assert c.shape == (9, 12) and abs(c.dx - 0.25) < 1e-12
    with pytest.raises(NotImplementedError):
        Cutout("/nonexistent/dir/new-cutout.nc")


@pytest.mark.skipif(not _have_h5py(), reason="needs the conda interpreter with h5py to write random files")
@pytest.mark.parametrize("seed", range(12))
def test_random_files(tmp_path, seed):
    rng = np.random.default_rng(100 + seed)
    T, Y, X = (int(v) for v in rng.integers(1, 30, size=3))
    ct, cy, cx = (int(rng.integers(1, d + 3)) for d in (T, Y, X))
    ct, cy, cx = min(ct, T), min(cy, Y), min(cx, X)
    libver = ["v108", "earliest", "latest"][seed % 3]
    unlimited = ["", "0", "1", "02", "012", "2"][seed % 6]  # earliest / v108: the v1 B-tree whatever the bounds; latest: EA / v2 B-tree
    path = tmp_path / "case.nc"
    r = subprocess.run([CONDA, MAKE, "--case", str(path), str(T), str(Y), str(X), str(ct), str(cy), str(cx), libver,
                        str(seed % 2), str(seed), unlimited], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    exp = np.load(tmp_path / "case.npz")
    for v in exp.files:
        assert np.array_equal(f.read(v), exp[v], equal_nan=True), (v, T, Y, X, ct, cy, cx, libver)


def test_c_struct_layout_matches_ctypes():
    """atl_nc_var in the header and _lib.NcVar agree (size checked against the C compiler)."""
    import tempfile

    root = os.path.dirname(os.path.dirname(__file__))
    src = '#include <stdio.h>\n#include "atlite_hip.h"\nint main(void){printf("%zu %zu %zu", sizeof(atl_nc_var),' \
          ' __builtin_offsetof(atl_nc_var, scale_factor), __builtin_offsetof(atl_nc_var, n_chunks));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(f"{d}/t.c", "w").write(src)
        subprocess.run(["gcc", "-std=c99", f"-I{root}/include", f"{d}/t.c", "-o", f"{d}/t"], check=True)
        size, off_scale, off_n = (int(v) for v in subprocess.run([f"{d}/t"], capture_output=True, text=True).stdout.split())
    assert size == C.sizeof(_lib.NcVar)
    assert off_scale == _lib.NcVar.scale_factor.offset and off_n == _lib.NcVar.n_chunks.offset


def _inflate(comp, n, which):
    lib = _lib.load()
    dst = np.zeros(max(n, 1), np.uint8)
    src = np.frombuffer(comp, np.uint8) if len(comp) else np.zeros(1, np.uint8)
    rc = lib.atl_inflate_probe(src.ctypes.data, len(comp), dst.ctypes.data, n, which, None)
    return rc, dst[:n].tobytes()


def test_fast_inflate_matches_zlib():
    """The library's own DEFLATE decoder (atl_inflate.cpp) against zlib on every block type, code shape
    and match pattern zlib can be made to emit; the product path (fast, zlib on any doubt) must reach
    zlib's verdict on corrupted streams and never crash."""
    import zlib

    rng = np.random.default_rng(0)
    f = (np.round(rng.random(30000) * 300 * 4096) / 4096).astype(np.float32)
    data = [b"", b"a", b"abc" * 1000, bytes(70000), rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(),
            rng.integers(0, 4, 100000, dtype=np.uint8).tobytes(), f.tobytes(),
            f.view(np.uint8).reshape(-1, 4).T.copy().tobytes(),  # what the shuffle filter hands to deflate
            b"the quick brown fox jumps over the lazy dog " * 2000,
            bytes(rng.integers(97, 123, 100000, dtype=np.uint8))]
    data += [rng.integers(0, int(rng.integers(1, 256)), int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
             for _ in range(12)]
    # periodic data with every period 1..9: the small-distance match copies
    data += [bytes((i % p) * 7 % 256 for i in range(5000)) for p in range(1, 10)]
    n = 0
    for d in data:
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY):
                for mem in (1, 9):
                    co = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strat)
                    comp = co.compress(d) + co.flush()
                    rc, out = _inflate(comp, len(d), 0)
                    assert rc == 0 and out == d, (len(d), level, strat, mem)
                    rc, out = _inflate(comp, len(d), 3)  # the device decoder's serial half on the host (atl_inflate_dev.h)
                    assert rc == 0 and out == d, ("device decoder emulation", len(d), level, strat, mem)
                    n += 1
    assert n == len(data) * 32
    # wrong expected size, truncated and corrupted streams
    d = data[7]
    comp = zlib.compress(d, 6)
    assert _inflate(comp, len(d) + 1, 0)[0] != 0 and _inflate(comp, len(d) - 1, 0)[0] != 0
    assert _inflate(comp, len(d) + 1, 2)[0] != 0
    for k in range(400):
        b = bytearray(comp)
        for pos in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
            b[int(pos)] = int(rng.integers(0, 256))
        if k % 5 == 0:
            b = b[: int(rng.integers(2, len(b)))]
        b = bytes(b)
        try:
            z = zlib.decompress(b)
            ok = len(z) == len(d)
        except Exception:
            ok = False
        rc, out = _inflate(b, len(d), 2)
        assert (rc == 0) == ok, k
        if ok:
            assert out == z
        _inflate(b, len(d), 0)  # the fast decoder alone: any verdict, no crash


def test_segment_scheme_matches_zlib():
    """Long streams are decoded block by block on the device (atl_inflate_dev.h, "SEGMENTS"): its host emulation
    (atl_inflate_probe(which = 4): the same finder tests, count / decode passes, markers and resolve as the kernels) against
    zlib on streams of many blocks - every payload kind x level, with small windows and memLevels (other block sizes), with
    stored and fixed blocks in between - and on corrupted ones (any error, never a wrong accept, no crash)."""
    import ctypes as C
    import zlib

    lib = _lib.load()

    def run(comp, n):
        dst = np.zeros(max(n, 1), np.uint8)
        src = np.frombuffer(comp, np.uint8)
        nseg = C.c_int64()
        rc = lib.atl_inflate_probe(src.ctypes.data, len(comp), dst.ctypes.data, n, 4, C.byref(nseg))
        return rc, dst[:n].tobytes(), int(nseg.value)

    rng = np.random.default_rng(5)
    n = 600000
    vocab = [rng.integers(97, 123, int(k), dtype=np.uint8) for k in rng.integers(2, 14, 500)]
    f = (np.cumsum(rng.standard_normal(n // 4)) * 50).astype("<f4")
    data = {
        "planes": np.ascontiguousarray(f.view(np.uint8).reshape(-1, 4).T).tobytes(),
        "words": np.concatenate([vocab[i] for i in rng.zipf(1.3, size=n // 5) % len(vocab)])[:n].tobytes(),
        "few": rng.integers(0, 4, n, dtype=np.uint8).tobytes(),
        "runs": np.repeat(rng.integers(0, 256, n // 10, dtype=np.uint8), rng.geometric(0.1, n // 10))[:n].tobytes(),
        "mixed": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes() + bytes(100000) + rng.integers(0, 3, 200000, dtype=np.uint8).tobytes() + b"ab" * 9,
    }
    split = 0
    for name, d in data.items():
        for level, wbits, mem in ((1, 15, 8), (6, 15, 8), (9, 15, 9), (6, 9, 1), (4, 12, 4)):
            co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem)
            comp = co.compress(d[: len(d) // 2]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[len(d) // 2:]) + co.flush()
            rc, out, nseg = run(comp, len(d))
            assert rc == 0 and out == d, (name, level, wbits, mem)
            split += nseg > 2
    assert split >= 15  # most of them really are decoded in several segments
    comp = zlib.compress(data["words"], 6)
    d = data["words"]
    accepted = 0
    for k in range(60):
        b = bytearray(comp)
        for pos in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
            b[int(pos)] ^= 1 << int(rng.integers(8))
        if k % 6 == 0:
            b = b[: int(rng.integers(2, len(b)))]
        b = bytes(b)
        try:
            z = zlib.decompress(b)
            ok = len(z) == len(d)
        except Exception:
            ok = False
        rc, out, _ = run(b, len(d))
        assert rc != 0 or (ok and out == z), k
        accepted += rc == 0
    assert accepted <= 3


def test_isel_time_is_lazy():
    """A rank's time shard of a file-backed cutout: coordinates sliced, variables still on disk."""
    ds = io.open_cutout(f"{NC}/cutout_small_f32.nc")
    sub = ds.isel_time(10, 31)
    assert sub.sizes == {"time": 21, "y": 9, "x": 12} and sub.coords["time"][0] == ds.coords["time"][10]
    fa = sub["temperature"].data
    assert fa.is_file_array and fa.shape == (21, 9, 12) and fa.row0 == 10
    full = np.asarray(ds["temperature"].data)
    assert np.array_equal(np.asarray(fa), full[10:31])
    assert np.array_equal(fa[3:7], full[13:17])
    assert np.array_equal(np.asarray(fa.slab(5, 9)), full[15:19])
    assert sub["height"].data is ds["height"].data  # static fields are shared
    with pytest.raises(IndexError):
        ds.isel_time(40, 50)
    # in-memory datasets: views, no copies
    from atlite_amd import Dataset

    a = np.arange(4 * 2 * 3, dtype=float).reshape(4, 2, 3)
    m = Dataset({"runoff": a}, dict(time=pd.date_range("2013-01-01", periods=4, freq="h"), y=[0.0, 1.0], x=[0.0, 1.0, 2.0]))
    v = m.isel_time(1, 3)["runoff"].data
    assert v.shape == (2, 2, 3) and np.shares_memory(v, a)


@pytest.mark.skipif(not _have_h5py(), reason="needs the conda interpreter with h5py to write the files")
@pytest.mark.parametrize("args,n_chunks", [(("700", "6", "7", "1", "6", "7", "v108", "1", "5"), 700),
                                           (("40", "30", "31", "1", "2", "3", "latest", "1", "6"), 6600),
                                           (("3000", "4", "6", "1", "2", "6", "latest", "1", "7", "0"), 6000),
                                           (("50", "24", "20", "1", "2", "3", "latest", "1", "8", "1"), 4200),
                                           (("60", "20", "12", "2", "1", "4", "latest", "0", "9", "01"), 1800)])
def test_many_chunks(tmp_path, args, n_chunks):
    """One time step per chunk (netCDF-C's default for an unlimited time axis): a multi-level v1 chunk
    B-tree; libver=latest with thousands of chunks: a paged fixed-array index; with an unlimited axis an extensible array
    deep into its super blocks and paged data blocks (time first, and swizzled: y unlimited); two unlimited axes: a
    multi-level v2 B-tree."""
    path = tmp_path / "many.nc"
    r = subprocess.run([CONDA, MAKE, "--case", str(path), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    exp = np.load(tmp_path / "many.npz")
    assert f.variables["temperature"].n_chunks == n_chunks
    for v in exp.files:
        assert np.array_equal(f.read(v), exp[v], equal_nan=True), v
    t0 = int(args[0]) // 2
    assert np.array_equal(f.read("runoff", t0, 3), exp["runoff"][t0:t0 + 3], equal_nan=True)


def test_corrupted_size_fields_are_refused(tmp_path):
    """Findings of the long corruption fuzz (tools/fuzz_reader.py under ASan + UBSan, 300 000 files): a fixed-array page
    size of 2^90 elements (shift past 64 bits), fractal-heap doubling tables whose row sizes leave 64 bits, chunk
    dimensions whose product is 4 GiB or more (a buffer of that size per chunk, allocated on worker threads), dataspaces
    of more than 2^48 elements.  All are format errors now, before anything is shifted, multiplied or allocated."""
    raw = open(f"{NC}/cutout_latest.nc", "rb").read()
    # every fixed-array header of the file (one per chunked variable): page bits 90 / 0
    for bits in (90, 0):
        b = bytearray(raw)
        pos = b.find(b"FAHD")
        while pos >= 0:
            b[pos + 7] = bits
            pos = b.find(b"FAHD", pos + 1)
        q = tmp_path / f"fahd{bits}.nc"
        q.write_bytes(bytes(b))
        with pytest.raises(ValueError, match="fixed array"):
            io.NcFile(q)
    b = bytearray(raw)  # fractal heap: 40 000 rows in the root indirect block / a 2^200-byte heap
    h = b.find(b"FRHP")
    fixed = 14 + 10 * 8 + 2 * 8
    rows_at = h + fixed + 2 + 2 * 8 + 4 + 8
    bits_at = h + fixed + 2 + 2 * 8
    for at, val in ((rows_at, 40000), (bits_at, 200)):
        c = bytearray(b)
        c[at:at + 2] = int(val).to_bytes(2, "little")
        q = tmp_path / f"frhp{val}.nc"
        q.write_bytes(bytes(c))
        try:
            g = io.NcFile(q)  # (the checksum of the header may reject the file before the table is looked at)
            g.close()
        except (ValueError, NotImplementedError):
            pass
    # a seeded stretch of the fuzzer itself, against whichever library the suite runs on (the sanitizer build included)
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_reader.py"), "400", "11"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "no crash" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
