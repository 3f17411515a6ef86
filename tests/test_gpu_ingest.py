"""GPU: cutout-file ingest.  atl_nc_read_slab (host inflate -> DMA in the on-disk
dtype -> device un-shuffle / widen / CF-decode) must reproduce, bit for bit, what h5py wrote and
what the host reader returns; atl_upload_convert_async must widen every dtype exactly; conversions
run straight from a cutout FILE must equal the same conversions on in-memory fp64 copies of its
variables (and the oracle to rtol 1e-10)."""
import os

import numpy as np
import pytest

from atlite_amd import Cutout, Dataset, _lib, io
from atlite_amd._lib import check
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
NC = os.path.join(os.path.dirname(__file__), "golden", "nc")


@pytest.fixture(autouse=True, params=["host", "device"])
def inflate_mode(request, monkeypatch):
    """Every test of this file runs twice: the chunks' zlib streams inflated on host threads, and on the device (one
    wavefront per stream, atl_nc_read_slab; forced here - by default only reads of >= 1024 chunks take it)."""
    monkeypatch.setenv("ATLITE_HIP_INFLATE", request.param)
    return request.param


def ingest_stats(ctx):
    import ctypes as C

    v = [C.c_int64() for _ in range(3)]
    check(ctx.lib.atl_nc_ingest_stats(ctx.handle, *[C.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)  # device chunks, host chunks, redone on the host


def close(a, b, s=1e-12):
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=s * np.nanmax(np.abs(b)), equal_nan=True)


def slab(ctx, f, name, t0, n):
    var = f.variables[name]
    out = ctx.zeros((max(n, 1),) + var.shape[1:])
    ctx.copy_after_compute()  # zeroed on the compute stream, filled on the copy stream
    f.read_slab(ctx, name, t0, n, out.ptr)
    ctx.copy_barrier()
    ctx.sync()
    return out.numpy()[:n]


@pytest.mark.parametrize("name", ["cutout_nc4", "cutout_earliest", "cutout_latest", "cutout_unlimited", "cutout_unlimited_ty"])
def test_read_slab_matches_h5py(ctx, name):
    f = io.NcFile(f"{NC}/{name}.nc")
    exp = np.load(f"{NC}/{name}.npz")
    for v in exp.files:
        T = exp[v].shape[0]
        got = slab(ctx, f, v, 0, T)
        assert np.array_equal(got, exp[v], equal_nan=True), v
        assert np.array_equal(got, f.read(v), equal_nan=True), v
        for t0, n in ((3, 9), (9, 2), (10, 10), (T - 1, 1), (0, 1)):
            if t0 + n <= T:
                assert np.array_equal(slab(ctx, f, v, t0, n), exp[v][t0:t0 + n], equal_nan=True), (v, t0, n)
    with pytest.raises(ValueError, match="outside"):
        slab(ctx, f, "x", 5, 100)


def test_read_slab_many_calls_reuse_staging(ctx):
    """Back-to-back calls rotate through the staging slots without waiting for the GPU."""
    f = io.NcFile(f"{NC}/cutout_nc4.nc")
    exp = np.load(f"{NC}/cutout_nc4.npz")
    names = ["runoff", "albedo", "temperature", "roughness", "u16cube", "soil_temperature", "influx_direct"]
    outs = [ctx.empty(exp[n].shape) for n in names for _ in range(3)]
    k = 0
    for _ in range(3):
        for n in names:
            f.read_slab(ctx, n, 0, exp[n].shape[0], outs[k].ptr)
            k += 1
    ctx.copy_barrier()
    ctx.sync()
    for i, o in enumerate(outs):
        assert np.array_equal(o.numpy(), exp[names[i % len(names)]], equal_nan=True)


@pytest.mark.parametrize("dtype", ["float32", "float64", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32",
                                   "uint64"])
def test_upload_convert(ctx, dtype):
    rng = np.random.default_rng(3)
    n = 100003
    if dtype.startswith("float"):
        a = (rng.standard_normal(n) * 1e3).astype(dtype)
        a[:4] = [np.nan, np.inf, -np.inf, -0.0]
    else:
        info = np.iinfo(dtype)
        a = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        a[:2] = [info.min, info.max]
    for pin in (False, True):
        out = ctx.zeros(n)
        if pin:
            check(ctx.lib.atl_host_register(a.ctypes.data, a.nbytes))
        try:
            check(ctx.lib.atl_upload_convert_async(ctx.handle, out.ptr, a.ctypes.data, _lib.NC_CODES[dtype], n))
            ctx.copy_barrier()
            ctx.sync()
        finally:
            if pin:
                check(ctx.lib.atl_host_unregister(a.ctypes.data))
        assert np.array_equal(out.numpy(), a.astype(np.float64), equal_nan=True), (dtype, pin)
    with pytest.raises(ValueError, match="dtype"):
        check(ctx.lib.atl_upload_convert_async(ctx.handle, out.ptr, a.ctypes.data, 99, n))


def _memory_twin(path):
    """The file's variables as in-memory fp64 arrays: the reference point for the file-backed run."""
    f = io.NcFile(path)
    ds = io.open_cutout(path)
    data = {n: f.read(n) for n in ds.keys()}
    mem = Dataset(data, {k: ds.coords[k] for k in ("time", "y", "x")}, chunked=True)
    return Cutout(ds), Cutout(mem), data


@pytest.mark.parametrize("fname,steps", [("cutout_small_f32", None), ("cutout_small_f32", "7"), ("cutout_small_f64", "9")])
def test_conversions_from_file(monkeypatch, fname, steps):
    _conversions_from_file(monkeypatch, fname, steps)


@pytest.mark.parametrize("passes", [None, "2"])
def test_conversions_from_file_block_by_block(monkeypatch, inflate_mode, passes):
    """The same conversions with every device read decoded block by block (the scheme of long chunk streams forced on these small
    ones: one decode pass into the pool of regions, and count + decode): pv, wind, runoff, heat demand, temperatures from the file
    == from memory."""
    if inflate_mode != "device":
        pytest.skip("the device decoder's scheme")
    monkeypatch.setenv("ATLITE_HIP_INFLATE_SPLIT", "1")
    if passes:
        monkeypatch.setenv("ATLITE_HIP_SPLIT_PASSES", passes)
    _conversions_from_file(monkeypatch, "cutout_small_f32", "7")


def _conversions_from_file(monkeypatch, fname, steps):
    if steps:
        monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", steps)  # slabs that do NOT line up with the chunks
    cf, cm, data = _memory_twin(f"{NC}/{fname}.nc")
    Y, X = cf.shape
    T = len(cf.coords["time"])
    M = H.blob_matrix(4, Y, X, seed=2)
    pvkw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0})
    flat = {k: v.reshape(T, -1) if v.ndim == 3 else v.reshape(-1) for k, v in data.items()}
    ori = dict(slope=np.radians(30.0), azimuth=np.radians(180.0))
    # aggregated pv: file-backed == in-memory, both (time, index) like the reference's dask branch
    a = cf.pv(matrix=M, aggregate_time=None, **pvkw)
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    b = cm.pv(matrix=M, aggregate_time=None, **pvkw)
    monkeypatch.delenv("ATLITE_HIP_STREAM")
    assert a.dims == b.dims == ("time", "dim_0")
    np.testing.assert_array_equal(a.values, b.values)
    close(a.values.T, orc.aggregate_matrix(orc.convert_pv(flat, H.CSI, ori), M))
    # per-cell wind series, time-summed runoff, daily heat demand, temperature mean
    w = cf.wind(turbine="Vestas_V112_3MW", aggregate_time=None)
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    w2 = cm.wind(turbine="Vestas_V112_3MW", aggregate_time=None)
    monkeypatch.delenv("ATLITE_HIP_STREAM")
    np.testing.assert_array_equal(w.values, w2.values)
    close(w.values.reshape(T, -1), orc.convert_wind(flat["wnd100m"], flat["roughness"], H.V112["V"], H.V112["POW"],
                                                    H.V112["P"], 80.0, 100.0, "logarithmic"))
    for fn, kw in (("runoff", dict(matrix=M, aggregate_time="sum")), ("heat_demand", dict(matrix=M, aggregate_time=None)),
                   ("temperature", dict(matrix=M, aggregate_time="mean")), ("soil_temperature", dict(aggregate_time="sum"))):
        r = getattr(cf, fn)(**kw)
        monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
        r2 = getattr(cm, fn)(**kw)
        monkeypatch.delenv("ATLITE_HIP_STREAM")
        close(r.values, r2.values)


def test_float32_host_arrays_stream_narrow(monkeypatch):
    """float32 host variables (what xarray hands over for a real cutout) are widened on the device."""
    T, Y, X = 50, 6, 9
    ds64 = H.pv_dataset(T, Y, X, seed=5)
    ds32 = {k: v.astype(np.float32) for k, v in ds64.items()}
    up = {k: v.astype(np.float64) for k, v in ds32.items()}
    x, y = H.grid(Y, X)
    coords = dict(time=H.times(T), y=y, x=x)
    M = H.blob_matrix(3, Y, X, seed=1)
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    monkeypatch.setenv("ATLITE_HIP_STREAM", "1")
    monkeypatch.setenv("ATLITE_HIP_SLAB_STEPS", "16")
    a = Cutout(Dataset(ds32, coords)).pv(**kw).values
    b = Cutout(Dataset(up, coords)).pv(**kw).values
    np.testing.assert_array_equal(a, b)
    c = Cutout(Dataset(ds32, coords).pin()).pv(**kw).values  # pinned: DMA straight from the arrays
    np.testing.assert_array_equal(c, b)


def test_file_array_to_device(ctx):
    ds = io.open_cutout(f"{NC}/cutout_small_f32.nc")
    fa = ds["temperature"].data
    d = fa.to_device(ctx, block_bytes=4096)  # several blocks
    ctx.sync()
    assert np.array_equal(d.numpy(), np.asarray(fa))
    dev = ds.device(ctx, "temperature")  # Dataset.device() goes the same way and caches
    assert dev.shape == (48, 9 * 12) and ds.device(ctx, "temperature").ptr == dev.ptr


def test_time_shards_of_a_file_cutout(monkeypatch):
    """Two ranks' shards (Dataset.isel_time) processed one after the other == the whole run."""
    from atlite_amd.distributed import time_partition

    ds = io.open_cutout(f"{NC}/cutout_small_f32.nc")
    Y, X = 9, 12
    M = H.blob_matrix(3, Y, X, seed=4)
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    whole = Cutout(ds).pv(**kw).values
    edges = time_partition(48, 2)
    parts = [Cutout(ds.isel_time(edges[r], edges[r + 1])).pv(**kw).values for r in range(2)]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), whole)
    # heat demand: shard boundaries on calendar days
    edges = time_partition(48, 2, align=24)
    whole = Cutout(ds).heat_demand(matrix=M, aggregate_time=None).values
    parts = [Cutout(ds.isel_time(edges[r], edges[r + 1])).heat_demand(matrix=M, aggregate_time=None).values for r in range(2)]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), whole)


@pytest.mark.parametrize("seed", range(4))
def test_read_slab_random_files(ctx, tmp_path, seed):
    """Random shapes / chunkings / container flavours written by h5py on the spot (when the conda
    interpreter is there): device decode == host decode == what was written, for random row ranges."""
    import subprocess

    conda = "/opt/conda/bin/python3.9"
    make = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")
    try:
        ok = subprocess.run([conda, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        ok = False
    if not ok:
        pytest.skip("needs the conda interpreter with h5py")
    rng = np.random.default_rng(500 + seed)
    T, Y, X = (int(v) for v in rng.integers(1, 40, size=3))
    ct, cy, cx = (int(min(rng.integers(1, d + 3), d)) for d in (T, Y, X))
    libver = ["v108", "earliest", "latest"][seed % 3]
    path = tmp_path / "case.nc"
    r = subprocess.run([conda, make, "--case", str(path), str(T), str(Y), str(X), str(ct), str(cy), str(cx), libver,
                        str(seed % 2), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    exp = np.load(tmp_path / "case.npz")
    for v in exp.files:
        n0 = exp[v].shape[0]
        for _ in range(3):
            t0 = int(rng.integers(0, n0))
            n = int(rng.integers(1, n0 - t0 + 1))
            got = slab(ctx, f, v, t0, n)
            assert np.array_equal(got, exp[v][t0:t0 + n], equal_nan=True), (v, t0, n, T, Y, X, ct, cy, cx, libver)


@pytest.mark.parametrize("seed", range(3))
def test_inflate_payload_statistics(ctx, tmp_path, seed, inflate_mode):
    """Chunk streams that are NOT weather fields (tests/golden/make_nc_fixtures.py --payloads: noise, four-letter bytes,
    runs - matches that overlap themselves -, long periods - maximum-length matches -, sparse bytes in zeros, a Zipf
    dictionary - far matches -, byte planes; zlib 1 / 6 / 9): what was written comes back, and in device mode every
    stream was inflated by k_inflate, none redone on the host."""
    import subprocess

    conda = "/opt/conda/bin/python3.9"
    make = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")
    try:
        ok = subprocess.run([conda, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        ok = False
    if not ok:
        pytest.skip("needs the conda interpreter with h5py")
    path = tmp_path / "payloads.nc"
    r = subprocess.run([conda, make, "--payloads", str(path), str(40 + seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    exp = np.load(tmp_path / "payloads.npz")
    d0, h0, r0 = ingest_stats(ctx)
    n_streams = 0
    for v in exp.files:
        assert np.array_equal(slab(ctx, f, v, 0, exp[v].shape[0]), exp[v]), v
        assert np.array_equal(slab(ctx, f, v, 5, 30), exp[v][5:35]), v
        n_streams += f.variables[v].n_chunks + 5
    d1, h1, r1 = ingest_stats(ctx)
    if inflate_mode == "device":
        assert (d1 - d0, h1 - h0, r1 - r0) == (n_streams, 0, 0)
    else:
        assert (d1 - d0, h1 - h0) == (0, n_streams)


def ingest_segments(ctx):
    import ctypes as C

    ms = (C.c_double * 5)()
    a, b = C.c_int64(), C.c_int64()
    check(ctx.lib.atl_nc_ingest_times(ctx.handle, ms, C.byref(a), C.byref(b)))
    return int(ms[3])  # the count of stream segments decoded side by side


def test_long_streams_are_decoded_block_by_block(ctx, tmp_path, monkeypatch, inflate_mode):
    """Few, long chunk streams - atlite's own cutouts have (time = 100, y, x) chunks, 16 MB each for a 200 x 200 grid - are
    split at their DEFLATE block headers and the blocks decoded side by side (block finder, one decode pass into a pool of output
    regions with markers for what a block copies from its predecessors - or a count pass and a decode pass -, the chains followed
    on the host, gather / resolve): 1 MiB chunks of seven payload kinds x zlib levels, forced
    ($ATLITE_HIP_INFLATE_SPLIT=1) and by the default policy; the bytes are what was written, no stream goes back to the host, and
    the compressible kinds really are decoded in several segments each."""
    import subprocess

    if inflate_mode != "device":
        pytest.skip("the device decoder's scheme")
    conda = "/opt/conda/bin/python3.9"
    make = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")
    try:
        ok = subprocess.run([conda, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        ok = False
    if not ok:
        pytest.skip("needs the conda interpreter with h5py")
    path = tmp_path / "long.nc"
    T, Y, X, ct = 64, 256, 256, 16
    r = subprocess.run([conda, make, "--payloads", str(path), "7", str(T), str(Y), str(X), str(ct)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    exp = np.load(tmp_path / "long.npz")
    # forced, in ONE decode pass into a pool of output regions (the default) and as count + decode passes; then the default policy
    for forced, passes in (("1", None), ("1", "2"), (None, None)):
        if forced:
            monkeypatch.setenv("ATLITE_HIP_INFLATE_SPLIT", forced)
        else:
            monkeypatch.delenv("ATLITE_HIP_INFLATE_SPLIT")
        if passes:
            monkeypatch.setenv("ATLITE_HIP_SPLIT_PASSES", passes)
        else:
            monkeypatch.delenv("ATLITE_HIP_SPLIT_PASSES", raising=False)
        for v in exp.files:
            d0, h0, r0 = ingest_stats(ctx)
            s0 = ingest_segments(ctx)
            got = slab(ctx, f, v, 0, T)
            assert np.array_equal(got, exp[v]), v
            assert np.array_equal(slab(ctx, f, v, 10, 30), exp[v][10:40]), v
            d1, h1, r1 = ingest_stats(ctx)
            n_streams = T // ct + 3
            assert (d1 - d0, h1 - h0, r1 - r0) == (n_streams, 0, 0), v
            segs = ingest_segments(ctx) - s0
            if forced:
                assert segs >= n_streams, (v, segs)
                if v.split("_")[0] in ("few", "words", "planes", "periodic"):
                    assert segs >= 2 * n_streams, (v, segs)  # a 1 MiB chunk of these is several blocks
    # a pool of output regions that runs out: the segments that found no room fail, their streams are decoded again by the host
    # decoders - the same bytes
    monkeypatch.setenv("ATLITE_HIP_INFLATE_SPLIT", "1")
    monkeypatch.delenv("ATLITE_HIP_SPLIT_PASSES", raising=False)
    monkeypatch.setenv("ATLITE_HIP_SPLIT_POOL_PERCENT", "20")
    d0, h0, r0 = ingest_stats(ctx)
    for v in ("words_6", "few_6", "periodic_9"):
        assert np.array_equal(slab(ctx, f, v, 0, T), exp[v]), v
    d1, h1, r1 = ingest_stats(ctx)
    assert r1 > r0 and (d1 - d0) + (r1 - r0) == 3 * (T // ct)
    monkeypatch.delenv("ATLITE_HIP_SPLIT_POOL_PERCENT")
    # the emulation on the host (atl_inflate_probe(which = 4)) agrees about a stream written the same way
    import zlib

    raw = exp["words_1"][:ct].astype(np.uint8).tobytes()
    comp = np.frombuffer(zlib.compress(raw, 1), np.uint8)
    out = np.zeros(len(raw), np.uint8)
    import ctypes as C

    nseg = C.c_int64()
    check(ctx.lib.atl_inflate_probe(comp.ctypes.data, comp.size, out.ctypes.data, out.size, 4, C.byref(nseg)))
    assert out.tobytes() == raw and nseg.value >= 3


def test_device_inflate_is_the_path_that_ran(ctx, inflate_mode):
    """The counters of atl_nc_ingest_stats: in device mode the deflated chunks of a read are inflated by k_inflate (and none
    had to be decoded again on the host), in host mode none is."""
    f = io.NcFile(f"{NC}/cutout_nc4.nc")
    exp = np.load(f"{NC}/cutout_nc4.npz")
    d0, h0, r0 = ingest_stats(ctx)
    got = slab(ctx, f, "temperature", 0, exp["temperature"].shape[0])
    assert np.array_equal(got, exp["temperature"], equal_nan=True)
    d1, h1, r1 = ingest_stats(ctx)
    if inflate_mode == "device" and f.variables["temperature"].deflate is not None:
        assert d1 > d0 and h1 == h0 and r1 == r0
    else:
        assert d1 == d0 and h1 > h0


def test_device_inflate_many_chunks(ctx, tmp_path, monkeypatch):
    """A cutout of a few thousand chunk streams (written by h5py when the conda interpreter is there): the device decoder's
    output == the host decoders' == h5py's, no stream declined, through FileArray.to_device (one big read per variable)
    and through ragged row ranges."""
    import subprocess

    conda = "/opt/conda/bin/python3.9"
    make = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")
    try:
        ok = subprocess.run([conda, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        ok = False
    if not ok:
        pytest.skip("needs the conda interpreter with h5py")
    path = tmp_path / "many.nc"
    T, Y, X = 120, 96, 80
    r = subprocess.run([conda, make, "--cutout", str(path), str(T), str(Y), str(X), "8", "24", "20", "f4", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    names = [n for n, v in f.variables.items() if v.ndim == 3]
    assert len(names) >= 7 and f.variables[names[0]].n_chunks == 15 * 4 * 4
    host = {}
    monkeypatch.setenv("ATLITE_HIP_INFLATE", "host")
    for n in names:
        host[n] = slab(ctx, f, n, 0, T)
        assert np.array_equal(host[n], f.read(n), equal_nan=True)
    monkeypatch.setenv("ATLITE_HIP_INFLATE", "device")
    d0, _, r0 = ingest_stats(ctx)
    for n in names:
        assert np.array_equal(slab(ctx, f, n, 0, T), host[n], equal_nan=True), n
        assert np.array_equal(slab(ctx, f, n, 13, 77), host[n][13:90], equal_nan=True), n
        dev = io.FileArray(f, n).to_device(ctx)
        ctx.sync()
        assert np.array_equal(dev.numpy().reshape(T, Y, X), host[n], equal_nan=True), n
    # ... and all of them in one read (atl_nc_read_slabs: one launch over every variable's streams)
    outs = [ctx.zeros((T, Y, X)) for _ in names]
    ctx.copy_after_compute()
    f.read_slabs(ctx, names, 5, 100, [o.ptr for o in outs])
    ctx.copy_barrier()
    ctx.sync()
    for n, o in zip(names, outs):
        assert np.array_equal(o.numpy()[:100], host[n][5:105], equal_nan=True), n
    d1, _, r1 = ingest_stats(ctx)
    assert d1 - d0 >= len(names) * 240 * 2 and r1 == r0
    # the default policy: a read of many chunks (default >= 1024) goes to the device, a small one stays on the host threads
    monkeypatch.delenv("ATLITE_HIP_INFLATE")
    monkeypatch.setenv("ATLITE_HIP_INFLATE_MIN_CHUNKS", "200")
    d1, h1, _ = ingest_stats(ctx)
    assert np.array_equal(slab(ctx, f, names[0], 0, T), host[names[0]], equal_nan=True)  # 240 chunks
    assert np.array_equal(slab(ctx, f, names[0], 0, 8), host[names[0]][:8], equal_nan=True)  # 16 chunks
    d2, h2, _ = ingest_stats(ctx)
    assert d2 - d1 == 240 and h2 - h1 == 16


def test_fed_launch_knobs_and_fallbacks_give_the_same_bytes(ctx, tmp_path, monkeypatch, inflate_mode):
    """The fed k_inflate launch (round 6: the kernel starts before its data, DMA batches and CPU-set arrival flags feed it) under
    every way it can be made to run: small DMA batches (many flags), several jobs per read, the unfed order, and a time-out so
    short that waves give up before their batch lands - those streams come back as "not run" and the host decoders take them
    (ingest_stats counts them) - always the bytes h5py wrote."""
    import subprocess

    if inflate_mode != "device":
        pytest.skip("device path only")
    conda = "/opt/conda/bin/python3.9"
    make = os.path.join(os.path.dirname(__file__), "golden", "make_nc_fixtures.py")
    try:
        ok = subprocess.run([conda, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        ok = False
    if not ok:
        pytest.skip("needs the conda interpreter with h5py")
    path = tmp_path / "fed.nc"
    T, Y, X = 96, 96, 80
    r = subprocess.run([conda, make, "--cutout", str(path), str(T), str(Y), str(X), "8", "24", "20", "f4", "9", "4", "pv"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    f = io.NcFile(path)
    names = [n for n, v in f.variables.items() if v.ndim == 3]
    assert len(names) == 7
    monkeypatch.setenv("ATLITE_HIP_INFLATE", "host")
    want = {n: slab(ctx, f, n, 0, T) for n in names}
    monkeypatch.setenv("ATLITE_HIP_INFLATE", "device")

    def all_at_once(t0=3, n=90):
        outs = [ctx.zeros((T, Y, X)) for _ in names]
        ctx.copy_after_compute()
        f.read_slabs(ctx, names, t0, n, [o.ptr for o in outs])
        ctx.copy_barrier()
        ctx.sync()
        for nm, o in zip(names, outs):
            assert np.array_equal(o.numpy()[:n], want[nm][t0:t0 + n], equal_nan=True), nm

    for env in ({"ATLITE_HIP_INGEST_BATCH": "1"},                                   # 1 MiB batches: a dozen flags per read
                {"ATLITE_HIP_INGEST_JOB_GB": "0.004"},                              # ~4 MB of inflated chunks per job: several jobs
                {"ATLITE_HIP_INGEST_FED": "0"},                                     # every byte first, then the launch
                {"ATLITE_HIP_INGEST_BATCH": "1", "ATLITE_HIP_INGEST_JOB_GB": "0.004"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d0, _, r0 = ingest_stats(ctx)
        all_at_once()
        d1, _, r1 = ingest_stats(ctx)
        assert d1 > d0 and r1 == r0, env
        for k in env:
            monkeypatch.delenv(k)
    # waves that give up: a time-out of one microsecond - a wave that finds its batch's flag unset leaves at once
    monkeypatch.setenv("ATLITE_HIP_INGEST_TIMEOUT_MS", "0.001")
    monkeypatch.setenv("ATLITE_HIP_INGEST_BATCH", "1")
    d0, _, r0 = ingest_stats(ctx)
    for _ in range(3):
        all_at_once()
    d1, _, r1 = ingest_stats(ctx)
    assert (d1 - d0) + (r1 - r0) >= 3 * 7 * 12 * 16  # every stream was settled one way or the other ...
    assert r1 > r0                                    # ... and some by the host decoders: the fallback ran


def test_a_failed_read_keeps_its_verdict_until_somebody_listens(ctx, tmp_path, monkeypatch, inflate_mode):
    """A device-inflate read of a corrupt chunk whose verdict nobody collects at first - the file is closed (which settles the
    read and must not swallow the failure), blocks are allocated and recycled (their fences do not consume it either) - reports
    the host decoders' error to the next caller that observes the copy stream, once; the context works on afterwards."""
    if inflate_mode != "device":
        pytest.skip("device path only")
    src = f"{NC}/cutout_nc4.nc"
    exp = np.load(f"{NC}/cutout_nc4.npz")
    good = io.NcFile(src)
    var = good.variables["temperature"]
    if var.deflate is None:
        pytest.skip("fixture variable is not deflated")
    # the last bytes of the file belong to some chunk's zlib stream only by luck; find one of 'temperature' by trial: flip bytes
    # until the HOST path fails on this variable (the host decoders' verdict is what must come back)
    raw = bytearray(open(src, "rb").read())
    rng = np.random.default_rng(3)
    path = None
    for attempt in range(200):
        b = bytearray(raw)
        for pos in rng.integers(len(b) // 3, len(b), size=3):
            b[int(pos)] ^= 0xFF
        cand = tmp_path / f"bad{attempt}.nc"
        open(cand, "wb").write(bytes(b))
        monkeypatch.setenv("ATLITE_HIP_INFLATE", "host")
        try:
            f = io.NcFile(cand)
            slab(ctx, f, "temperature", 0, exp["temperature"].shape[0])
            f.close()
        except ValueError:
            path = cand
            break
        except Exception:
            continue
    assert path is not None, "no corruption of the fixture made the host path fail"
    monkeypatch.setenv("ATLITE_HIP_INFLATE", "device")
    f = io.NcFile(path)
    T = exp["temperature"].shape[0]
    out = ctx.zeros((T,) + var.shape[1:])
    ctx.copy_after_compute()
    f.read_slab(ctx, "temperature", 0, T, out.ptr)  # enqueued; nobody has looked at the copy stream yet
    f.close()                                       # settles the read: the failure must survive this
    junk = [ctx.empty((1000,)) for _ in range(4)]    # pooled blocks: allocated, released, recycled behind fence events
    del junk
    again = ctx.empty((1000,))
    with pytest.raises(ValueError):
        ctx.copy_barrier()                          # the first observer gets the host decoders' error ...
    ctx.copy_barrier()                              # ... once
    ctx.sync()
    g = io.NcFile(src)
    assert np.array_equal(slab(ctx, g, "temperature", 0, T), exp["temperature"], equal_nan=True)
    del again


def test_corrupt_streams_get_the_host_decoders_verdict(ctx, tmp_path, monkeypatch):
    """Bytes of a cutout file overwritten at random: whatever the host path says about a variable - an error, or data -
    the device path says as well (streams the device decoder declines are decoded again by the host decoders before the
    copy stream can be observed)."""
    import shutil

    src = f"{NC}/cutout_nc4.nc"
    raw = bytearray(open(src, "rb").read())
    rng = np.random.default_rng(11)
    names = ["temperature", "influx_direct", "runoff", "albedo"]
    n_err = n_diff = 0
    for case in range(12):
        b = bytearray(raw)
        for pos in rng.integers(len(b) // 3, len(b), size=int(rng.integers(1, 6))):
            b[int(pos)] ^= 1 << int(rng.integers(8))
        path = tmp_path / f"c{case}.nc"
        open(path, "wb").write(bytes(b))
        verdicts = {}
        for mode in ("host", "device"):
            monkeypatch.setenv("ATLITE_HIP_INFLATE", mode)
            try:
                f = io.NcFile(path)
            except Exception as e:  # the container itself no longer parses: nothing to compare
                verdicts[mode] = ("open", type(e).__name__)
                continue
            out = {}
            for n in names:
                try:
                    out[n] = slab(ctx, f, n, 0, f.variables[n].shape[0])
                except Exception as e:
                    out[n] = type(e).__name__
            f.close()
            verdicts[mode] = out
        h, d = verdicts["host"], verdicts["device"]
        if isinstance(h, tuple) or isinstance(d, tuple):
            assert h == d
            continue
        for n in names:
            if isinstance(h[n], str) or isinstance(d[n], str):
                assert h[n] == d[n], (case, n, h[n] if isinstance(h[n], str) else "data", d[n] if isinstance(d[n], str) else "data")
                n_err += 1
            else:
                assert np.array_equal(h[n], d[n], equal_nan=True), (case, n)
                n_diff += 1
    assert n_diff > 0
