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


def close(a, b, s=1e-12):
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=s * np.nanmax(np.abs(b)), equal_nan=True)


def slab(ctx, f, name, t0, n):
    var = f.variables[name]
    out = ctx.zeros((max(n, 1),) + var.shape[1:])
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
    """Back-to-back calls alternate between the two staging slots without waiting for the GPU."""
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
