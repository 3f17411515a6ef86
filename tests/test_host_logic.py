"""CPU: host-side logic that needs no GPU - C ABI surface, argument validation of the gateway
(raised before any device work), resources, orientation factories, indicator matrix, labelled
containers."""
import os
import re
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from atlite_amd import Cutout, Dataset, LabeledArray, _lib, gis, resource
from atlite_amd.convert import convert_and_aggregate
from atlite_amd.pv.orientation import get_orientation

ROOT = Path(__file__).resolve().parent.parent
G = Path(__file__).parent / "golden"


def test_cabi_exports_every_declared_symbol():
    """The library loads and exports every symbol include/atlite_hip.h declares."""
    hdr = (ROOT / "include" / "atlite_hip.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(atl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.atl_version() == 102


def test_no_gpu_fails_loudly():
    import ctypes as C

    lib = _lib.load()
    n = C.c_int()
    assert lib.atl_device_count(C.byref(n)) == 0
    if n.value == 0:
        from atlite_amd.device import Context

        with pytest.raises(_lib.AtliteHipError, match="no CPU fallback"):
            Context(0)


def test_product_does_not_import_oracle():
    for f in (ROOT / "atlite_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


@pytest.fixture
def cutout():
    np.random.seed(42)
    times = pd.date_range("2020-01-01", periods=24, freq="h")
    return Cutout(Dataset({"var": np.random.rand(24, 3, 4)}, dict(time=times, y=[50.0, 51.0, 52.0], x=[5.0, 6.0, 7.0, 8.0])))


def identity_convert(ds, **kwargs):
    return ds["var"]


class TestInvalidArgs:  # reference: test/test_aggregate_time.py:137-169 (raised before any compute)
    @pytest.mark.parametrize("bad", ["invalid", False, True])
    def test_invalid_aggregate_time_value(self, cutout, bad):
        with pytest.raises(ValueError, match="aggregate_time must be"):
            convert_and_aggregate(cutout, identity_convert, aggregate_time=bad)

    def test_capacity_factor_with_aggregate_time_raises(self, cutout):
        with pytest.raises(ValueError, match="Cannot use"):
            convert_and_aggregate(cutout, identity_convert, capacity_factor=True, aggregate_time="mean")


def test_grid_order_and_mock_equivalence(cutout):
    g = cutout.grid
    exp = np.array([(x, y) for y in cutout.data.coords["y"] for x in cutout.data.coords["x"]])
    np.testing.assert_array_equal(g[["x", "y"]].values, exp)  # test_aggregate_time.py:16


def test_turbine_config_and_padding():
    g = dict(np.load(G / "wind.npz"))
    for name in ("Vestas_V112_3MW", "Enercon_E101_3000kW", "NREL_ReferenceTurbine_5MW_offshore"):
        t = resource.get_windturbineconfig(name, add_cutout_windspeed=False)
        np.testing.assert_array_equal(np.asarray(t["V"], float), g[f"{name}_V"])
        np.testing.assert_array_equal(np.asarray(t["POW"], float), g[f"{name}_POW"])
        np.testing.assert_array_equal([t["P"], t["hub_height"]], g[f"{name}_P_hub"])
    p = resource.get_windturbineconfig(dict(V=[0, 10, 20], POW=[0, 1.0, 1.0], P=1.0, hub_height=90.0),
                                       add_cutout_windspeed=True)
    np.testing.assert_array_equal(p["V"], g["padded_V"])  # test/test_resource.py:28-34
    np.testing.assert_array_equal(p["POW"], g["padded_POW"])
    assert p["POW"][-1] == 0.0
    q = resource.get_windturbineconfig(dict(V=[0, 10, 20], POW=[0, 1.0, 1.0], P=1.0, hub_height=90.0),
                                       add_cutout_windspeed=False)
    assert q["POW"][-1] == 1.0
    with pytest.raises(ValueError, match="ascending"):
        resource.get_windturbineconfig(dict(V=[0, 10, 5], POW=[0, 1, 1], P=1.0, hub_height=90.0))
    with pytest.raises(ValueError, match="equal length"):
        resource.get_windturbineconfig(dict(V=[0, 10], POW=[0, 1, 1], P=1.0, hub_height=90.0))
    with pytest.raises(KeyError):
        resource.get_windturbineconfig(3.0)
    sm = resource.windturbine_smooth(resource.get_windturbineconfig("Vestas_V112_3MW", add_cutout_windspeed=False),
                                     params=True)
    np.testing.assert_allclose(sm["POW"], g["smooth_POW"], rtol=1e-13, atol=1e-16)
    assert sm["P"] == g["smooth_P"][0]
    # 27 name-addressable turbines of the reference + its three eno_126_* files (no .yaml suffix there)
    assert len(resource.windturbines()) == 30 and resource.solarpanels() == ["CSi", "CdTe", "KANENA"]
    assert resource.get_solarpanelconfig("CSi")["k_1"] == -0.017162


def test_orientation_factories():
    from oracle import atlite_oracle as orc

    lat = LabeledArray(np.radians([-60.0, -30.0, 0.0, 10.0, 25.0, 40.0, 50.0, 72.0]), ("y",), {"y": np.arange(8.0)})
    o = get_orientation("latitude_optimal")(None, lat, None)
    ref = orc.orientation_latitude_optimal(lat.values)
    np.testing.assert_array_equal(o["slope"].values, ref["slope"])
    np.testing.assert_array_equal(o["azimuth"].values, ref["azimuth"])
    c = get_orientation({"slope": 30.0, "azimuth": 180.0})(None, lat, None)
    assert c["slope"] == np.radians(30.0) and c["azimuth"] == np.radians(180.0)
    la = get_orientation({"name": "latitude", "azimuth": 170.0})(None, lat, None)
    assert la["slope"] is lat and la["azimuth"] == np.radians(170.0)


def test_indicator_matrix():
    x, y = np.arange(10.0), 10.0 + 2.0 * np.arange(5.0)
    cell = np.array([[1.5, 11], [2.5, 11], [2.5, 13], [1.5, 13]])  # exactly cell (iy=1, ix=2)
    M = gis.compute_indicatormatrix(x, y, [cell])
    assert M.nnz == 1 and M[0, 1 * 10 + 2] == 1.0  # test/test_gis.py:322-332
    # orientation of the ring does not matter; holes subtract; multi-part shapes add
    rect = np.array([[0.25, 9.5], [3.5, 9.5], [3.5, 12.0], [0.25, 12.0]])
    A = gis.compute_indicatormatrix(x, y, [rect, rect[::-1]]).toarray()
    np.testing.assert_array_equal(A[0], A[1])
    assert abs(A[0].sum() * 2.0 - 3.25 * 2.5) < 1e-12
    hole = np.array([[1.0, 10.0], [2.0, 10.0], [2.0, 11.0], [1.0, 11.0]])
    B = gis.compute_indicatormatrix(x, y, [dict(exterior=rect, holes=[hole])]).toarray()
    assert abs(B.sum() * 2.0 - (3.25 * 2.5 - 1.0)) < 1e-12
    far = rect + np.array([5.0, 0.0])
    C2 = gis.compute_indicatormatrix(x, y, [[rect, far]]).toarray()
    assert abs(C2.sum() * 2.0 - 2 * 3.25 * 2.5) < 1e-12
    # star polygons strictly inside the grid conserve their area
    X = Y = 60
    xx, yy = -25 + (70 / X) * np.arange(X), 30 + (42 / Y) * np.arange(Y)
    polys = gis.random_star_polygons(30, (-15, 35, 35, 66), seed=3)
    Mx = gis.compute_indicatormatrix(xx, yy, polys)
    ca = (70 / X) * (42 / Y)
    for i, p in enumerate(polys):
        a = 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))
        assert abs(Mx[i].sum() * ca - a) < 1e-10 * a
    # a tessellation covers every cell exactly once
    tess = gis.random_tessellation(25, (xx[0] - 35 / X, yy[0] - 21 / Y, xx[-1] + 35 / X, yy[-1] + 21 / Y), seed=1)
    cs = np.asarray(gis.compute_indicatormatrix(xx, yy, tess).sum(0)).ravel()
    np.testing.assert_allclose(cs, 1.0, rtol=0, atol=1e-11)
    s = gis.spdiag(np.array([1.0, 2.0, 3.0]))
    np.testing.assert_array_equal(s.toarray(), np.diag([1.0, 2.0, 3.0]))


def test_labeled_containers():
    t = pd.date_range("2020-01-01", periods=4, freq="h")
    ds = Dataset({"a": np.arange(24.0).reshape(4, 2, 3), "h": np.ones((2, 3))}, dict(time=t, y=[1.0, 2.0], x=[0.0, 1.0, 2.0]))
    assert ds["a"].dims == ("time", "y", "x") and ds["h"].dims == ("y", "x") and "a" in ds and "zz" not in ds
    assert ds["lon"].dims == ("x",) and list(ds) == ["a", "h"]
    a = ds["a"]
    assert a.mean("time").dims == ("y", "x") and a.transpose("x", "y", "time").shape == (3, 2, 4)
    np.testing.assert_array_equal(a.sum("time").values, a.values.sum(0))
    with pytest.raises(KeyError):
        ds["missing"]
    with pytest.raises(ValueError):
        Dataset({"bad": np.zeros((5, 5))}, dict(time=t, y=[1.0, 2.0], x=[0.0, 1.0, 2.0]))


def test_ctypes_structs_match_the_c_header(tmp_path):
    """sizeof / field offsets of every struct in include/atlite_hip.h (plain C, gcc) equal the ctypes
    mirrors in atlite_amd/_lib.py and the stub printed in INTEGRATION.md."""
    import ctypes as C
    import subprocess

    structs = {"atl_pv_inputs": _lib.PvInputs, "atl_pv_params": _lib.PvParams, "atl_wind_inputs": _lib.WindInputs,
               "atl_wind_params": _lib.WindParams, "atl_heat_params": _lib.HeatParams,
               "atl_thermo_params": _lib.ThermoParams, "atl_synth_solar": _lib.SynthSolar}
    last = {k: v._fields_[-1][0] for k, v in structs.items()}
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"atlite_hip.h\"\nint main(void){\n"
    for k in structs:
        prog += f'printf("{k} %zu %zu\\n", sizeof({k}), offsetof({k}, {last[k]}));\n'
    prog += "return 0;}\n"
    src = tmp_path / "sizes.c"
    src.write_text(prog)
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line in filter(None, out):
        name, size, off = line.split()
        st = structs[name]
        assert C.sizeof(st) == int(size), (name, C.sizeof(st), size)
        assert getattr(st, last[name]).offset == int(off), (name, last[name])
    doc = (ROOT / "INTEGRATION.md").read_text()
    ns = {"C": C}
    exec(doc[doc.index("class PvInputs(C.Structure)"):doc.index("class Ctx:")], ns)
    assert C.sizeof(ns["PvInputs"]) == C.sizeof(_lib.PvInputs) and C.sizeof(ns["PvParams"]) == C.sizeof(_lib.PvParams)
    for name in ("PvInputs", "PvParams"):  # sizeof alone hides a trailing int inside the padding
        mine, doc_st = getattr(_lib, name), ns[name]
        assert [f[0] for f in doc_st._fields_] == [f[0] for f in mine._fields_], name
        assert all(getattr(doc_st, f[0]).offset == getattr(mine, f[0]).offset for f in mine._fields_), name


def test_tile_geometry_selfcheck():
    """Cell-tile geometry on the HOST (atl_agg_selfcheck): the lane -> cell mapping the kernels use and
    the plan builder's inverse agree, every cell is owned by exactly one lane and no lane points outside
    the cube - for every row-length residue modulo 16 (straddling 128-byte lines), odd row lengths,
    grids smaller than a tile, single rows and the benchmark grids."""
    import ctypes as C

    from atlite_amd import _lib

    lib = _lib.load()
    grids = [(5, x) for x in range(1, 70)] + [(1, 37), (1, 1), (2, 7), (9, 3), (17, 15), (64, 16), (130, 31),
                                              (200, 200), (400, 400), (33, 250), (7, 1000)]
    for Y, X in grids:
        for tw in (16, 32, 64, 128):
            nt, no, ne = C.c_int64(), C.c_int64(), C.c_int64()
            _lib.check(lib.atl_agg_selfcheck(Y * X, X, tw, C.byref(nt), C.byref(no), C.byref(ne)))
            assert ne.value == 0 and no.value == Y * X, (Y, X, tw, nt.value, no.value, ne.value)
    # flat layout (unknown grid): one row of n cells in 128-cell tiles
    for n in (1, 127, 128, 129, 40000):
        nt, no, ne = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.atl_agg_selfcheck(n, 0, 128, C.byref(nt), C.byref(no), C.byref(ne)))
        assert ne.value == 0 and no.value == n and nt.value == (n + 127) // 128
    with pytest.raises(ValueError):
        _lib.check(lib.atl_agg_selfcheck(10, 3, 16, C.byref(nt), C.byref(no), C.byref(ne)))


def test_aligned_tile_geometry_selfcheck():
    """... and the tilings of a line-aligned plan (atl_agg_selfcheck_aligned): for contiguous cubes with S % 16 != 0 every
    alignment class (16 / gcd(S, 16) of them) tiles the grid with its rows on the line grid of ITS slots - every class owns
    every cell once, lanes load 16-byte aligned pairs, tile rows start 128-byte lines, the plan builder's inverse agrees."""
    import ctypes as C
    import math

    from atlite_amd import _lib

    lib = _lib.load()
    grids = [(5, x) for x in range(4, 70)] + [(1, 37), (9, 3), (17, 15), (130, 31), (201, 201), (189, 157), (201, 200), (33, 250), (3, 1001)]
    for Y, X in grids:
        S = Y * X
        if S % 16 == 0 or S < 16:
            continue
        for tw in (16, 32, 64, 128):
            nc, nt, no, ne = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            _lib.check(lib.atl_agg_selfcheck_aligned(S, X, tw, C.byref(nc), C.byref(nt), C.byref(no), C.byref(ne)))
            assert nc.value == 16 // math.gcd(S, 16)
            assert ne.value == 0 and no.value == nc.value * S, (Y, X, tw, nc.value, nt.value, no.value, ne.value)
    for n in (17, 127, 129, 40401):  # flat strips
        nc, nt, no, ne = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.atl_agg_selfcheck_aligned(n, 0, 128, C.byref(nc), C.byref(nt), C.byref(no), C.byref(ne)))
        assert ne.value == 0 and no.value == nc.value * n and nt.value == nc.value * ((n + 15 + 127) // 128)
    with pytest.raises(ValueError):
        _lib.check(lib.atl_agg_selfcheck_aligned(64, 0, 128, C.byref(nc), C.byref(nt), C.byref(no), C.byref(ne)))


def test_gateway_asks_for_line_aligned_plans_only_where_they_pay(monkeypatch):
    """convert._aligned_plan_wanted: the caller's own contiguous device cubes, a cell count off the 16-cell line grid, and room for
    16 / gcd(S, 16) stacked copies of the matrix in a plan's 65535 rows; heat / cooling demand never ask (day groups)."""
    from atlite_amd import convert

    class DS:
        def __init__(self, caller):
            self.caller = caller

        def _caller_layout(self):
            return self.caller

    M = sp.csr_matrix(np.ones((100, 201 * 201)))
    assert convert._aligned_plan_wanted(DS(True), M, 201 * 201)
    assert not convert._aligned_plan_wanted(DS(False), M, 201 * 201)       # the library's own padded copies
    assert not convert._aligned_plan_wanted(DS(True), sp.csr_matrix(np.ones((3, 40000))), 40000)  # slots on the line grid already
    assert not convert._aligned_plan_wanted(object(), M, 201 * 201)         # a duck-typed dataset without the hook
    big = sp.csr_matrix((5000, 201 * 201))
    assert not convert._aligned_plan_wanted(DS(True), big, 201 * 201)       # 16 x 5000 rows do not fit
    assert convert._aligned_plan_wanted(DS(True), sp.csr_matrix((5000, 40200)), 40200)  # S % 16 = 8: two classes
    monkeypatch.setenv("ATLITE_HIP_ALIGNED_PLANS", "0")
    assert not convert._aligned_plan_wanted(DS(True), M, 201 * 201)
    assert convert._Spec.aligned_ok and not convert._HeatSpec.aligned_ok and not convert._CoolSpec.aligned_ok


def test_streaming_sources_and_policy(monkeypatch):
    """Host-side policy of the slab pipeline (no GPU): which datasets stream, and how sources are
    normalised (fp64 as is, narrower native dtypes kept for the device decode, exotic ones widened)."""
    from atlite_amd import Dataset, io, streaming

    T, Y, X = 6, 2, 3
    t = pd.date_range("2013-01-01", periods=T, freq="h")
    coords = dict(time=t, y=[0.0, 1.0], x=[0.0, 1.0, 2.0])

    class Spec:
        time_vars = ("runoff",)

    a64 = np.arange(T * Y * X, dtype=np.float64).reshape(T, Y, X)
    ds = Dataset({"runoff": a64}, coords)
    monkeypatch.setenv("ATLITE_HIP_STREAM", "auto")
    assert not streaming.wanted(ds, Spec())  # small host data: whole-variable upload
    monkeypatch.setenv("ATLITE_HIP_STREAM_MIN_BYTES", "1")
    assert streaming.wanted(ds, Spec())
    monkeypatch.setenv("ATLITE_HIP_STREAM", "0")
    assert not streaming.wanted(ds, Spec())
    monkeypatch.setenv("ATLITE_HIP_STREAM", "auto")
    monkeypatch.delenv("ATLITE_HIP_STREAM_MIN_BYTES")
    fds = io.open_cutout(os.path.join(os.path.dirname(__file__), "golden", "nc", "cutout_small_f32.nc"))
    assert streaming.wanted(fds, Spec())  # file-backed variables always stream

    class NoTime:
        time_vars = ()

    assert not streaming.wanted(fds, NoTime())
    # source normalisation
    s = streaming._source(a64, T, Y * X)
    assert s.dtype == np.float64 and s.shape == (T, Y * X) and np.shares_memory(s, a64)
    for dt in ("float32", "int16", "uint8", "int64"):
        s = streaming._source(a64.astype(dt), T, Y * X)
        assert s.dtype == np.dtype(dt) and s.flags.c_contiguous
    assert streaming._source(a64.astype(">f4"), T, Y * X).dtype == np.float64  # big-endian: widened on the host
    assert streaming._source(a64.astype(np.float16), T, Y * X).dtype == np.float64
    assert streaming._source(a64.astype(bool), T, Y * X).dtype == np.float64
    nc = np.asfortranarray(a64.astype(np.float32))
    s = streaming._source(nc, T, Y * X)
    assert s.flags.c_contiguous and np.array_equal(s.reshape(T, Y, X), nc)
    fa = fds["temperature"].data
    assert streaming._source(fa, 48, 108) is fa


def test_time_partition_properties():
    from atlite_amd.distributed import time_partition

    rng = np.random.default_rng(0)
    for _ in range(300):
        n, w = int(rng.integers(0, 5000)), int(rng.integers(1, 17))
        align = int(rng.choice([1, 24]))
        first = int(rng.integers(0, 24)) if align > 1 else 0
        e = time_partition(n, w, align=align, first=first)
        assert len(e) == w + 1 and e[0] == 0 and e[-1] == n and all(b >= a for a, b in zip(e, e[1:]))
        if align > 1:
            assert all((v - first) % align == 0 or v in (0, n) for v in e[1:-1])
        if n >= w * 2 * align:
            sizes = np.diff(e)
            assert sizes.max() - sizes.min() <= 2 * align


def test_product_never_calls_the_host_probes():
    """The host builds of the kernel math (atl_*_probe_host, atl_wind_interp_host, atl_agg_selfcheck,
    atl_inflate_probe) are test entry points: the package only declares their signatures, no product
    module calls them - there is no CPU path to fall back on."""
    pkg = Path(__file__).resolve().parent.parent / "atlite_amd"
    names = ("atl_math_probe_host", "atl_pv_probe_host", "atl_wind_probe_host", "atl_wind_interp_host", "atl_agg_selfcheck",
             "atl_inflate_probe")
    for py in pkg.rglob("*.py"):
        text = py.read_text()
        for n in names:
            hits = [l for l in text.splitlines() if n in l]
            if py.name == "_lib.py":
                assert all(l.strip().startswith(f'"{n}') for l in hits), (py, n, hits)  # the signature table only (atl_agg_selfcheck[_aligned])
            else:
                assert not hits, (py, n)


def test_multigpu_device_selection_and_day_aligned_shards(monkeypatch):
    """Host side of atlite_amd.multigpu: device-list resolution order and shard edges (no GPU needed)."""
    import pandas as pd

    from atlite_amd import multigpu
    from atlite_amd.convert import _HeatSpec, _RunoffSpec
    from atlite_amd.labeled import Dataset

    class C:
        devices = None

    monkeypatch.delenv("ATLITE_HIP_DEVICES", raising=False)
    assert multigpu.devices_for(C()) is None
    monkeypatch.setenv("ATLITE_HIP_DEVICES", "0, 1,2")
    assert multigpu.devices_for(C()) == [0, 1, 2]
    multigpu.set_devices([3, 4])
    assert multigpu.devices_for(C()) == [3, 4]
    c = C()
    c.devices = [5]
    assert multigpu.devices_for(c) == [5]
    multigpu.set_devices(None)

    T = 24 * 10 + 7
    time = pd.date_range("2013-01-01 05:00", periods=T, freq="h")
    ds = Dataset({"temperature": np.zeros((T, 2, 3)), "runoff": np.zeros((T, 2, 3)), "height": np.zeros((2, 3))},
                 dict(time=time, y=[0.0, 1.0], x=[0.0, 1.0, 2.0]))
    heat = _HeatSpec(ds, 15.0, 1.0, 0.0, 3.0)
    for n in (1, 2, 3, 8, 16):
        e = heat.shard_edges(T, n)
        assert e[0] == 0 and e[-1] == T and len(e) == n + 1 and all(b >= a for a, b in zip(e, e[1:]))
        assert set(e) <= set(int(v) for v in heat.day_ptr)  # shard boundaries are day boundaries of the shifted axis
        slots = [heat.out_slots(a, b) for a, b in zip(e, e[1:])]
        assert slots[0][0] == 0 and slots[-1][1] == len(heat.days)
        assert all(s[1] == t[0] for s, t in zip(slots, slots[1:]))
    e = _RunoffSpec(ds).shard_edges(T, 4)
    assert e[0] == 0 and e[-1] == T and max(b - a for a, b in zip(e, e[1:])) - min(b - a for a, b in zip(e, e[1:])) <= 1


def test_xarray_bridge_with_the_stand_in(monkeypatch):
    """LabeledArray.to_xarray / Dataset.from_xarray / _finish against a DataArray / Dataset double (xarray is not
    installable here): dims, coords, attrs, name and values survive both directions, dimension order is normalised
    to (time, y, x) on the way in."""
    import pandas as pd

    from atlite_amd import convert, labeled
    from atlite_amd.labeled import Dataset, LabeledArray
    from tests import helpers as H

    xr = H.xarray_stand_in()
    monkeypatch.setattr(labeled, "xr", xr)
    t = pd.date_range("2013-01-01", periods=4, freq="h")
    la = LabeledArray(np.arange(8.0).reshape(2, 4), ("bus", "time"), {"bus": np.array(["a", "b"]), "time": t},
                      {"units": "MWh/MWp"}, "specific generation")
    da = la.to_xarray()
    assert isinstance(da, xr.DataArray) and da.dims == ("bus", "time") and da.name == "specific generation"
    assert da.attrs == {"units": "MWh/MWp"}
    np.testing.assert_array_equal(da.values, la.values)
    np.testing.assert_array_equal(da.coords["bus"].values, ["a", "b"])
    np.testing.assert_array_equal(pd.DatetimeIndex(da.coords["time"].values), t)
    assert isinstance(convert._finish(la), xr.DataArray)

    y, x = np.array([50.0, 50.25, 50.5]), np.array([8.0, 8.25])
    cube = np.arange(24.0).reshape(4, 3, 2)
    ds = xr.Dataset(
        {"temperature": xr.DataArray(cube.transpose(2, 0, 1), dims=["x", "time", "y"], coords={"x": x, "time": t, "y": y}),
         "height": xr.DataArray(np.ones((3, 2)), dims=["y", "x"], coords={"y": y, "x": x})},
        coords={"time": t, "y": y, "x": x}, attrs={"module": "era5"})
    d = Dataset.from_xarray(ds)
    assert d["temperature"].dims == ("time", "y", "x") and d["height"].dims == ("y", "x")
    np.testing.assert_array_equal(d["temperature"].values, cube)
    np.testing.assert_array_equal(d.coords["y"], y)
    assert d.attrs == {"module": "era5"} and not d.chunked
    assert isinstance(convert._as_dataset(ds), Dataset)


def _check_plan(M, row_len, env=None, aligned=False):
    import ctypes as C

    from atlite_amd import _lib

    M = sp.csr_matrix(M)
    indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(M.indices, dtype=np.int32)
    data = np.ascontiguousarray(M.data, dtype=np.float64)
    P, dense, err = C.c_int64(), C.c_int64(), C.c_int64()
    fn = _lib.load().atl_agg_check_host_aligned if aligned else _lib.load().atl_agg_check_host
    _lib.check(fn(M.shape[0], M.shape[1], row_len, indptr.ctypes.data, indices.ctypes.data if len(indices) else None,
                  data.ctypes.data if len(data) else None, C.byref(P), C.byref(dense), C.byref(err)))
    return P.value, dense.value, err.value


@pytest.mark.parametrize("tile", [None, "16x8", "64x2", "flat"])
def test_aligned_plan_builder_is_consistent_with_its_matrix(monkeypatch, tile):
    """... and the plan of atl_agg_create_aligned: 16 / gcd(S, 16) tilings of the grid, each verified entry by entry against
    its copy of the matrix (rows of class r only in class r's tiles, every weight on the lane that owns the cell under that
    class's origin, masks, MFMA images) - no device."""
    if tile:
        monkeypatch.setenv("ATLITE_HIP_TILE", tile)
    rng = np.random.default_rng(11)
    done = 0
    while done < 16:
        Y, X = int(rng.integers(1, 30)), int(rng.integers(1, 60))
        if (Y * X) % 16 == 0 or Y * X < 16:
            continue
        done += 1
        N = int(rng.integers(1, 30))
        M = sp.random(N, Y * X, density=float(rng.choice([0.02, 0.2, 1.0])), random_state=int(rng.integers(1 << 30)), format="csr")
        M.data[:] = rng.normal(size=M.nnz)
        if M.nnz and done % 4 == 0:
            M.data[int(rng.integers(M.nnz))] = np.nan
        P, dense, err = _check_plan(M, X if done % 3 else 0, aligned=True)
        assert err == 0, (done, Y, X, N, tile)
    monkeypatch.setenv("ATLITE_HIP_FORCE_MFMA", "1")
    W = rng.normal(size=(20, 11 * 27))
    W[rng.random(W.shape) < 0.3] = 0.0
    P, dense, err = _check_plan(sp.csr_matrix(W), 27, aligned=True)
    assert err == 0 and dense > 0
    with pytest.raises(ValueError, match="not a multiple of 16"):
        _check_plan(sp.csr_matrix(np.ones((2, 64))), 8, aligned=True)


@pytest.mark.parametrize("tile", [None, "16x8", "32x4", "64x2", "flat"])
def test_plan_builder_is_consistent_with_its_matrix(monkeypatch, tile):
    """The aggregation plan (partial rows per tile, their order per shape, coverage masks, the MFMA operand image of
    dense tiles) rebuilt on the host and verified entry by entry against the CSR matrix it was built from - no device:
    random sparse matrices with explicit zeros, negative and NaN weights, empty rows and columns, unsorted columns,
    grids whose rows are not a multiple of a 128-byte line, dense stacks of rows, degenerate shapes."""
    if tile:
        monkeypatch.setenv("ATLITE_HIP_TILE", tile)
    rng = np.random.default_rng(5)
    for case in range(24):
        Y, X = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        N = int(rng.integers(1, 60))
        dens = float(rng.choice([0.01, 0.1, 0.5, 1.0]))
        M = sp.random(N, Y * X, density=dens, random_state=int(rng.integers(1 << 30)), format="csr")
        M.data[:] = rng.normal(size=M.nnz)
        if M.nnz:
            M.data[rng.random(M.nnz) < 0.05] = 0.0          # explicit zeros are structural entries
            if case % 5 == 0:
                M.data[int(rng.integers(M.nnz))] = np.nan    # poisons its row
        row_len = X if case % 4 else 0                        # unknown grid: flat 128-cell tiles
        P, dense, err = _check_plan(M, row_len)
        assert err == 0, (case, Y, X, N, dens, tile)
        assert P <= M.nnz or M.nnz == 0
    # duplicates (COO-style input) are summed; columns need not be sorted
    indptr = np.array([0, 4, 4, 6], dtype=np.int64)
    indices = np.array([5, 2, 5, 0, 7, 7], dtype=np.int32)
    data = np.array([1.0, 2.0, 3.0, 4.0, -1.0, 1.0])
    import ctypes as C

    from atlite_amd import _lib

    P, dense, err = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(_lib.load().atl_agg_check_host(3, 12, 4, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data,
                                              C.byref(P), C.byref(dense), C.byref(err)))
    assert err.value == 0 and P.value >= 2
    # dense stacks: every tile carries 16 / 17 / 40 rows -> MFMA operand images
    monkeypatch.setenv("ATLITE_HIP_FORCE_MFMA", "1")
    for R in (16, 17, 28, 40):
        W = rng.normal(size=(R, 11 * 27))
        W[rng.random(W.shape) < 0.3] = 0.0
        P, dense, err = _check_plan(sp.csr_matrix(W), 27)
        assert err == 0 and dense > 0, (R, dense, err)
    # bad input is refused, not crashed on
    nerr = C.c_int64()
    with pytest.raises(ValueError):
        _lib.check(_lib.load().atl_agg_check_host(1, 4, 0, np.array([0, 1], dtype=np.int64).ctypes.data,
                                                  np.array([9], dtype=np.int32).ctypes.data, np.array([1.0]).ctypes.data,
                                                  None, None, C.byref(nerr)))


def test_indicator_matrix_line_integrals_against_the_clipper():
    """The device indicator-matrix algorithm (edges bucketed by grid column, area(shape n cell) as a line integral of the
    clamped edge heights, compaction) with its candidate cells evaluated on the HOST - the same source as the kernel -
    against the polygon clipper: two independent algorithms for the same contract (atlite/gis.py:104-145)."""
    x, y = np.arange(10.0), 10.0 + 2.0 * np.arange(5.0)
    rect = np.array([[0.25, 9.5], [3.5, 9.5], [3.5, 12.0], [0.25, 12.0]])
    hole = np.array([[1.0, 10.0], [2.0, 10.0], [2.0, 11.0], [1.0, 11.0]])
    shapes = [rect, rect[::-1], dict(exterior=rect, holes=[hole]), [rect, rect + np.array([5.0, 0.0])],
              rect + np.array([8.0, 6.0]), rect + np.array([100.0, 100.0]),
              np.array([[1.5, 11], [2.5, 11], [2.5, 13], [1.5, 13]])]
    A = gis.compute_indicatormatrix(x, y, shapes, ctx="integral-host")
    B = gis.compute_indicatormatrix(x, y, shapes)
    assert (A != 0).toarray().tolist() == (B != 0).toarray().tolist()
    np.testing.assert_allclose(A.toarray(), B.toarray(), rtol=0, atol=1e-13)
    X = Y = 90
    xx, yy = -25 + (70 / X) * np.arange(X), 30 + (42 / Y) * np.arange(Y)
    polys = gis.random_star_polygons(30, (-15, 35, 35, 66), seed=3)
    th = np.linspace(0, 2 * np.pi, 3000, endpoint=False)
    rad = 9.0 + 0.8 * np.sin(17 * th) + 0.3 * np.cos(61 * th)
    polys.append(np.stack([5.0 + 1.6 * rad * np.cos(th), 50.0 + rad * np.sin(th)], axis=1))  # ~100 edges per column
    polys += gis.random_tessellation(20, (xx[0] - 35 / X, yy[0] - 21 / Y, xx[-1] + 35 / X, yy[-1] + 21 / Y), seed=1)
    A = gis.compute_indicatormatrix(xx, yy, polys, ctx="integral-host")
    B = gis.compute_indicatormatrix(xx, yy, polys)
    np.testing.assert_allclose(A.toarray(), B.toarray(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.asarray(A[31:].sum(0)).ravel(), 1.0, rtol=0, atol=1e-11)  # the tessellation covers every cell once



def test_cutout_descriptive_properties_and_layout_helpers():
    """The small Cutout helpers around the hot path (cutout.py:284-345, 537-642): affine transform, dt, area and
    density layout in the cutout's crs, layout_from_capacity_list (nearest cell, the reference's wrap for entries at
    or below the first coordinate included), equals."""
    import pandas as pd

    from atlite_amd import Cutout, Dataset

    x, y = np.linspace(5.0, 8.0, 13), np.linspace(47.0, 49.0, 9)
    t = pd.date_range("2013-01-01", periods=6, freq="h")
    rng = np.random.default_rng(0)
    ds = Dataset({"temperature": rng.random((6, 9, 13))}, dict(time=t, y=y, x=x), attrs={"module": "era5"})
    c = Cutout(ds)
    assert c.dx == 0.25 and c.dy == 0.25 and c.dt in ("h", "H") and c.module == "era5" and c.name is None
    tr = c.transform
    assert tuple(tr) == (0.25, 0.0, 4.875, 0.0, 0.25, 46.875) and tr * (0.5, 0.5) == (5.0, 47.0)
    assert tuple(c.transform_r) == (0.25, 0.0, 4.875, 0.0, -0.25, 49.125)
    assert list(c.prepared_features.index) == [("era5", "temperature")]
    a = c.area()
    assert a.shape == (9, 13) and np.allclose(np.asarray(a.values), 0.0625)
    np.testing.assert_allclose(np.asarray(c.uniform_density_layout(3.0).values), 0.1875)
    with pytest.raises(NotImplementedError):
        c.area(crs=3035)
    plants = pd.DataFrame({"x": [5.0, 5.1, 5.13, 7.99, 9.5, 6.0], "y": [47.0, 47.3, 47.38, 48.9, 50.0, 46.0],
                           "Capacity": [1.0, 2.0, 4.0, 8.0, 16.0, 32.0]})
    lay = np.asarray(c.layout_from_capacity_list(plants).values)
    # a restatement of cutout.py:623-642 with plain loops
    ref = np.zeros((9, 13))
    for px, py, cap in plants.itertuples(index=False):
        ix = min(max(int(np.searchsorted(x, px, side="left")), 0), 12)
        iy = min(max(int(np.searchsorted(y, py, side="left")), 0), 8)
        ix -= int(px - x[ix - 1] < x[ix] - px)
        iy -= int(py - y[iy - 1] < y[iy] - py)
        ref[iy, ix] += cap
    np.testing.assert_array_equal(lay, ref)
    # (5.0, 47.0) sits exactly on the first coordinates and (6.0, 46.0) below the first row: the reference wraps both
    assert lay.sum() == 63.0 and lay[1, 0] == 2.0 and lay[2, 1] == 4.0 and lay[8, 12] == 1.0 + 8.0 + 16.0 and lay[8, 4] == 32.0
    other = Cutout(Dataset({"temperature": ds["temperature"].values.copy()}, dict(time=t, y=y, x=x)))
    assert c.equals(other) and c.equals(c)
    other.data["temperature"].values[0, 0, 0] += 1.0
    assert not c.equals(other) and c.equals(3) is NotImplemented


def test_shapes_as_geojson_and_geodataframe_like():
    """Shapes may come as GeoJSON Polygon / MultiPolygon mappings, Features, objects with __geo_interface__ (what
    shapely / geopandas geometries expose) or a GeoDataFrame-like object with .geometry and .index
    (atlite/gis.py:127, convert.py:236-238) - the same matrix as the plain vertex arrays."""
    import pandas as pd

    from atlite_amd import gis

    x, y = np.linspace(0.0, 9.0, 10), np.linspace(50.0, 55.0, 6)
    outer = np.array([[1.2, 50.6], [6.7, 50.9], [7.4, 54.1], [2.0, 53.8]])
    hole = np.array([[3.0, 51.5], [4.5, 51.6], [4.4, 52.9], [3.1, 52.7]])
    tri = np.array([[7.5, 50.0], [9.4, 50.2], [8.8, 52.0]])
    ref = gis.compute_indicatormatrix(x, y, [dict(exterior=outer, holes=[hole]), [outer, tri], tri]).toarray()
    closed = lambda r: np.vstack([r, r[:1]]).tolist()  # noqa: E731  (GeoJSON rings repeat their first vertex)
    gj = [{"type": "Polygon", "coordinates": [closed(outer), closed(hole)]},
          {"type": "MultiPolygon", "coordinates": [[closed(outer)], [closed(tri)]]},
          {"type": "Feature", "properties": {}, "geometry": {"type": "Polygon", "coordinates": [closed(tri)]}}]
    np.testing.assert_allclose(gis.compute_indicatormatrix(x, y, gj).toarray(), ref, rtol=0, atol=1e-14)

    class Geom:
        def __init__(self, m):
            self.__geo_interface__ = m

    class Frame:  # the two attributes the gateway reads off a GeoDataFrame
        def __init__(self, geoms, index):
            self.geometry = pd.Series(geoms, index=index)
            self.index = self.geometry.index

    frame = Frame([Geom(m) for m in gj], pd.Index(["a", "b", "c"], name="region"))
    np.testing.assert_allclose(gis.compute_indicatormatrix(x, y, frame).toarray(), ref, rtol=0, atol=1e-14)
    with pytest.raises(ValueError, match="not a polygon"):
        gis.compute_indicatormatrix(x, y, [{"type": "LineString", "coordinates": [[0, 0], [1, 1]]}])


def test_rated_capacity_helpers():
    """resource.solarpanel_rated_capacity_per_unit / windturbine_rated_capacity_per_unit (atlite/resource.py:204-224)."""
    from atlite_amd import resource as r

    assert r.solarpanel_rated_capacity_per_unit("CSi") == r.get_solarpanelconfig("CSi")["efficiency"] == 0.1
    k = r.get_solarpanelconfig("KANENA")
    assert r.solarpanel_rated_capacity_per_unit(k) == (k["A"] + k["B"] * 1000.0 + k["C"] * np.log(1000.0)) * 1e3
    assert r.windturbine_rated_capacity_per_unit("Vestas_V112_3MW") == r.get_windturbineconfig("Vestas_V112_3MW")["P"]


def test_turbine_and_panel_catalogues_like_the_reference():
    """atlite.windturbines / atlite.solarpanels (resource.py:514-515): attribute and item access, iteration; the values
    are what the turbine= / panel= arguments accept."""
    import atlite_amd
    from atlite_amd import resource

    wt = atlite_amd.windturbines
    assert wt is resource.windturbines and wt.Vestas_V112_3MW == wt["Vestas_V112_3MW"] == "Vestas_V112_3MW"
    assert len(wt) == 30 and "Vestas_V112_3MW" in wt and sorted(wt) == wt() and "Vestas_V112_3MW" in dir(wt)
    assert resource.get_windturbineconfig(wt.Vestas_V112_3MW)["P"] == 3.06
    assert atlite_amd.solarpanels() == ["CSi", "CdTe", "KANENA"] and atlite_amd.solarpanels.KANENA == "KANENA"
    assert resource.get_solarpanelconfig(atlite_amd.solarpanels.CSi)["efficiency"] == 0.1
    with pytest.raises(AttributeError):
        wt.no_such_turbine
    with pytest.raises(KeyError):
        wt["no_such_turbine"]
    # atlite.compute_indicatormatrix(cutout.grid, shapes) - the reference's calling convention
    from atlite_amd import Cutout, Dataset, gis
    import pandas as pd

    x, y = np.linspace(0.0, 4.0, 5), np.linspace(40.0, 42.0, 3)
    c = Cutout(Dataset({"temperature": np.zeros((2, 3, 5))}, dict(time=pd.date_range("2013-01-01", periods=2, freq="h"), y=y, x=x)))
    tri = np.array([[0.2, 39.8], [3.7, 40.3], [1.9, 42.2]])
    a = atlite_amd.compute_indicatormatrix(c.grid, [tri])
    np.testing.assert_array_equal(a.toarray(), gis.compute_indicatormatrix(x, y, [tri]).toarray())
    # shapes in a projected crs: the cell corners go through atlite_amd.crs (tests/test_crs.py); other codes are refused
    from atlite_amd import crs

    tri_laea = np.stack(crs.forward(3035, tri[:, 0], tri[:, 1]), axis=1)
    b = atlite_amd.compute_indicatormatrix(c.grid, [tri_laea], 4326, 3035)
    assert b.shape == a.shape and abs(b.sum() - a.sum()) < 0.02 * a.sum() and np.abs((b - a).toarray()).max() < 0.1
    with pytest.raises(NotImplementedError):
        atlite_amd.compute_indicatormatrix(c.grid, [tri], 4326, 27700)
    with pytest.raises(NotImplementedError):
        atlite_amd.compute_indicatormatrix(c.grid.iloc[::-1], [tri])


def test_cutout_sel_time_and_space():
    """Cutout.sel (cutout.py:387-414): label slices are inclusive on both ends like xarray's, a contiguous time range is
    a view, lists of snapshots and spatial windows copy; bounds + buffer; unknown labels raise KeyError."""
    import pandas as pd

    from atlite_amd import Cutout, Dataset

    x, y = np.linspace(5.0, 8.0, 13), np.linspace(47.0, 49.0, 9)
    t = pd.date_range("2013-01-01", periods=72, freq="h")
    rng = np.random.default_rng(0)
    temp, height = rng.random((72, 9, 13)), rng.random((9, 13))
    c = Cutout(Dataset({"temperature": temp, "height": height}, dict(time=t, y=y, x=x), attrs={"module": "era5"}))
    a = c.sel(time=slice("2013-01-02", "2013-01-02 23:00"))
    assert list(a.coords["time"]) == list(t[24:48]) and a.data["height"].values is not None
    assert np.shares_memory(np.asarray(a.data["temperature"].values), temp)  # a view
    np.testing.assert_array_equal(a.data["temperature"].values, temp[24:48])
    assert len(c.sel(time="2013-01-03").coords["time"]) == 24  # a day string selects the day, like pandas / xarray
    snaps = t[::7]
    b = c.sel(time=snaps)
    np.testing.assert_array_equal(b.data["temperature"].values, temp[::7])
    assert list(b.coords["time"]) == list(snaps) and b.data.attrs["module"] == "era5"
    w = c.sel(x=slice(5.5, 6.75), y=slice(47.25, 48.0))
    assert w.shape == (4, 6) and w.coords["x"][0] == 5.5 and w.coords["x"][-1] == 6.75 and w.coords["lon"][0] == 5.5
    np.testing.assert_array_equal(w.data["temperature"].values, temp[:, 1:5, 2:8])
    np.testing.assert_array_equal(w.data["height"].values, height[1:5, 2:8])
    bb = c.sel(bounds=(6.0, 47.5, 6.5, 48.0), buffer=0.25, time=slice(t[3], t[10]))
    np.testing.assert_array_equal(bb.data["temperature"].values, temp[3:11, 1:6, 3:8])
    assert bb.dx == 0.25 and np.allclose(bb.bounds, [5.625, 47.125, 6.875, 48.375])
    with pytest.raises(KeyError):
        c.sel(time=[pd.Timestamp("2014-01-01")])
    with pytest.raises(KeyError):
        c.sel(level=3)
    assert c.sel(time=slice("2015", "2016")).data.sizes["time"] == 0


def test_labeled_array_arithmetic_and_sel():
    """Without xarray the results are LabeledArrays: elementwise arithmetic aligned by dimension name (new dimensions
    appended in order of first appearance, attributes dropped), comparisons, label selection."""
    import pandas as pd

    from atlite_amd import LabeledArray

    t = pd.date_range("2013-01-01", periods=4, freq="h")
    rng = np.random.default_rng(0)
    a = LabeledArray(rng.random((3, 4)), ("region", "time"), {"region": pd.Index(["DE", "FR", "PL"], name="region"), "time": t},
                     {"units": "MW"}, "power")
    cap = LabeledArray(np.array([2.0, 4.0, 0.0]), ("region",), {"region": pd.Index(["DE", "FR", "PL"])}, {"units": "MW"}, "cap")
    with np.errstate(all="ignore"):
        pu = a / cap
    assert pu.dims == ("region", "time") and pu.attrs == {} and pu.name is None
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(pu.values, a.values / cap.values[:, None])
    np.testing.assert_array_equal((2 * a + 1).values, 2 * a.values + 1)
    np.testing.assert_array_equal((1 - a).values, 1 - a.values)
    np.testing.assert_array_equal((a.values * a).values, a.values ** 2)  # ndarray on the left defers to LabeledArray
    tt = LabeledArray(np.arange(4.0), ("time",), {"time": t})
    np.testing.assert_array_equal((tt * cap).values, np.outer(np.arange(4.0), cap.values))
    assert (tt * cap).dims == ("time", "region") and (cap * tt).dims == ("region", "time")
    assert ((a > 0.5).values == (a.values > 0.5)).all() and (-a).values[0, 0] == -a.values[0, 0] and abs(-a).max() == a.max()
    np.testing.assert_array_equal(a.max("time").values, a.values.max(1))
    with pytest.raises(ValueError, match="disagree"):
        a + LabeledArray(np.zeros(5), ("time",))
    # label selection
    np.testing.assert_array_equal(a.sel(region="FR").values, a.values[1])
    assert a.sel(region="FR").dims == ("time",)
    np.testing.assert_array_equal(a.sel(region=["PL", "DE"]).values, a.values[[2, 0]])
    np.testing.assert_array_equal(a.sel(time=slice(t[1], t[2])).values, a.values[:, 1:3])
    with pytest.raises(KeyError):
        a.sel(region="ES")


def test_cutout_merge_same_grid():
    """Cutout.merge (cutout.py:416-450) for cutouts on the same grid and time axis: the union of the variables, module
    and prepared_features united; other coordinates would need xarray's outer join."""
    import pandas as pd

    from atlite_amd import Cutout, Dataset

    x, y = np.linspace(5.0, 8.0, 4), np.linspace(47.0, 49.0, 3)
    t = pd.date_range("2013-01-01", periods=5, freq="h")
    rng = np.random.default_rng(0)
    temp, wnd, h = rng.random((5, 3, 4)), rng.random((5, 3, 4)), rng.random((3, 4))
    a = Cutout(Dataset({"temperature": temp, "height": h}, dict(time=t, y=y, x=x), attrs={"module": "era5"}))
    b = Cutout(Dataset({"wnd100m": wnd, "height": h}, dict(time=t, y=y, x=x), attrs={"module": "sarah"}))
    m = a.merge(b)
    assert set(m.data.data_vars) == {"temperature", "height", "wnd100m"} and m.module == ["era5", "sarah"]
    assert sorted(m.data.attrs["prepared_features"]) == ["height", "temperature", "wnd100m"]
    np.testing.assert_array_equal(m.data["wnd100m"].values, wnd)
    assert m.data["height"].dims == ("y", "x") and m.equals(b.merge(a))
    with pytest.raises(NotImplementedError):
        a.merge(b.sel(x=slice(5.0, 7.0)))
    c = Cutout(Dataset({"height": h + 1.0}, dict(time=t, y=y, x=x)))
    with pytest.raises(ValueError, match="conflicting"):
        a.merge(c)


def test_hostile_offsets_are_refused_before_they_are_followed():
    """Findings of the long host fuzzers (tools/fuzz_plan.py, tools/fuzz_gis.py under ASan + UBSan): CSR row pointers and
    the shape / ring offsets of the polygon entry points come from the caller - a non-monotone INTERIOR offset used to be
    followed past the arrays before the per-row check noticed; polygon coordinates of 1e300 / inf went through an
    undefined double -> int64 conversion.  Both are refused resp. defined now."""
    import ctypes as C

    from atlite_amd import _lib, gis

    lib = _lib.load()
    # CSR: 3 rows, 4 entries, the middle pointer far beyond the data
    indptr = np.array([0, 2, 10**9, 4], dtype=np.int64)
    indices = np.array([0, 1, 2, 3], dtype=np.int32)
    data = np.ones(4)
    err = C.c_int64()
    with pytest.raises(ValueError, match="monotone"):
        _lib.check(lib.atl_agg_check_host(3, 8, 0, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, None, None, C.byref(err)))
    # polygons: one shape, two rings, the middle ring offset beyond the vertex array / negative
    xy = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0], [0.2, 0.2], [0.4, 0.2], [0.4, 0.4]])
    shape_ptr = np.array([0, 2], dtype=np.int64)
    holes = np.array([0, 1], dtype=np.uint8)
    for bad in ([0, 10**9, 7], [0, -5, 7], [3, 2, 7]):
        ring_ptr = np.array(bad, dtype=np.int64)
        for fn in (lib.atl_indicator_polygons, lib.atl_indicator_polygons_integral_host):
            p_ip, p_ix, p_d = C.c_void_p(), C.c_void_p(), C.c_void_p()
            with pytest.raises(ValueError, match="offsets"):
                _lib.check(fn(1, shape_ptr.ctypes.data, 2, ring_ptr.ctypes.data, holes.ctypes.data, xy.ctypes.data, 4, 4, 0.0, 1.0, 0.0, 1.0,
                              C.byref(p_ip), C.byref(p_ix), C.byref(p_d)))
    # coordinates far outside the 64-bit index range: no entries, no undefined conversion (the sanitizer suite runs this too)
    x, y = np.arange(5.0), np.arange(4.0)
    for ctx in (None, "integral-host"):
        M = gis.compute_indicatormatrix(x, y, [np.array([[1e300, 0.0], [2e300, 0.0], [2e300, 1e300]]),
                                               np.array([[-np.inf, 0.0], [1.0, 0.0], [1.0, 1.0]]),
                                               np.array([[0.5, 0.5], [2.5, 0.5], [2.5, 2.5], [0.5, 2.5]])], ctx=ctx).toarray()
        assert M[0].sum() == 0.0 and np.isfinite(M).all() and abs(M[2].sum() - 4.0) < 1e-12


def test_a_nan_vertex_in_the_middle_of_a_ring_empties_the_shape_in_both_algorithms():
    """std::min / std::max drop a NaN operand, so a NaN vertex that is not the first one used to leave a valid
    bounding box around a ring with two broken edges (NaN areas in one column, wrong areas elsewhere).  A shape with any
    non-finite coordinate gets an empty row - from the host clipper and from the line-integral algorithm alike."""
    from atlite_amd import gis

    x, y = np.arange(6.0), np.arange(5.0)
    good = np.array([[0.5, 0.5], [3.5, 0.5], [3.5, 2.5], [0.5, 2.5]])
    for bad_vertex in (np.nan, np.inf, -np.inf):
        for k in range(4):  # the non-finite vertex at every position of the ring, x or y
            for xy in (0, 1):
                ring = good.copy()
                ring[k, xy] = bad_vertex
                A = gis.compute_indicatormatrix(x, y, [good, ring, good + 1.0], ctx=None).toarray()
                B = gis.compute_indicatormatrix(x, y, [good, ring, good + 1.0], ctx="integral-host").toarray()
                for M in (A, B):
                    assert np.isfinite(M).all() and not M[1].any() and M[0].any() and M[2].any()
                np.testing.assert_allclose(A, B, atol=1e-12)


def test_pinned_result_blocks_are_pooled_and_outlive_their_views():
    """device._PinnedBlock: a download's destination is page-locked memory wrapped as the base of the NumPy array the caller
    gets; views keep it alive, the block returns to a pool when the last one dies and the next download of that size reuses
    it (no GPU needed: the allocator is the only library call)."""
    import ctypes as C
    import gc

    from atlite_amd import device

    class FakeLib:
        def __init__(self):
            self.live, self.allocs = {}, 0

        def atl_pinned_alloc(self, n, pp):
            buf = (C.c_char * n)()
            self.live[C.addressof(buf)] = buf
            pp._obj.value = C.addressof(buf)
            self.allocs += 1
            return 0

        def atl_pinned_free(self, p):
            self.live.pop(p)
            return 0

    lib = FakeLib()
    device._PinnedBlock._pool.clear()
    device._PinnedBlock._pooled = 0
    a = device._host_array(lib, (100, 1000), np.float64)
    assert isinstance(a, np.ndarray) and a.flags.writeable and isinstance(a.base, device._PinnedBlock)
    a[:] = 3.0
    view = a[5:10]
    del a
    gc.collect()
    assert view.sum() == 15000.0 and device._PinnedBlock._pooled == 0  # the view keeps the block
    del view
    gc.collect()
    assert device._PinnedBlock._pooled == 800000 and lib.allocs == 1
    b = device._host_array(lib, (1000, 100), np.float64)  # same size: the pooled block
    assert lib.allocs == 1 and device._PinnedBlock._pooled == 0
    small = device._host_array(lib, (10,), np.float64)  # below 64 KiB: ordinary memory
    assert small.base is None and lib.allocs == 1
    del b
    gc.collect()
    device._PinnedBlock._pool.clear()
    device._PinnedBlock._pooled = 0


def test_pinned_results_respect_the_budget_and_trim(monkeypatch):
    """Past ``_PinnedBlock.BUDGET`` page-locked bytes a download's destination is ordinary memory; ``trim`` frees the pool."""
    import ctypes as C
    import gc

    from atlite_amd import device

    class FakeLib:
        def __init__(self):
            self.live = {}

        def atl_pinned_alloc(self, n, pp):
            buf = (C.c_char * n)()
            self.live[C.addressof(buf)] = buf
            pp._obj.value = C.addressof(buf)
            return 0

        def atl_pinned_free(self, p):
            self.live.pop(p)
            return 0

    lib = FakeLib()
    device._PinnedBlock.trim(lib)
    monkeypatch.setattr(device._PinnedBlock, "_live", 0)
    monkeypatch.setattr(device._PinnedBlock, "BUDGET", 2_000_000)
    a = device._host_array(lib, (100, 1000), np.float64)  # 800 kB
    b = device._host_array(lib, (100, 1000), np.float64)
    c = device._host_array(lib, (100, 1000), np.float64)  # would be 2.4 MB page-locked: ordinary memory
    assert isinstance(a.base, device._PinnedBlock) and isinstance(b.base, device._PinnedBlock) and c.base is None
    assert device._PinnedBlock._live == 1_600_000 and len(lib.live) == 2
    del a, b
    gc.collect()
    assert device._PinnedBlock._pooled == 1_600_000 and device._PinnedBlock._live == 1_600_000
    device._PinnedBlock.trim(lib)
    assert device._PinnedBlock._pooled == 0 and device._PinnedBlock._live == 0 and not lib.live


def test_device_block_recycling_is_ordered_by_events(monkeypatch):
    """Context.empty / DeviceArray.free: a released block is handed out again only after the events recorded on the compute
    and the copy stream at its release have been waited for; releases from another thread are queued for the owner;
    ``ATLITE_HIP_RECYCLE=0`` and blocks marked ``no_recycle()`` go straight back to the driver; an allocation that fails for
    lack of memory drains the pool and is retried."""
    import threading

    from atlite_amd import _lib, device

    class FakeLib:
        def __init__(self):
            self.log, self.next, self.fail_once = [], 0x1000, False
            self.freed = []

        def atl_alloc(self, h, n, pp):
            if self.fail_once:
                self.fail_once = False
                return _lib.ATL_E_NOMEM
            self.next += 0x1000
            pp._obj.value = self.next
            self.log.append(("alloc", self.next))
            return 0

        def atl_free(self, h, p):
            self.freed.append(p)
            return 0

        def atl_event_create(self, h, pe):
            self.next += 1
            pe._obj.value = self.next
            return 0

        def atl_event_record(self, h, ev, which):
            self.log.append(("record", ev.value, which))
            return 0

        def atl_event_synchronize(self, ev):
            self.log.append(("wait", ev.value))
            return 0

        def atl_event_destroy(self, ev):
            return 0

        def atl_destroy(self, h):
            return 0

    def make():
        ctx = device.Context.__new__(device.Context)
        ctx.lib, ctx.handle, ctx.device = FakeLib(), object(), 0
        return ctx

    monkeypatch.delenv("ATLITE_HIP_RECYCLE", raising=False)
    monkeypatch.delenv("ATLITE_HIP_FENCE", raising=False)
    ctx = make()
    a = ctx.empty((1000,))
    pa = a.ptr
    del a  # released by the owner: events recorded on both streams right away
    recs = [e for e in ctx.lib.log if e[0] == "record"]
    # (2 = the copy stream as a fence: ordered behind the device-inflate reads in flight without consuming their verdicts)
    assert [r[2] for r in recs] == [0, 2] and not ctx.lib.freed
    b = ctx.empty((1000,))  # same size: the pooled block, after both events were waited for
    waits = [e[1] for e in ctx.lib.log if e[0] == "wait"]
    assert b.ptr == pa and waits == [recs[0][1], recs[1][1]]
    c = ctx.empty((999,))  # another size: a fresh allocation
    assert c.ptr != pa
    # released by another thread (a garbage collector's __del__): queued, the owner records the events at its next allocation
    n_rec = len([e for e in ctx.lib.log if e[0] == "record"])
    t = threading.Thread(target=b.free)
    t.start()
    t.join()
    assert len([e for e in ctx.lib.log if e[0] == "record"]) == n_rec and ctx._pool_st["deferred"]
    d = ctx.empty((1000,))
    assert d.ptr == pa and len([e for e in ctx.lib.log if e[0] == "record"]) == n_rec + 2 and not ctx._pool_st["deferred"]
    # no_recycle(): back to the driver
    d.no_recycle().free()
    assert ctx.lib.freed == [pa]
    # out of memory: the pool is drained, the allocation retried
    c.free()
    assert ctx._pool_st["held"] == 999 * 8
    ctx.lib.fail_once = True
    e = ctx.empty((5,))
    assert e.ptr and ctx._pool_st["held"] == 0 and len(ctx.lib.freed) == 2
    ctx.close()
    # switched off
    monkeypatch.setenv("ATLITE_HIP_RECYCLE", "0")
    ctx2 = make()
    x = ctx2.empty((1000,))
    px = x.ptr
    del x
    assert ctx2.lib.freed == [px] and not [e for e in ctx2.lib.log if e[0] == "record"]
    ctx2.close()


def test_bench_counts_the_lines_a_ragged_plan_touches():
    """bench.py's star-polygon leg prices the kernel on the 128-byte lines that hold a covered cell (what the memory system
    fetches when lanes skip unweighted cells), next to the covered cells' own bytes."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench

    cov = np.zeros(64, bool)
    assert bench.lines_touched(cov, 64) == 0
    cov[[0, 15, 16, 47]] = True  # lines 0, 1, 2
    assert bench.lines_touched(cov, 64) == 3
    assert bench.lines_touched(np.ones(40, bool), 48) == 3  # 40 cells in slots of 48: the last line is half used
    assert bench.lines_touched(np.ones(17, bool), 17) is None  # slots that do not start on a line: no single count
    # BASELINE's star polygons on C2's grid: the figure DESIGN.md quotes (2160 lines of 2500, 23758 cells of 40000)
    from atlite_amd import gis

    X = Y = 200
    x, y = -25 + (70 / X) * np.arange(X), 30 + (42 / Y) * np.arange(Y)
    dx, dy = x[1] - x[0], y[1] - y[0]
    polys = gis.random_star_polygons(100, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
    M = gis.compute_indicatormatrix(x, y, polys)
    mask = np.asarray((M != 0).sum(0)).ravel() > 0
    assert int(mask.sum()) == 23758 and bench.lines_touched(mask, X * Y) == 2160
