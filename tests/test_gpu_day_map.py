"""GPU: the night early-out's day map (atl_pv_day_map, round 5).  The fused pv kernel with night_skip normally loads a
tile's altitudes and votes; with a day map - the same votes, computed once per (plan, altitude cube, cut-off) and kept with
the cube's allocation - it reads one byte per batch instead.  Contract: the SAME BITS as the voting kernel and as the
kernel without the early-out (reference: atlite/pv/irradiation.py:251-252 - below the cut-off the result is 0 whatever the
other cubes hold)."""
import numpy as np
import pytest

from atlite_amd import Cutout, Dataset, gis
from atlite_amd.device import root_block
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
PV_PARAMS = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))


def up(ctx, ds):
    return {k: ctx.upload(v) for k, v in ds.items()}


@pytest.mark.parametrize("T,Y,X", [(72, 9, 16), (61, 17, 33), (130, 8, 200), (7, 40, 7), (65, 1, 130), (200, 32, 48)])
def test_day_map_gives_the_voting_kernels_bits(ctx, T, Y, X):
    S, N = Y * X, 5
    ds = H.pv_dataset(T, Y, X, seed=T)
    ds["solar_altitude"][min(40, T - 1), 3 % S] = np.nan  # a NaN altitude is not "night"
    ds["temperature"][2 % T, :] = np.nan                 # night rows: NaN inputs must not leak either way
    ds["influx_direct"][3 % T, :] = np.inf
    M = H.blob_matrix(N, Y, X, seed=10, overlap=False)    # leaves cells uncovered: tiles whose covered cells are all dark
    plan = ctx.plan(M, row_len=X)
    dev = up(ctx, ds)
    _, y = H.grid(Y, X)
    lo = orc.orientation_latitude_optimal(np.radians(y))
    per_cell = dict(H.CSI, slope=np.repeat(lo["slope"], X), azimuth=np.repeat(lo["azimuth"], X))
    for params in (PV_PARAMS, per_cell):
        for trigon in ("simple", "other"):
            for agg in (None, "sum", "mean"):
                o = dict(trigon_model=trigon)
                full = ctx.pv(dev, params, T, S, plan=plan, time_agg=agg, options=dict(o, night_skip=False)).numpy()
                vote = ctx.pv(dev, params, T, S, plan=plan, time_agg=agg, options=dict(o, night_skip=True, day_map=False)).numpy()
                mapped = ctx.pv(dev, params, T, S, plan=plan, time_agg=agg, options=dict(o, night_skip=True, day_map=True)).numpy()
                np.testing.assert_array_equal(vote, full)
                np.testing.assert_array_equal(mapped, full)
    maps = root_block(dev["solar_altitude"]).__dict__.get("_day_maps", {})
    assert len(maps) == 1  # one (plan, cube, cut-off): built once, found again by every later call
    ref = orc.aggregate_matrix(orc.convert_pv(ds, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    got = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(night_skip=True, day_map=True)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref[np.isfinite(ref)])), equal_nan=True)


def test_day_map_bits_are_the_votes(ctx):
    """The map itself, against NumPy: byte t of tile s is non-zero iff some covered cell of the tile has an altitude that is not
    below the cut-off (NaN counts as day), for a cut-off other than the default as well.  (Its bits are the tile's 128-byte
    lines: test_line_bits_follow_the_terminator.)"""
    T, Y, X = 77, 24, 40
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=3)
    ds["solar_altitude"][5, 17] = np.nan
    M = H.blob_matrix(4, Y, X, seed=2, overlap=False)
    covered = np.asarray(M.sum(axis=0)).ravel() != 0
    plan = ctx.plan(M, row_len=X)
    dev = up(ctx, ds)
    for thr in (np.radians(1.0), np.radians(-3.0), 0.4):
        params = dict(PV_PARAMS, altitude_threshold=thr)
        ctx.pv(dev, params, T, S, plan=plan, options=dict(night_skip=True, day_map=True))
        maps = root_block(dev["solar_altitude"])._day_maps
        _, dmap, ld = next(v for k, v in maps.items() if k[-1] == float(thr))
        bits = dmap.numpy().reshape(-1, ld)
        day_any = np.zeros(T, bool)
        for s in range(bits.shape[0]):
            for t in range(T):
                if bits[s, t]:
                    day_any[t] = True
        alt = ds["solar_altitude"]
        # over ALL tiles: no day of a covered cell is missed; a step in which every cell of the grid is dark has no bit (the vote
        # is per lane = pair of adjacent cells, so the uncovered partner of a covered cell may add a day, never hide one)
        covered_day = (~(alt < thr) & covered[None, :]).any(axis=1)
        any_day = (~(alt < thr)).any(axis=1)
        assert (day_any | ~covered_day).all() and (~day_any | any_day).all() and day_any.any() and not day_any.all()
        assert not bits[:, T:].any()  # nothing behind the last time step


def test_line_bits_follow_the_terminator(ctx):
    """Line granularity (round 6): a grid of ONE 16 x 8 tile row band whose altitude is a step function of x - west half up, east
    half down - must light exactly the lines (16 consecutive cells) that hold a cell of the west half, in every covered tile;
    the result stays bit-identical to the kernel without the early-out while the dark lines hold NaN / inf in every other cube
    (they are not read: were they, NaN would reach the shapes' sums through 0 * NaN)."""
    T, Y, X = 24, 8, 64
    S = Y * X
    ds = H.pv_dataset(T, Y, X, seed=5)
    alt = np.full((T, Y, X), -0.5)
    alt[:, :, :24] = 0.6  # the first line and a half of every row is up
    ds["solar_altitude"] = alt.reshape(T, S)
    M = H.blob_matrix(3, Y, X, seed=1, overlap=True)
    import scipy.sparse as sp

    M = sp.csr_matrix(np.ones((1, S)))  # every cell covered: the coverage mask plays no part here
    plan = ctx.plan(M, row_len=X)
    dev = up(ctx, ds)
    full = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(night_skip=False)).numpy()
    mapped = ctx.pv(dev, PV_PARAMS, T, S, plan=plan, options=dict(night_skip=True, day_map=True)).numpy()
    np.testing.assert_array_equal(mapped, full)
    _, dmap, ld = next(iter(root_block(dev["solar_altitude"])._day_maps.values()))
    bits = dmap.numpy().reshape(-1, ld)[:, :T]
    assert (bits == bits[:, :1]).all()  # the same every step
    # every 128-byte line (16 consecutive cells; X is a multiple of 16) belongs to exactly one tile: the lit lines over all tiles
    # are the two westernmost lines of each of the Y rows, and no tile is lit as a whole
    lit = sum(bin(int(b)).count("1") for b in bits[:, 0])
    assert lit == 2 * Y, (lit, [bin(int(b)) for b in bits[:, 0]])
    assert all(int(b) != 0xFF for b in bits[:, 0])


def test_day_map_follows_the_dataset(ctx):
    """Cutout.pv(): the dataset's own device copies carry day maps (second call: same bits, no rebuild); replacing the
    altitude cube drops them with the old copy."""
    T, Y, X = 48, 12, 20
    ds_np = H.pv_dataset(T, Y, X, seed=21)
    x, y = H.grid(Y, X)
    coords = {"time": H.times(T), "y": y, "x": x}
    ds = Dataset(dict(ds_np), coords)
    M = H.blob_matrix(3, Y, X, seed=4)
    kw = dict(panel="CSi", orientation={"slope": 30.0, "azimuth": 180.0}, matrix=M, aggregate_time=None)
    a = Cutout(ds).pv(**kw).values
    b = Cutout(ds).pv(**kw).values
    np.testing.assert_array_equal(a, b)
    ref = orc.aggregate_matrix(orc.convert_pv(ds_np, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    np.testing.assert_allclose(np.asarray(a).T if a.shape != ref.shape else a, ref, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref)))
    alt2 = ds_np["solar_altitude"] - 0.3  # an earlier sunset everywhere
    ds["solar_altitude"] = alt2
    c = Cutout(ds).pv(**kw).values
    ref2 = orc.aggregate_matrix(orc.convert_pv(dict(ds_np, solar_altitude=alt2), H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    np.testing.assert_allclose(np.asarray(c).T if c.shape != ref2.shape else c, ref2, rtol=1e-10, atol=1e-12 * np.nanmax(np.abs(ref2)))
