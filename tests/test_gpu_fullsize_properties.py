"""
GPU, BASELINE.json configs[1] at FULL size (8760 x 200 x 200 fp64, 100 shapes, 19.6 GB of inputs
generated on the device): size-independent properties of the fused convert+aggregate path.
The oracle cannot run at this size in seconds; the same run is spot-checked against it on a
sample of time steps (as bench.py does).
"""
import numpy as np
import pytest

from atlite_amd import gis, synthetic
from oracle import atlite_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
T, Y, X, N = 8760, 200, 200, 100
PARAMS = dict(H.CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))


@pytest.fixture(scope="module", params=["one allocation per cube", "slot-interleaved"])
def c2(ctx, request):
    # both residencies: a caller's own device arrays, and the layout of the library's own device copies (device.SlotPool)
    inputs, coords = synthetic.pv_inputs(ctx, T, Y, X, interleaved=request.param == "slot-interleaved")
    x, y = coords["x"], coords["y"]
    dx, dy = x[1] - x[0], y[1] - y[0]
    polys = gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42)
    M = gis.compute_indicatormatrix(x, y, polys)
    out = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M, row_len=X), options=dict(night_skip=False)).numpy()
    yield inputs, M, out
    del inputs


def test_checksum_of_checksums(ctx, c2):
    """The tessellation's column sums are 1, so summing the aggregated series over shapes and time
    must equal the per-cell time sums (a different kernel: k_cells_timered) summed over cells."""
    inputs, M, out = c2
    np.testing.assert_allclose(np.asarray(M.sum(0)).ravel(), 1.0, atol=1e-11)
    cells = ctx.pv(inputs, PARAMS, T, Y * X, time_agg="sum").numpy()
    assert np.isfinite(out).all() and out.min() >= 0.0
    np.testing.assert_allclose(out.sum(), cells.sum(), rtol=1e-10)
    # per shape: time-sum of the series == M @ (per-cell time sums)
    np.testing.assert_allclose(out.sum(1), M @ cells, rtol=1e-10)
    # and the fused time reductions agree with reducing the series
    np.testing.assert_allclose(ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M, row_len=X), time_agg="mean",
                                      options=dict(night_skip=False)).numpy(), out.mean(1), rtol=1e-12)


def test_linearity_permutation_determinism(ctx, c2):
    inputs, M, out = c2
    # scaling the weights by 2 is exact in fp64 -> bit-identical doubling
    twice = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(2.0 * M, row_len=X), options=dict(night_skip=False)).numpy()
    np.testing.assert_array_equal(twice, 2.0 * out)
    # permuting the shapes permutes the rows, bit for bit (fixed reduction tree, no atomics)
    perm = np.random.default_rng(0).permutation(N)
    pout = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M[perm], row_len=X), options=dict(night_skip=False)).numpy()
    np.testing.assert_array_equal(pout, out[perm])
    # run-to-run determinism, and the night early-out changes nothing
    again = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M, row_len=X), options=dict(night_skip=False)).numpy()
    np.testing.assert_array_equal(again, out)
    skip = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M, row_len=X), options=dict(night_skip=True)).numpy()
    np.testing.assert_array_equal(skip, out)
    # a different tile shape changes the summation tree, not the result beyond rounding
    flat = ctx.pv(inputs, PARAMS, T, Y * X, plan=ctx.plan(M), options=dict(night_skip=False)).numpy()
    np.testing.assert_allclose(flat, out, rtol=1e-12, atol=1e-12 * out.max())


def test_night_is_exactly_zero_and_sample_matches_oracle(ctx, c2):
    inputs, M, out = c2
    sel = np.concatenate([np.arange(0, 72), np.arange(4300, 4372)])
    host = {k: np.stack([v.slab(int(t), int(t) + 1).numpy()[0] for t in sel]) for k, v in inputs.items()}
    dark = (host["solar_altitude"] < np.radians(1.0)).all(axis=1)
    assert dark.any() and (~dark).any()
    assert (out[:, sel[dark]] == 0.0).all()
    ref = orc.aggregate_matrix(orc.convert_pv(host, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0))), M)
    np.testing.assert_allclose(out[:, sel], ref, rtol=1e-10, atol=1e-12 * ref.max())


def test_per_cell_kernels_with_the_night_early_out(ctx, c2):
    """Capacity-factor map and per-cell series at full size (k_cells_night: what Cutout.pv() without shapes runs):
    the same bits as the kernels without the early-out, the map equal to the reduced series, sampled steps equal to
    the oracle."""
    inputs, M, out = c2
    S = Y * X
    a = ctx.pv(inputs, PARAMS, T, S, time_agg="mean", options=dict(night_skip=False)).numpy()
    b = ctx.pv(inputs, PARAMS, T, S, time_agg="mean", options=dict(night_skip=True, row_len=X)).numpy()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, ctx.pv(inputs, PARAMS, T, S, time_agg="mean", options=dict(night_skip=True)).numpy())  # strips
    ser = ctx.pv(inputs, PARAMS, T, S, options=dict(night_skip=True, row_len=X)).numpy()
    assert ser.shape == (T, S) and np.isfinite(ser).all()
    np.testing.assert_allclose(b, ser.mean(0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(M @ ser.T, out, rtol=1e-11, atol=1e-12 * out.max())  # and the fused path agrees
    sel = np.concatenate([np.arange(0, 40), np.arange(4300, 4340), [T - 1]])
    host = {k: np.stack([v.slab(int(t), int(t) + 1).numpy()[0] for t in sel]) for k, v in inputs.items()}
    ref = orc.convert_pv(host, H.CSI, dict(slope=np.radians(30.0), azimuth=np.radians(180.0)))
    np.testing.assert_allclose(ser[sel], ref, rtol=1e-10, atol=1e-12 * ref.max())
    dark = (host["solar_altitude"] < np.radians(1.0)).all(axis=1)
    assert dark.any() and (ser[sel[dark]] == 0.0).all()
    nos = ctx.pv(inputs, PARAMS, T, S, options=dict(night_skip=False))
    np.testing.assert_array_equal(nos.slab(4300, 4400).numpy(), ser[4300:4400])


def test_the_two_residencies_give_the_same_bits(ctx):
    sep, coords = synthetic.pv_inputs(ctx, T, Y, X)
    il, _ = synthetic.pv_inputs(ctx, T, Y, X, interleaved=True)
    assert next(iter(il.values())).ld == 7 * Y * X and next(iter(sep.values())).ld is None
    x, y = coords["x"], coords["y"]
    dx, dy = x[1] - x[0], y[1] - y[0]
    M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0] - dx / 2, y[0] - dy / 2, x[-1] + dx / 2, y[-1] + dy / 2), seed=42))
    plan = ctx.plan(M, row_len=X)
    for skip in (False, True):
        a = ctx.pv(sep, PARAMS, T, Y * X, plan=plan, options=dict(night_skip=skip)).numpy()
        b = ctx.pv(il, PARAMS, T, Y * X, plan=plan, options=dict(night_skip=skip)).numpy()
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ctx.pv(sep, PARAMS, T, Y * X, time_agg="mean").numpy(), ctx.pv(il, PARAMS, T, Y * X, time_agg="mean").numpy())
