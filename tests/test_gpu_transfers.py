"""GPU: host <-> device transfers through the context's page-locked bounce buffers (atl::h2d / d2h / h2d_2d / d2h_2d, round 5).
Host memory that is not page-locked - every NumPy array a caller hands in, every small result - never reaches the HIP runtime
as it lies (the runtime would pin the pages on the fly): it travels in 4-MiB slices through two
page-locked buffers.  Round trips must be exact for sizes around the slice boundaries, pitched 2-d copies, rows wider than a
buffer, and page-locked memory must still be used directly."""
import ctypes as C

import numpy as np
import pytest

from atlite_amd._lib import check

pytestmark = pytest.mark.gpu
SLICE = 4 << 20


@pytest.mark.parametrize("nbytes", [8, 4096, SLICE - 8, SLICE, SLICE + 8, 2 * SLICE + 24, 5 * SLICE + 4096 + 8])
def test_round_trip_of_pageable_arrays(ctx, nbytes):
    rng = np.random.default_rng(nbytes)
    a = rng.random(nbytes // 8)
    d = ctx.upload(a)
    a_copy = a.copy()
    a[:] = -1.0  # the source has been read when upload returns
    back = np.empty_like(a_copy)  # pageable destination, whatever the size
    check(ctx.lib.atl_download(ctx.handle, back.ctypes.data, d.ptr, back.nbytes))
    np.testing.assert_array_equal(back, a_copy)


def test_pitched_copies_of_pageable_arrays(ctx):
    rng = np.random.default_rng(5)
    for rows, cols, ld in ((37, 243, 256), (3000, 1000, 1008), (5, 700000, 700016)):  # the last: rows wider than a slice
        a = rng.random((rows, cols))
        d = ctx.upload(a, ld=ld)  # atl_copy_2d, host -> padded device rows
        assert d.ld == ld
        back = np.empty((rows, cols))
        check(ctx.lib.atl_copy_2d(ctx.handle, back.ctypes.data, cols * 8, d.ptr, ld * 8, cols * 8, rows, 1, 0))
        np.testing.assert_array_equal(back, a)
        np.testing.assert_array_equal(d.numpy(), a)  # (page-locked result block for the big ones, pageable for the small)
        # a strided host source: every other column block
        wide = rng.random((rows, 2 * cols))
        e = ctx.empty_pitched((rows, cols), ld)
        check(ctx.lib.atl_copy_2d(ctx.handle, e.ptr, ld * 8, wide.ctypes.data, 2 * cols * 8, cols * 8, rows, 0, 0))
        np.testing.assert_array_equal(e.numpy(), wide[:, :cols])


def test_page_locked_memory_goes_as_it_lies(ctx):
    """A registered (Dataset.pin) or library-allocated page-locked block is handed to the runtime directly - no bounce - and
    asynchronous uploads from it overlap as before; results are the same either way."""
    n = 3 * SLICE // 8 + 5
    a = np.random.default_rng(1).random(n)
    p = C.c_void_p()
    check(ctx.lib.atl_pinned_alloc(a.nbytes, C.byref(p)))
    try:
        pinned = np.frombuffer((C.c_char * a.nbytes).from_address(p.value), dtype=np.float64)
        pinned[:] = a
        d = ctx.empty((n,))
        check(ctx.lib.atl_upload_async(ctx.handle, d.ptr, p.value, a.nbytes))
        ctx.copy_barrier()
        ctx.sync()
        np.testing.assert_array_equal(d.numpy(), a)
        pinned[:] = 0.0
        check(ctx.lib.atl_download(ctx.handle, p.value, d.ptr, a.nbytes))
        np.testing.assert_array_equal(pinned, a)
    finally:
        check(ctx.lib.atl_pinned_free(p))
    # pageable source through the asynchronous entry point: read before the call returns
    b = a.copy()
    d2 = ctx.empty((n,))
    check(ctx.lib.atl_upload_async(ctx.handle, d2.ptr, b.ctypes.data, b.nbytes))
    b[:] = -1.0
    ctx.copy_barrier()
    ctx.sync()
    np.testing.assert_array_equal(d2.numpy(), a)
