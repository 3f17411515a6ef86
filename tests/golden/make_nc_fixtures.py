#!/opt/conda/bin/python3.9
"""
Writes the small NetCDF-4 style HDF5 fixtures under tests/golden/nc/ that pin the native
container reader (atlite_amd/csrc/atl_h5.cpp) - run with the conda interpreter, the only one in
this image that has h5py (3.3.0 / HDF5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/make_nc_fixtures.py [outdir]

The files mimic what netCDF-C writes for an atlite cutout (atlite/data.py:139,246-248 ->
xarray.to_netcdf): creation-order tracked groups and attributes (=> new-style groups, dense link /
attribute storage once there are more than 8), dimension scales with DIMENSION_LIST, chunked
float32 variables with shuffle + deflate, CF packing attributes.  The expected decoded values go
into <name>.npz (fp64, NaN where _FillValue / missing_value / never written).  Variants cover
the old symbol-table groups (libver earliest, no order tracking) and the libver=latest chunk
indexes (single chunk, implicit, fixed array; with unlimited dimensions the extensible array - one unlimited dimension,
what netCDF-C >= 4.9 files with an unlimited time axis carry - and the v2 B-tree - several).
tests/test_nc_reader.py also calls write_case() with random shapes when this interpreter exists.
"""
import os
import sys

import h5py
import numpy as np


def _scales(f, T, Y, X, track):
    t = f.create_dataset("time", data=np.arange(T, dtype=np.int64) + 991416, track_order=track)
    t.attrs["units"] = np.string_("hours since 1900-01-01 00:00:00.0")
    t.attrs["calendar"] = np.string_("proleptic_gregorian")
    y = f.create_dataset("y", data=30.0 + 0.25 * np.arange(Y), track_order=track)
    x = f.create_dataset("x", data=-10.0 + 0.25 * np.arange(X), track_order=track)
    for d, n in ((t, "time"), (y, "y"), (x, "x")):
        d.make_scale(n)
    return t, y, x


def _attach(v, scales):
    for i, s in enumerate(scales):
        v.dims[i].attach_scale(s)


def write_case(path, T=23, Y=7, X=9, chunks=(10, 4, 5), libver=("earliest", "v108"), track=True, seed=0,
               n_extra=0, big_attrs=True, variants=True, gzip=9, unlimited=()):
    """``unlimited``: axes of the chunked (time, y, x) variables without an upper bound (maxshape None)."""
    rng = np.random.default_rng(seed)
    exp = {}
    kw = {"track_order": True} if track else {}
    with h5py.File(path, "w", libver=libver, **kw) as f:
        f.attrs["module"] = np.string_("era5")
        f.attrs["prepared_features"] = np.string_("influx,temperature,wind")
        f.attrs["dx"] = 0.25
        f.attrs["vlen_note"] = "written as a variable-length string"
        t, y, x = _scales(f, T, Y, X, track)
        exp["time"], exp["y"], exp["x"] = t[...].astype(np.float64), y[...], x[...]

        def var(name, data, scales=(t, y, x), **opts):
            if unlimited and data.ndim == 3 and opts.get("chunks"):
                opts["maxshape"] = tuple(None if i in unlimited else n for i, n in enumerate(data.shape))
            v = f.create_dataset(name, data=data, track_order=track, **opts)
            _attach(v, scales[: data.ndim] if data.ndim == 3 else scales[3 - data.ndim:])
            return v

        # the common case: float32, chunked, shuffle + deflate (edge chunks in every dimension)
        for k, name in enumerate(["influx_direct", "influx_diffuse", "temperature", "wnd100m"]):
            a = rng.random((T, Y, X), dtype=np.float32) * (300 if k < 2 else 20) + (0 if k != 2 else 270)
            v = var(name, a, chunks=chunks, compression="gzip", compression_opts=gzip, shuffle=True)
            v.attrs["units"] = np.string_("W m**-2")
            v.attrs["long_name"] = np.string_(name.replace("_", " "))
            exp[name] = a.astype(np.float64)
        # static field, contiguous float32
        h = rng.random((Y, X), dtype=np.float32) * 2000
        var("height", h)
        exp["height"] = h.astype(np.float64)
        if variants:
            # float64, deflate without shuffle
            a = rng.standard_normal((T, Y, X))
            var("roughness", a, chunks=chunks, compression="gzip", compression_opts=4)
            exp["roughness"] = a
            # CF-packed int16 with _FillValue, shuffle + deflate + fletcher32
            q = rng.integers(-32000, 32000, size=(T, Y, X)).astype(np.int16)
            q[rng.random((T, Y, X)) < 0.05] = -32767
            v = var("runoff", q, chunks=chunks, compression="gzip", shuffle=True, fletcher32=True)
            v.attrs["_FillValue"] = np.int16(-32767)
            v.attrs["scale_factor"] = np.float64(1.5e-4)
            v.attrs["add_offset"] = np.float64(4.25)
            e = q.astype(np.float64) * 1.5e-4 + 4.25
            e[q == -32767] = np.nan
            exp["runoff"] = e
            # big-endian float32 with missing_value, chunked, no filters
            b = rng.random((T, Y, X), dtype=np.float32).astype(">f4")
            b[0, 0, :3] = 9.96921e36
            v = var("albedo", b, chunks=chunks)
            v.attrs["missing_value"] = np.array(9.96921e36, dtype=">f4")
            e = b.astype(np.float64)
            e[b == np.float32(9.96921e36)] = np.nan
            exp["albedo"] = e
            # partially written variable: untouched chunks read back as _FillValue -> NaN
            v = f.create_dataset("soil_temperature", shape=(T, Y, X), dtype="f4", chunks=chunks, compression="gzip",
                                 shuffle=True, fillvalue=np.float32(-999.0), track_order=track,
                                 **({"maxshape": tuple(None if i in unlimited else n for i, n in enumerate((T, Y, X)))} if unlimited else {}))
            _attach(v, (t, y, x))
            v.attrs["_FillValue"] = np.float32(-999.0)
            part = rng.random((chunks[0], Y, X), dtype=np.float32) + 280
            v[: chunks[0]] = part
            e = np.full((T, Y, X), np.nan)
            e[: chunks[0]] = part
            exp["soil_temperature"] = e
            # uint8 / int32 / int64 / uint16 small ones, 1-d and 2-d
            u = rng.integers(0, 255, size=(Y, X)).astype(np.uint8)
            var("mask_u8", u, chunks=(min(4, Y), min(5, X)))
            exp["mask_u8"] = u.astype(np.float64)
            i4 = rng.integers(-2**31, 2**31 - 1, size=(T,), dtype=np.int64).astype(np.int32)
            var("count_i32", i4, scales=(t,) * 3)
            exp["count_i32"] = i4.astype(np.float64)
            u2 = rng.integers(0, 65535, size=(T, Y, X)).astype(np.uint16)
            var("u16cube", u2, chunks=chunks, shuffle=True)
            exp["u16cube"] = u2.astype(np.float64)
            # compact storage is not reachable through h5py's high-level API; contiguous big-endian int16
            be = rng.integers(-3000, 3000, size=(Y, X)).astype(">i2")
            var("be_i16", be)
            exp["be_i16"] = be.astype(np.float64)
        if big_attrs:
            v = f["temperature"]
            for k in range(12):  # > 8 attributes: dense attribute storage in new-style files
                v.attrs[f"extra_{k:02d}"] = np.float64(k) * 1.25
            v.attrs["ints"] = np.arange(5, dtype=np.int32)
        for k in range(n_extra):  # many links: dense link storage, deeper v2 B-trees
            d = f.create_dataset(f"aux_{k:04d}", data=np.float32(k) + np.arange(3, dtype=np.float32), track_order=track)
            exp[f"aux_{k:04d}"] = d[...].astype(np.float64)
    np.savez_compressed(os.path.splitext(path)[0] + ".npz", **exp)
    return exp


def _write_chunks_direct(v, a, chunks, level, threads, t0=0):
    """Same bytes libhdf5's own pipeline would store (shuffle, then zlib at `level`; edge chunks padded with the fill value
    0), produced on `threads` threads - zlib releases the GIL - and handed over with write_direct_chunk."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    es = a.dtype.itemsize
    origins = [(t, y, x) for t in range(0, a.shape[0], chunks[0]) for y in range(0, a.shape[1], chunks[1])
               for x in range(0, a.shape[2], chunks[2])]

    def pack(o):
        part = a[o[0]:o[0] + chunks[0], o[1]:o[1] + chunks[1], o[2]:o[2] + chunks[2]]
        if part.shape == tuple(chunks):
            blk = np.ascontiguousarray(part)
        else:
            blk = np.zeros(chunks, dtype=a.dtype)
            blk[:part.shape[0], :part.shape[1], :part.shape[2]] = part
        shuffled = blk.view(np.uint8).reshape(-1, es).T.copy() if es > 1 else blk.view(np.uint8)
        return zlib.compress(shuffled.tobytes(), level)

    with ThreadPoolExecutor(threads) as ex:
        for o, payload in zip(origins, ex.map(pack, origins)):
            v.id.write_direct_chunk((o[0] + t0, o[1], o[2]), payload)


PV_VARS = ("influx_toa", "influx_direct", "influx_diffuse", "albedo", "temperature", "solar_altitude", "solar_azimuth")


def write_cutout(path, T=48, Y=9, X=12, chunks=(20, 5, 7), dtype="f4", seed=7, gzip=6, start_hours=990552, threads=1, only=None):
    """An ERA5-shaped cutout with every input of pv / wind / heat_demand / runoff (random but in range).  ``only``: write just
    these cubes (the bench's year-long file holds the seven pv inputs; the random draws stay those of the full set)."""
    rng = np.random.default_rng(seed)
    with h5py.File(path, "w", libver=("earliest", "v108"), track_order=True) as f:
        f.attrs["module"] = np.string_("era5")
        f.attrs["prepared_features"] = np.string_("height,wind,influx,temperature,runoff")
        t = f.create_dataset("time", data=(np.arange(T) + start_hours).astype(np.int32), track_order=True)
        t.attrs["units"] = np.string_("hours since 1900-01-01 00:00:00.0")
        t.attrs["calendar"] = np.string_("proleptic_gregorian")
        y = f.create_dataset("y", data=35.0 + 0.25 * np.arange(Y), track_order=True)
        x = f.create_dataset("x", data=-5.0 + 0.25 * np.arange(X), track_order=True)
        for d, n in ((t, "time"), (y, "y"), (x, "x")):
            d.make_scale(n)
        f.create_dataset("lon", data=x[...], track_order=True).dims[0].attach_scale(x)
        f.create_dataset("lat", data=y[...], track_order=True).dims[0].attach_scale(y)
        def make_fields(rng, Tb):
            u = lambda: rng.random((Tb, Y, X))
            alt = (u() - 0.35) * 1.6
            toa = 1361.0 * np.maximum(np.sin(alt), 0.0)
            kt, fd = 0.2 + 0.55 * u(), 0.3 + 0.5 * u()
            return {
                "influx_toa": toa, "influx_direct": toa * kt * fd, "influx_diffuse": toa * kt * (1 - fd),
                "albedo": 0.05 + 0.3 * u(), "temperature": 268.0 + 30.0 * u(), "solar_altitude": alt,
                "solar_azimuth": 2 * np.pi * u(), "wnd100m": 25.0 * u() ** 2, "roughness": 0.001 + 1.5 * u() ** 3,
                "runoff": 1e-4 * u(), "soil temperature": 270.0 + 20.0 * u(),
            }

        def finish(v, n):
            for i, s in enumerate((t, y, x)):
                v.dims[i].attach_scale(s)
            v.attrs["units"] = np.string_("unit of " + n)

        if threads > 1:
            # big bench files: generated and written in blocks of whole chunk rows (memory stays bounded: a year of 200 x 200
            # would be 30 GB of float64 fields at once), the chunks shuffled + deflated on a thread pool and written as they are
            tb = max(chunks[0], (720 // chunks[0]) * chunks[0])
            dsets = {}
            for t0 in range(0, T, tb):
                blk = make_fields(np.random.default_rng([seed, t0]), min(tb, T - t0))
                for n, a in blk.items():
                    if only is not None and n not in only:
                        continue
                    if n not in dsets:
                        dsets[n] = f.create_dataset(n, shape=(T, Y, X), dtype=dtype, chunks=chunks, compression="gzip",
                                                    compression_opts=gzip, shuffle=True, track_order=True)
                    _write_chunks_direct(dsets[n], (np.round(a * 4096) / 4096).astype(dtype), chunks, gzip, threads, t0)
            for n, v in dsets.items():
                finish(v, n)
        else:
            for n, a in make_fields(rng, T).items():
                if only is not None and n not in only:
                    continue
                a = np.round(a * 4096) / 4096  # keeps the deflated fixture small
                v = f.create_dataset(n, data=a.astype(dtype), chunks=chunks, compression="gzip", compression_opts=gzip,
                                     shuffle=True, track_order=True)
                finish(v, n)
        h = f.create_dataset("height", data=(2000.0 * rng.random((Y, X))).astype(dtype), track_order=True)
        h.dims[0].attach_scale(y)
        h.dims[1].attach_scale(x)


def write_payloads(path, seed=0, T=64, Y=128, X=128, ct=8):
    """uint8 variables whose chunk streams exercise a DEFLATE decoder beyond weather fields: no matches at all, short close
    matches, runs (matches that overlap themselves), long periods (maximum-length matches), sparse bytes in zeros, far
    matches of a dictionary, byte planes of smooth integers; zlib levels 1 / 6 / 9.  Expected values beside it (.npz)."""
    rng = np.random.default_rng(seed)
    n = T * Y * X

    def runs():
        ln = rng.geometric(1.0 / rng.choice([3, 20, 400]), size=n // 2 + 1)
        return np.repeat(rng.integers(0, 256, ln.size, dtype=np.uint8), ln)[:n]

    def periodic():
        out = np.empty(n, np.uint8)
        i = 0
        while i < n:
            period = int(rng.choice([1, 2, 3, 5, 17, 100, 257, 300, 4000]))
            m = min(int(rng.integers(period, 40 * period + 600)), n - i)
            out[i:i + m] = np.resize(rng.integers(0, 256, period, dtype=np.uint8), m)
            i += m
        return out

    def sparse():
        out = np.zeros(n, np.uint8)
        idx = rng.integers(0, n, n // 700)
        out[idx] = rng.integers(1, 256, idx.size, dtype=np.uint8)
        return out

    def words():
        vocab = [rng.integers(97, 123, int(k), dtype=np.uint8) for k in rng.integers(2, 14, 900)]
        pick = rng.zipf(1.3, size=n // 4) % len(vocab)
        return np.concatenate([vocab[i] for i in pick])[:n] if n else np.zeros(0, np.uint8)

    def planes():
        v = (np.cumsum(rng.standard_normal(n // 4 + 1)) * 50).astype("<i4")
        return np.ascontiguousarray(v.view(np.uint8).reshape(-1, 4).T).reshape(-1)[:n]

    kinds = {"noise": lambda: rng.integers(0, 256, n, dtype=np.uint8), "few": lambda: rng.integers(0, 4, n, dtype=np.uint8),
             "runs": runs, "periodic": periodic, "sparse": sparse, "words": words, "planes": planes}
    exp = {}
    with h5py.File(path, "w", libver=("earliest", "v108")) as f:
        t, y, x = _scales(f, T, Y, X, True)
        for k, (name, make) in enumerate(kinds.items()):
            a = make()
            a = np.resize(a, n).reshape(T, Y, X)
            for level in ((1, 6, 9) if name in ("runs", "periodic", "words") else ((1, 6, 9)[k % 3],)):
                v = f.create_dataset(f"{name}_{level}", data=a, chunks=(ct, Y, X), compression="gzip", compression_opts=level)
                _attach(v, (t, y, x))
                exp[f"{name}_{level}"] = a
    np.savez_compressed(os.path.splitext(path)[0] + ".npz", **exp)


def main(out):
    os.makedirs(out, exist_ok=True)
    # 1. what netCDF-C produces: v108 (1.8) bounds, creation order tracked
    write_case(f"{out}/cutout_nc4.nc", seed=1)
    # 2. classic HDF5 defaults: symbol-table groups, v1 object headers
    write_case(f"{out}/cutout_earliest.nc", libver="earliest", track=False, seed=2)
    # 3. many variables: dense link storage with a two-level name index
    write_case(f"{out}/cutout_many.nc", T=5, Y=3, X=4, chunks=(2, 2, 3), seed=3, n_extra=160, variants=False)
    # 4. libver latest: v4 layouts (fixed array / implicit / single chunk index)
    write_case(f"{out}/cutout_latest.nc", libver="latest", seed=4)
    # 5. latest + unlimited time: extensible-array chunk index (refused until round 3)
    with h5py.File(f"{out}/unlimited_latest.nc", "w", libver="latest") as f:
        f.create_dataset("influx", data=np.ones((6, 4, 5), "f4"), chunks=(2, 4, 5), maxshape=(None, 4, 5))
        f.create_dataset("y", data=np.arange(4.0))
    #    ... every variable flavour with an unlimited time axis (one time step per chunk along time, as netCDF-C chunks
    #    a record dimension), with an unlimited middle axis (the array index is "swizzled"), with two unlimited axes (v2 B-tree)
    write_case(f"{out}/cutout_unlimited.nc", T=23, Y=7, X=9, chunks=(1, 4, 5), libver="latest", seed=11, unlimited=(0,), big_attrs=False)
    write_case(f"{out}/cutout_unlimited_y.nc", T=9, Y=7, X=9, chunks=(4, 2, 5), libver="latest", seed=12, unlimited=(1,), big_attrs=False)
    write_case(f"{out}/cutout_unlimited_ty.nc", T=9, Y=7, X=9, chunks=(4, 3, 5), libver="latest", seed=13, unlimited=(0, 1), big_attrs=False)
    # 6. 4-byte offsets are not reachable from h5py; a user block shifts the superblock instead
    with h5py.File(f"{out}/userblock.nc", "w", userblock_size=512, libver=("earliest", "v108")) as f:
        f.create_dataset("temperature", data=np.arange(24, dtype="f4").reshape(2, 3, 4), chunks=(1, 3, 4),
                         compression="gzip", shuffle=True)
    np.savez_compressed(f"{out}/userblock.npz", temperature=np.arange(24, dtype="f8").reshape(2, 3, 4))
    # 7. complete small cutouts for the end-to-end tests (float32 as atlite writes them; float64)
    write_cutout(f"{out}/cutout_small_f32.nc")
    write_cutout(f"{out}/cutout_small_f64.nc", dtype="f8", T=30, chunks=(7, 9, 5), seed=8)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--cutout":  # --cutout path T Y X ct cy cx dtype seed [threads ["pv"]]
        a = sys.argv[2:]
        write_cutout(a[0], int(a[1]), int(a[2]), int(a[3]), (int(a[4]), int(a[5]), int(a[6])), a[7], int(a[8]), gzip=int(os.environ.get("ATL_FIXTURE_GZIP", "1")),
                     threads=int(a[9]) if len(a) > 9 else 1, only=PV_VARS if len(a) > 10 and a[10] == "pv" else None)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--case":  # --case path T Y X ct cy cx libver track seed [unlimited axes, e.g. 0 or 01]
        a = sys.argv[2:]
        lv = a[7] if a[7] in ("earliest", "latest") else ("earliest", "v108")
        unl = tuple(int(c) for c in a[10]) if len(a) > 10 else ()
        write_case(a[0], int(a[1]), int(a[2]), int(a[3]), (int(a[4]), int(a[5]), int(a[6])), lv, a[8] == "1", int(a[9]), unlimited=unl)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--payloads":  # --payloads path seed [T Y X ct]
        a = sys.argv[2:]
        write_payloads(a[0], int(a[1]) if len(a) > 1 else 0, *[int(v) for v in a[2:6]])
        sys.exit(0)
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "nc"))
