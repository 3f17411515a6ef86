"""
ctypes binding of ``libatlite_hip.so`` (C ABI: ``include/atlite_hip.h``).

The library is the product path; there is no CPU fallback.  If the shared object is missing
or a GPU call fails, an exception is raised - nothing silently reroutes to NumPy.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("ATLITE_HIP_LIB", _HERE / "lib" / "libatlite_hip.so"))

ATL_OK, ATL_E_INVALID, ATL_E_HIP, ATL_E_NOMEM, ATL_E_UNSUPPORTED = 0, -1, -2, -3, -4
TIME_NONE, TIME_SUM, TIME_MEAN, TIME_SUM_COUNT = 0, 1, 2, 3
WIND_NONE, WIND_LOG, WIND_POWER = 0, 1, 2
SYN_UNIFORM, SYN_RAYLEIGH, SYN_EXPLOG, SYN_NEGLOG = 0, 1, 2, 3

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)


class PvInputs(C.Structure):
    _fields_ = [
        (n, C.c_void_p)
        for n in (
            "d_influx_direct",
            "d_influx_diffuse",
            "d_influx_toa",
            "d_albedo",
            "d_temperature",
            "d_solar_altitude",
            "d_solar_azimuth",
            "d_sin_dec",
            "d_cos_dec",
            "d_hour_angle",
            "d_cos_hour_angle",
            "d_sin_lat",
            "d_cos_lat",
        )
    ] + [("X", C.c_int64)] + [(n, C.c_void_p) for n in ("d_influx", "d_outflux", "d_humidity", "d_day_map")] + [("day_map_ld", C.c_int64)]


class PvParams(C.Structure):
    _fields_ = [
        ("c_temp_amb", C.c_double),
        ("c_temp_irrad", C.c_double),
        ("r_tmod", C.c_double),
        ("r_irradiance", C.c_double),
        ("k_1", C.c_double),
        ("k_2", C.c_double),
        ("k_3", C.c_double),
        ("k_4", C.c_double),
        ("k_5", C.c_double),
        ("k_6", C.c_double),
        ("inverter_efficiency", C.c_double),
        ("slope", C.c_double),
        ("azimuth", C.c_double),
        ("d_cell_slope", C.c_void_p),
        ("d_cell_azimuth", C.c_void_p),
        ("altitude_threshold", C.c_double),
        ("tracking", C.c_int),
        ("trigon_model", C.c_int),
        ("clearsky_model", C.c_int),
        ("irradiation", C.c_int),
        ("panel_model", C.c_int),
    ] + [(n, C.c_double) for n in ("bof_A", "bof_B", "bof_C", "bof_D", "bof_NOCT", "bof_Tstd", "bof_Tamb", "bof_Intc",
                                   "bof_ta", "bof_threshold", "st_c0", "st_c1", "st_t_store_K")] + [("night_skip", C.c_int), ("orientation_per_time", C.c_int)]


TRACKING = {None: 0, "horizontal": 1, "tilted_horizontal": 2, "vertical": 3, "dual": 4}
TRIGON = {"simple": 0, "other": 1}
CLEARSKY = {"simple": 0, "enhanced": 1}
IRRADIATION = {"total": 0, "direct": 1, "diffuse": 2, "ground": 3}
PANEL = {"huld": 0, "bofinger": 1, "none": 2, "solar_thermal": 3}


class WindInputs(C.Structure):
    _fields_ = [("d_wnd", C.c_void_p), ("d_aux", C.c_void_p), ("aux_is_static", C.c_int)]


class WindParams(C.Structure):
    _fields_ = [
        ("method", C.c_int),
        ("to_height", C.c_double),
        ("from_height", C.c_double),
        ("n_knots", C.c_int),
        ("h_V", c_double_p),
        ("h_POWn", c_double_p),
    ]


class HeatParams(C.Structure):
    _fields_ = [
        ("threshold_K", C.c_double),
        ("a", C.c_double),
        ("constant", C.c_double),
        ("n_days", C.c_int64),
        ("d_day_ptr", C.c_void_p),
        ("cooling", C.c_int),
    ]


class ThermoParams(C.Structure):
    _fields_ = [
        ("offset", C.c_double),
        ("fillna0", C.c_int),
        ("quadratic", C.c_int),
        ("sink_T", C.c_double),
        ("c0", C.c_double),
        ("c1", C.c_double),
        ("c2", C.c_double),
    ]


class SynthSolar(C.Structure):
    _fields_ = [
        ("d_sin_dec", C.c_void_p),
        ("d_cos_dec", C.c_void_p),
        ("d_h", C.c_void_p),
        ("d_lat_rad", C.c_void_p),
        ("d_tseason", C.c_void_p),
        ("X", C.c_int64),
        ("Y", C.c_int64),
        ("seed", C.c_uint64),
        ("ld_cells", C.c_int64),
    ]


# name -> (restype, argtypes); must list every symbol include/atlite_hip.h declares
_vp, _i, _i64, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double
class NcVar(C.Structure):
    """atl_nc_var (include/atlite_hip.h)."""

    _fields_ = (
        [(n, C.c_int32) for n in ("ndim", "dtype", "elem_size", "big_endian")]
        + [("shape", C.c_int64 * 4), ("chunk", C.c_int64 * 4)]
        + [(n, C.c_int32) for n in ("layout", "shuffle", "deflate", "fletcher32", "has_scale", "has_fill",
                                    "has_missing", "reserved_")]
        + [(n, C.c_double) for n in ("scale_factor", "add_offset", "fill_value", "missing_value")]
        + [("n_chunks", C.c_int64), ("stored_bytes", C.c_int64)]
    )


NC_DTYPES = {1: "float32", 2: "float64", 3: "int8", 4: "int16", 5: "int32", 6: "int64", 7: "uint8", 8: "uint16",
             9: "uint32", 10: "uint64"}
NC_CODES = {v: k for k, v in NC_DTYPES.items()}

SIGNATURES = {
    "atl_version": (_i, []),
    "atl_last_error": (C.c_char_p, []),
    "atl_device_count": (_i, [C.POINTER(_i)]),
    "atl_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "atl_destroy": (_i, [_vp]),
    "atl_sync": (_i, [_vp]),
    "atl_device_name": (_i, [_vp, C.c_char_p, _sz]),
    "atl_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "atl_free": (_i, [_vp, _vp]),
    "atl_upload": (_i, [_vp, _vp, _vp, _sz]),
    "atl_download": (_i, [_vp, _vp, _vp, _sz]),
    "atl_memset": (_i, [_vp, _vp, _i, _sz]),
    "atl_pinned_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "atl_pinned_free": (_i, [_vp]),
    "atl_host_register": (_i, [_vp, _sz]),
    "atl_host_unregister": (_i, [_vp]),
    "atl_upload_async": (_i, [_vp, _vp, _vp, _sz]),
    "atl_event_create": (_i, [_vp, C.POINTER(_vp)]),
    "atl_event_destroy": (_i, [_vp]),
    "atl_event_record": (_i, [_vp, _vp, _i]),
    "atl_stream_wait_event": (_i, [_vp, _i, _vp]),
    "atl_event_synchronize": (_i, [_vp]),
    "atl_timer_start": (_i, [_vp]),
    "atl_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "atl_set_profiling": (_i, [_vp, _i]),
    "atl_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "atl_kernel_times": (_i, [_vp, C.POINTER(C.c_float), _i64, C.POINTER(_i64)]),
    "atl_agg_create": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "atl_agg_create_aligned": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "atl_agg_destroy": (_i, [_vp]),
    "atl_agg_selfcheck": (_i, [_i64, _i64, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_agg_selfcheck_aligned": (_i, [_i64, _i64, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_agg_info": (_i, [_vp, c_int64_p, c_int64_p, c_int64_p, c_int64_p, c_int32_p, c_int32_p]),
    "atl_spmm_csr": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _vp, _i64]),
    "atl_pv_convert": (_i, [_vp, C.POINTER(PvInputs), C.POINTER(PvParams), _i64, _i64, _i, _vp]),
    "atl_pv_convert_aggregate": (
        _i,
        [_vp, C.POINTER(PvInputs), C.POINTER(PvParams), _i64, _i64, _vp, _i, _vp, _i64],
    ),
    "atl_wind_convert": (
        _i,
        [_vp, C.POINTER(WindInputs), C.POINTER(WindParams), _i64, _i64, _i, _vp],
    ),
    "atl_wind_convert_aggregate": (
        _i,
        [_vp, C.POINTER(WindInputs), C.POINTER(WindParams), _i64, _i64, _vp, _i, _vp, _i64],
    ),
    "atl_heat_demand_convert": (_i, [_vp, _vp, C.POINTER(HeatParams), _i64, _i64, _i, _vp]),
    "atl_heat_demand_convert_aggregate": (
        _i,
        [_vp, _vp, C.POINTER(HeatParams), _i64, _i64, _vp, _i, _vp, _i64],
    ),
    "atl_thermo_convert": (_i, [_vp, _vp, C.POINTER(ThermoParams), _i64, _i64, _i, _vp]),
    "atl_thermo_convert_aggregate": (_i, [_vp, _vp, C.POINTER(ThermoParams), _i64, _i64, _vp, _i, _vp, _i64]),
    "atl_runoff_convert": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _vp]),
    "atl_runoff_convert_aggregate": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _i, _vp, _i64]),
    "atl_indicator_polygons": (
        _i,
        [_i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, C.POINTER(_vp), C.POINTER(_vp),
         C.POINTER(_vp)],
    ),
    "atl_agg_check_host": (_i, [_i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_agg_check_host_aligned": (_i, [_i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_indicator_polygons_integral_host": (
        _i,
        [_i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    ),
    "atl_indicator_polygons_device": (
        _i,
        [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _d, _d, _d, _d, C.POINTER(_vp), C.POINTER(_vp),
         C.POINTER(_vp)],
    ),
    "atl_host_free": (_i, [_vp]),
    "atl_nc_open": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "atl_nc_close": (_i, [_vp]),
    "atl_nc_list": (_i, [_vp, C.c_char_p, _i64, C.POINTER(_i64)]),
    "atl_nc_inquire": (_i, [_vp, C.c_char_p, C.POINTER(NcVar)]),
    "atl_nc_dims": (_i, [_vp, C.c_char_p, C.c_char_p, _i64, C.POINTER(_i64)]),
    "atl_nc_att_text": (_i, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, _i64, C.POINTER(_i64)]),
    "atl_nc_att_double": (_i, [_vp, C.c_char_p, C.c_char_p, _vp, _i64, C.POINTER(_i64)]),
    "atl_nc_read_host": (_i, [_vp, C.c_char_p, _i64, _i64, _vp]),
    "atl_nc_read_slab": (_i, [_vp, _vp, C.c_char_p, _i64, _i64, _vp, _i]),
    "atl_nc_read_slabs": (_i, [_vp, _vp, _i, C.POINTER(C.c_char_p), _i64, _i64, C.POINTER(_vp), _i]),
    "atl_nc_read_slabs_ld": (_i, [_vp, _i64, _vp, _i, C.POINTER(C.c_char_p), _i64, _i64, C.POINTER(_vp), _i]),
    "atl_nc_ingest_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_pv_day_map": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64]),
    "atl_nc_ingest_times": (_i, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "atl_upload_convert_async": (_i, [_vp, _vp, _vp, _i, _i64]),
    "atl_upload_convert_2d_async": (_i, [_vp, _vp, _i64, _vp, _i, _i64, _i64]),
    "atl_inflate_probe": (_i, [_vp, _sz, _vp, _sz, _i, C.POINTER(_i64)]),
    "atl_comm_unique_id": (_i, [_vp]),
    "atl_comm_init": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "atl_comm_destroy": (_i, [_vp]),
    "atl_allgather_time": (_i, [_vp, _vp, _i64, _i64, _vp, _i64]),
    "atl_allreduce_sum": (_i, [_vp, _vp, _i64]),
    "atl_allgather_time_v": (_i, [_vp, _vp, _i64, c_int64_p, _vp, _i64]),
    "atl_allgather_time_v_async": (_i, [_vp, _vp, _i64, c_int64_p, _vp, _i64, C.POINTER(_i64)]),
    "atl_comm_wait": (_i, [_vp, _i64]),
    "atl_comm_sync": (_i, [_vp]),
    "atl_comm_init_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "atl_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "atl_capture_begin": (_i, [_vp]),
    "atl_capture_end": (_i, [_vp, C.POINTER(_vp)]),
    "atl_graph_launch": (_i, [_vp, _vp]),
    "atl_graph_destroy": (_i, [_vp]),
    "atl_rolling_mean": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64]),
    "atl_order_statistic": (_i, [_vp, _vp, _i64, _i64, _i64, _d, c_int64_p, c_double_p, c_int64_p, c_int64_p]),
    "atl_zero_below": (_i, [_vp, _vp, _i64, _i64, _i64, _d]),
    "atl_normalize_rows": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "atl_copy_2d": (_i, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _i, _i]),
    "atl_comm_group_create": (_i, [_i, C.POINTER(_vp)]),
    "atl_comm_group_destroy": (_i, [_vp]),
    "atl_comm_init_local": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "atl_comm_abort": (_i, [_vp]),
    "atl_gather_place_v_host": (_i, [_vp, _i, _i64, c_int64_p, _vp, _i64]),
    "atl_math_probe": (_i, [_vp, _i, _vp, _i64, _vp]),
    "atl_math_probe_host": (_i, [_i, _vp, _i64, _vp]),
    "atl_wind_probe_host": (_i, [C.POINTER(WindParams), _i64, _vp, _vp, _vp]),
    "atl_pv_probe_host": (_i, [C.POINTER(PvParams), _i, _i64, C.POINTER(_vp), _vp]),
    "atl_wind_interp_host": (_i, [_vp, _vp, _i, _vp, _i64, _vp]),
    "atl_synth_field": (
        _i,
        [_vp, _i, C.c_uint64, C.c_uint64, _d, _d, _i, _i64, _i64, _vp],
    ),
    "atl_synth_pv_inputs": (
        _i,
        [_vp, C.POINTER(SynthSolar), _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    ),
}

# the twins that take the slot stride as their second argument (include/atlite_hip.h: "the slot stride as an ARGUMENT")
for _name in ("atl_agg_create", "atl_spmm_csr", "atl_pv_convert", "atl_pv_convert_aggregate", "atl_pv_day_map", "atl_wind_convert",
              "atl_wind_convert_aggregate", "atl_heat_demand_convert", "atl_heat_demand_convert_aggregate", "atl_thermo_convert",
              "atl_thermo_convert_aggregate", "atl_runoff_convert", "atl_runoff_convert_aggregate", "atl_nc_read_slab"):
    _res, _args = SIGNATURES[_name]
    SIGNATURES[_name + "_ld"] = (_res, [_args[0], _i64] + list(_args[1:]))
del _name, _res, _args

_lib = None


class AtliteHipError(RuntimeError):
    """A HIP runtime failure reported by libatlite_hip.so."""


def load():
    """Load (once) and return the ctypes handle; raise loudly if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C atlite_amd/csrc`. There is no CPU fallback."
        )
    if os.environ.get("ATLITE_HIP_NO_TORCH", "0") != "1":
        # torch bundles its own libamdhip64.so.7; importing it first makes this library bind
        # to the same HIP runtime, so device pointers can be shared with torch tensors.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    """Map an ATL_E_* return code to the Python exception class the reference would raise."""
    if rc == ATL_OK:
        return
    msg = load().atl_last_error().decode("utf-8", "replace")
    if rc == ATL_E_INVALID:
        raise ValueError(msg)
    if rc == ATL_E_NOMEM:
        raise MemoryError(msg)
    if rc == ATL_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise AtliteHipError(msg)
