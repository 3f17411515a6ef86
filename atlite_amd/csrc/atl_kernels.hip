// HIP kernels (gfx950 / CDNA4, wave64) for atlite's convert_and_aggregate hot path and the
// C-ABI launchers declared in include/atlite_hip.h.
//
// One streaming pass over the (time, cell) fp64 cubes.  Three kernel shapes, each a template
// over a "converter" that turns the input variables of one (slot, cell) into one fp64 value:
//   k_cells_series   out[slot, cell]               (aggregate_time=None, no matrix)
//   k_cells_timered  out[cell] = sum_t / mean_t     (no matrix, aggregate_time sum/mean)
//   k_fused_segred   partial[prow, slot]            (matrix / shapes / layout given)
// k_fused_segred never materialises the converted cube: a wave owns 128 consecutive cells
// (2 per lane, 16-byte loads) and walks a chunk of output slots; per batch of 8 slots the
// per-lane values are multiplied by the segment-local indicator weights and reduced across
// the wave by a 3+3 stage shuffle butterfly that ends with lane 8g holding slot g, so the
// partial row is written with one 64-byte store.  k_combine then sums, in a fixed order, the
// partial rows of each shape: deterministic, no atomics.
//
// Reference arithmetic (file:line under /root/reference/atlite):
//   pv   : convert.py:840-854, pv/orientation.py:114-117,188, pv/irradiation.py:196-226,
//          247-255, pv/solar_panel_model.py:12-44
//   wind : convert.py:634-662 (np.interp), wind.py:76-112
//   heat : convert.py:405-418      runoff : convert.py:1028-1034
//   agg  : aggregate.py:16-35 (scipy CSR product), convert.py:51-56 (_aggregate_time)
#include "atl_kernel_templates.h"

namespace {

#include "atl_conv_wind.h"

// ---------------------------------------------------------------------------------------
// synthetic fields
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double hash_u01(uint64_t seed, uint64_t var, uint64_t idx) {
    uint64_t z = (seed ^ (var * 0xD1B54A32D192ED03ull)) + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return double(z >> 11) * 0x1.0p-53;
}

__global__ __launch_bounds__(256) void k_synth_field(int kind, uint64_t seed, uint64_t var, double p0,
                                                     double p1, int per_cell_static, int64_t T, int64_t S,
                                                     double *__restrict__ out) {
    const int64_t n = T * S;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const uint64_t idx = per_cell_static ? uint64_t(i % S) : uint64_t(i);
        const double u = hash_u01(seed, var, idx);
        double r;
        switch (kind) {
            case ATL_SYN_UNIFORM: r = p0 + (p1 - p0) * u; break;
            case ATL_SYN_RAYLEIGH: r = p0 * sqrt(-log1p(-u)) * 1.1283791670955126; break;
            case ATL_SYN_EXPLOG: r = exp(log(p0) + u * log(p1)); break;
            default: r = -p0 * log1p(-u); break;
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(256) void k_synth_pv(atl_synth_solar s, int64_t T, int64_t S, int64_t ld, double *dir,
                                                  double *dif, double *toa, double *alb, double *tmp,
                                                  double *altp, double *azp) {
    const int64_t n = T * S;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const int64_t t = i / S, c = i % S;
        const int64_t o = t * ld + c;  // slots ld cells apart (ld_cells); the hash stays on the logical index
        const int64_t y = c / s.X, x = c % s.X;
        const double lat = s.d_lat_rad[y];
        const double sl = sin(lat), cl = cos(lat);
        const double sd = s.d_sin_dec[t], cd = s.d_cos_dec[t];
        const double h = s.d_h[t * s.X + x];
        const double ch = cos(h);
        // pv/solar_position.py:100-114
        double sa = sd * sl + cd * cl * ch;
        sa = fmin(fmax(sa, -1.0), 1.0);
        const double alt = asin(sa);
        double caz = (sd * cl - cd * sl * ch) / cos(alt);
        caz = fmin(fmax(caz, -1.0), 1.0);
        double az = acos(caz);
        if (!(h <= 0.0)) az = 2.0 * M_PI - az;
        const double u1 = hash_u01(s.seed, 1, i), u2 = hash_u01(s.seed, 2, i);
        const double u3 = hash_u01(s.seed, 3, i), u4 = hash_u01(s.seed, 4, i);
        const double top = 1361.0 * fmax(sa, 0.0);
        const double kt = 0.2 + 0.55 * u1, fd = 0.3 + 0.5 * u2;
        altp[o] = alt;
        azp[o] = az;
        toa[o] = top;
        dir[o] = top * kt * fd;
        dif[o] = top * kt * (1.0 - fd);
        alb[o] = 0.05 + 0.30 * u3;
        tmp[o] = s.d_tseason[t] - 0.4 * (lat * (180.0 / M_PI) - 50.0) + 4.0 * (u4 - 0.5);
    }
}

// ---------------------------------------------------------------------------------------
// math probe (accuracy tests of atl_math.h against numpy)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_math_probe(int fn, const double *__restrict__ in, int64_t n,
                                                    double *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) double ltab[2 * kLogTabN];
    log_table_init(ltab);
    __syncthreads();
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double r;
    switch (fn) {
        case 0: r = lean_sin(x); break;
        case 1: r = lean_cos(x); break;
        case 2: r = lean_log(x); break;
        case 3: {
            double s, c;
            lean_sincos(x, &s, &c);
            r = s;
            out[n + i] = c;
            break;
        }
        case 5: r = log_core_tab(x, ltab); break;  // positive normal finite arguments only
        case 7: r = lean_sqrt(x); break;
        case 8: r = lean_sqrt_rsqrt(x, &out[n + i]); break;
        default: r = fast_div(x, in[n + i]); break;
    }
    out[i] = r;
}

int make_wind(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T, int64_t S,
              WindConvT<-1> *c, bool *vec, size_t *lds_bytes, bool *table_finite) {
    ATL_REQUIRE(in && p, "atl_wind: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_wind: negative shape");
    ATL_REQUIRE(in->d_wnd, "atl_wind: wind speed is NULL");
    ATL_REQUIRE(p->method == ATL_WIND_NONE || p->method == ATL_WIND_LOG || p->method == ATL_WIND_POWER,
                "Interpolation method must be 'logarithmic' or 'power' (got code %d)", p->method);
    ATL_REQUIRE(p->method == ATL_WIND_NONE || in->d_aux,
                "atl_wind: method needs roughness / wnd_shear_exp (wind.py:94-98,106-110)");
    if (p->n_knots == 0) {  // no power curve: the extrapolated wind speed itself (wind.py:76-112)
        ATL_REQUIRE(p->method == ATL_WIND_NONE || (p->to_height > 0 && p->from_height > 0 && std::isfinite(p->to_height) &&
                                                    std::isfinite(p->from_height)),
                    "atl_wind: heights must be positive and finite");
        c->wnd = in->d_wnd;
        c->aux = in->d_aux;
        c->S = slot_stride_of(ctx, S);  // the converter's S is what separates the slots of its cubes
        c->aux_static = in->aux_is_static;
        c->method = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;
        c->to_height = p->to_height;
        c->from_height = p->from_height;
        c->log_ratio = log(p->to_height / p->from_height);
        c->table = nullptr;
        c->n_knots = c->n_pad = c->tab_doubles = c->b0 = 0;
        c->vmin = c->vmax = c->inv_w = 0.0;
        *table_finite = true;
        *lds_bytes = size_t(2 * kLogTabN) * sizeof(double);
        *vec = vec_ok(T, S, slot_stride_of(ctx, S), {in->d_wnd, in->aux_is_static ? nullptr : in->d_aux});
        return ATL_OK;
    }
    ATL_REQUIRE(p->n_knots >= 1 && p->n_knots <= kMaxKnots && p->h_V && p->h_POWn,
                "atl_wind: power curve needs 1..%d knots", kMaxKnots);
    std::vector<double> tbl;
    bool finite = true;
    int n_pad = 0;
    const int n = wind_table_build(p->h_V, p->h_POWn, p->n_knots, tbl, &n_pad, &finite);
    ATL_REQUIRE(n > 0, "wind speed 'V' in the turbine config is expected to be increasing");
    ATL_REQUIRE(n <= kMaxKnots, "atl_wind: power curve needs 1..%d knots", kMaxKnots);
    // grid-aligned knots (the usual case): the bucket table replaces the search table
    std::vector<double> grid;
    double inv_w = 0.0;
    int b0 = 0;
    const int n_grid = (finite && !getenv("ATLITE_HIP_WIND_NO_GRID")) ? wind_grid_build(tbl.data(), n, n_pad, grid, &inv_w, &b0) : 0;
    const std::vector<double> tbl_search = tbl;  // vmin / vmax below
    if (n_grid > 0) tbl = grid;
    // pinned staging buffer: wait until the previous call's copy has left it, then enqueue the H2D
    // stream-ordered after any earlier kernel that still reads the device table
    if (ctx->table_pending) ATL_HIP_TRY(hipEventSynchronize(ctx->ev_table));
    memcpy(ctx->h_table, tbl.data(), tbl.size() * sizeof(double));
    ATL_HIP_TRY(hipMemcpyAsync(ctx->d_table, ctx->h_table, tbl.size() * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    ATL_HIP_TRY(hipEventRecord(ctx->ev_table, ctx->stream));
    ctx->table_pending = true;
    c->wnd = in->d_wnd;
    c->aux = in->d_aux;
    c->S = slot_stride_of(ctx, S);
    c->aux_static = in->aux_is_static;
    // (to / from) ** shear with to == from is 1 whatever the shear exponent holds (pow(1, NaN) = 1, which
    // exp(NaN * log 1) would not give): no extrapolation at all
    c->method = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;
    c->to_height = p->to_height;
    c->from_height = p->from_height;
    c->log_ratio = log(p->to_height / p->from_height);
    c->table = ctx->d_table;
    c->n_knots = n;
    c->n_pad = n_pad;
    c->tab_doubles = 5 * n_pad;
    c->vmin = tbl_search[0];
    c->vmax = tbl_search[size_t(n - 1)];
    c->inv_w = 0.0;  // 0: not grid-aligned
    c->b0 = 0;
    if (n_grid > 0) {
        c->tab_doubles = 4 * n_grid;
        c->inv_w = inv_w;
        c->b0 = b0;
    }
    *table_finite = finite;
    *lds_bytes = size_t(c->tab_doubles + 2 * kLogTabN) * sizeof(double);
    *vec = vec_ok(T, S, slot_stride_of(ctx, S), {in->d_wnd, in->aux_is_static ? nullptr : in->d_aux});
    return ATL_OK;
}

int make_heat(const atl_ctx *ctx, const double *d_temperature, const atl_heat_params *p, int64_t T, int64_t S, HeatConv *c,
              bool *vec) {
    ATL_REQUIRE(d_temperature && p, "atl_heat_demand: temperature/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0 && p->n_days >= 0, "atl_heat_demand: negative shape");
    ATL_REQUIRE(p->n_days == 0 || p->d_day_ptr, "atl_heat_demand: d_day_ptr is NULL");
    c->temperature = d_temperature;
    c->day_ptr = p->d_day_ptr;
    c->S = slot_stride_of(ctx, S);
    c->threshold_K = p->threshold_K;
    c->a = p->a;
    c->constant = p->constant;
    c->cooling = p->cooling ? 1 : 0;
    *vec = vec_ok(T, S, slot_stride_of(ctx, S), {d_temperature});
    return ATL_OK;
}

template <int M, int STEPS = 0>
WindConvT<M, STEPS> wind_as(const WindConvT<-1> &g) {
    WindConvT<M, STEPS> c;
    c.wnd = g.wnd;
    c.aux = g.aux;
    c.S = g.S;
    c.aux_static = g.aux_static;
    c.method = g.method;
    c.to_height = g.to_height;
    c.from_height = g.from_height;
    c.log_ratio = g.log_ratio;
    c.table = g.table;
    c.n_knots = g.n_knots;
    c.n_pad = g.n_pad;
    c.tab_doubles = g.tab_doubles;
    c.vmin = g.vmin;
    c.vmax = g.vmax;
    c.inv_w = g.inv_w;
    c.b0 = g.b0;
    return c;
}

// run `f(converter)` with the instantiation matching (method, table finiteness)
template <class F>
int wind_dispatch(const WindConvT<-1> &g, bool finite, F &&f) {
    // the fast log-law path also needs positive, finite heights (their logs are taken once)
    const bool heights_ok = g.to_height > 0 && g.from_height > 0 && std::isfinite(g.to_height) &&
                            std::isfinite(g.from_height);
    if (g.n_knots == 0) {  // no power curve (make_wind has checked the heights)
        switch (g.method) {
            case ATL_WIND_LOG: return f(wind_as<ATL_WIND_LOG, kWindIdentity>(g));
            case ATL_WIND_POWER: return f(wind_as<ATL_WIND_POWER, kWindIdentity>(g));
            default: return f(wind_as<ATL_WIND_NONE, kWindIdentity>(g));
        }
    }
    if (!finite || (g.method == ATL_WIND_LOG && !heights_ok)) return f(g);
    // unrolled knot search for the usual table sizes (make_wind pads to 16 / 32 / 128 knots)
    auto sized = [&](auto method) {
        constexpr int M = decltype(method)::value;
        if (g.inv_w > 0.0) return f(wind_as<M, kWindGrid>(g));  // grid-aligned knots: every shipped turbine but a few
        if constexpr (M == ATL_WIND_LOG) {  // the default law carries the unrolled searches; the others the sized loop
            if (g.n_pad == 16) return f(wind_as<M, 4>(g));
            if (g.n_pad == 32) return f(wind_as<M, 5>(g));
            if (g.n_pad == 128) return f(wind_as<M, 7>(g));
        }
        return f(wind_as<M>(g));
    };
    switch (g.method) {
        case ATL_WIND_LOG: return sized(std::integral_constant<int, ATL_WIND_LOG>());
        case ATL_WIND_POWER: return sized(std::integral_constant<int, ATL_WIND_POWER>());
        default: return sized(std::integral_constant<int, ATL_WIND_NONE>());
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI (pv: atl_kernels_pv.hip)
// ---------------------------------------------------------------------------------------
extern "C" {

int atl_spmm_csr(atl_ctx *ctx, const atl_agg *agg, const double *d_dense, int64_t T, int64_t S,
                 int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && d_dense, "atl_spmm_csr: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_spmm_csr: negative shape");
    IdentityConv c{d_dense, slot_stride_of(ctx, S)};
    return run_fused(ctx, c, vec_ok(T, S, slot_stride_of(ctx, S), {d_dense}), 0, T, S, agg, time_agg, d_out, ld_out, "atl_spmm_csr");
}

int atl_wind_convert(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T,
                     int64_t S, int time_agg, double *d_out) {
    ATL_REQUIRE(ctx, "atl_wind_convert: ctx is NULL");
    WindConvT<-1> g;
    bool vec, finite;
    size_t lds;
    int rc = make_wind(ctx, in, p, T, S, &g, &vec, &lds, &finite);
    if (rc) return rc;
    return wind_dispatch(g, finite, [&](const auto &c) {
        return run_cells(ctx, c, vec, lds, T, S, time_agg, d_out, "atl_wind_convert");
    });
}

int atl_wind_convert_aggregate(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p,
                               int64_t T, int64_t S, const atl_agg *agg, int time_agg, double *d_out,
                               int64_t ld_out) {
    ATL_REQUIRE(ctx, "atl_wind_convert_aggregate: ctx is NULL");
    WindConvT<-1> g;
    bool vec, finite;
    size_t lds;
    int rc = make_wind(ctx, in, p, T, S, &g, &vec, &lds, &finite);
    if (rc) return rc;
    return wind_dispatch(g, finite, [&](const auto &c) {
        return run_fused(ctx, c, vec, lds, T, S, agg, time_agg, d_out, ld_out, "atl_wind_convert_aggregate");
    });
}

int atl_heat_demand_convert(atl_ctx *ctx, const double *d_temperature, const atl_heat_params *p, int64_t T,
                            int64_t S, int time_agg, double *d_out) {
    ATL_REQUIRE(ctx, "atl_heat_demand_convert: ctx is NULL");
    HeatConv c;
    bool vec;
    int rc = make_heat(ctx, d_temperature, p, T, S, &c, &vec);
    if (rc) return rc;
    return run_cells(ctx, c, vec, 0, p->n_days, S, time_agg, d_out, "atl_heat_demand_convert");
}

int atl_heat_demand_convert_aggregate(atl_ctx *ctx, const double *d_temperature, const atl_heat_params *p,
                                      int64_t T, int64_t S, const atl_agg *agg, int time_agg,
                                      double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx, "atl_heat_demand_convert_aggregate: ctx is NULL");
    HeatConv c;
    bool vec;
    int rc = make_heat(ctx, d_temperature, p, T, S, &c, &vec);
    if (rc) return rc;
    return run_fused(ctx, c, vec, 0, p->n_days, S, agg, time_agg, d_out, ld_out,
                     "atl_heat_demand_convert_aggregate");
}

int atl_thermo_convert(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T, int64_t S,
                       int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && d_var && p, "atl_thermo_convert: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_thermo_convert: negative shape");
    ThermoConv c{d_var, slot_stride_of(ctx, S), p->offset, p->sink_T, p->c0, p->c1, p->c2, p->fillna0, p->quadratic};
    return run_cells(ctx, c, vec_ok(T, S, slot_stride_of(ctx, S), {d_var}), 0, T, S, time_agg, d_out, "atl_thermo_convert");
}

int atl_thermo_convert_aggregate(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T,
                                 int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && d_var && p, "atl_thermo_convert_aggregate: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_thermo_convert_aggregate: negative shape");
    ThermoConv c{d_var, slot_stride_of(ctx, S), p->offset, p->sink_T, p->c0, p->c1, p->c2, p->fillna0, p->quadratic};
    return run_fused(ctx, c, vec_ok(T, S, slot_stride_of(ctx, S), {d_var}), 0, T, S, agg, time_agg, d_out, ld_out,
                     "atl_thermo_convert_aggregate");
}

int atl_runoff_convert(atl_ctx *ctx, const double *d_runoff, const double *d_height, int64_t T, int64_t S,
                       int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && d_runoff, "atl_runoff_convert: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_runoff_convert: negative shape");
    RunoffConv c{d_runoff, d_height, slot_stride_of(ctx, S)};
    return run_cells(ctx, c, vec_ok(T, S, slot_stride_of(ctx, S), {d_runoff}), 0, T, S, time_agg, d_out, "atl_runoff_convert");
}

int atl_runoff_convert_aggregate(atl_ctx *ctx, const double *d_runoff, const double *d_height, int64_t T,
                                 int64_t S, const atl_agg *agg, int time_agg, double *d_out,
                                 int64_t ld_out) {
    ATL_REQUIRE(ctx && d_runoff, "atl_runoff_convert_aggregate: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_runoff_convert_aggregate: negative shape");
    RunoffConv c{d_runoff, d_height, slot_stride_of(ctx, S)};
    return run_fused(ctx, c, vec_ok(T, S, slot_stride_of(ctx, S), {d_runoff}), 0, T, S, agg, time_agg, d_out, ld_out,
                     "atl_runoff_convert_aggregate");
}

int atl_math_probe(atl_ctx *ctx, int fn, const double *d_in, int64_t n, double *d_out) {
    ATL_REQUIRE(ctx && d_in && d_out && n >= 0, "atl_math_probe: bad argument");
    ATL_REQUIRE((fn >= 0 && fn <= 5) || fn == 7 || fn == 8, "atl_math_probe: fn must be 0..5, 7 or 8");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return ATL_OK;
    hipLaunchKernelGGL(k_math_probe, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, fn, d_in, n,
                       d_out);
    return check_launch("atl_math_probe");
}

int atl_synth_field(atl_ctx *ctx, int kind, uint64_t seed, uint64_t var_id, double p0, double p1,
                    int per_cell_static, int64_t T, int64_t S, double *d_out) {
    ATL_REQUIRE(ctx && d_out, "atl_synth_field: bad argument");
    ATL_REQUIRE(kind >= ATL_SYN_UNIFORM && kind <= ATL_SYN_NEGLOG, "atl_synth_field: bad kind %d", kind);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (T * S == 0) return ATL_OK;
    const unsigned grid = unsigned(std::min<int64_t>((T * S + 255) / 256, int64_t(ctx->n_cu) * 32));
    hipLaunchKernelGGL(k_synth_field, dim3(grid), dim3(256), 0, ctx->stream, kind, seed, var_id, p0, p1,
                       per_cell_static, T, S, d_out);
    return check_launch("atl_synth_field");
}

int atl_synth_pv_inputs(atl_ctx *ctx, const atl_synth_solar *s, int64_t T, int64_t S,
                        double *d_influx_direct, double *d_influx_diffuse, double *d_influx_toa,
                        double *d_albedo, double *d_temperature, double *d_solar_altitude,
                        double *d_solar_azimuth) {
    ATL_REQUIRE(ctx && s, "atl_synth_pv_inputs: bad argument");
    ATL_REQUIRE(s->X > 0 && s->Y > 0 && s->X * s->Y == S, "atl_synth_pv_inputs: X*Y != S");
    ATL_REQUIRE(d_influx_direct && d_influx_diffuse && d_influx_toa && d_albedo && d_temperature &&
                    d_solar_altitude && d_solar_azimuth,
                "atl_synth_pv_inputs: NULL output");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (T * S == 0) return ATL_OK;
    const unsigned grid = unsigned(std::min<int64_t>((T * S + 255) / 256, int64_t(ctx->n_cu) * 32));
    hipLaunchKernelGGL(k_synth_pv, dim3(grid), dim3(256), 0, ctx->stream, *s, T, S, s->ld_cells > 0 ? s->ld_cells : S, d_influx_direct,
                       d_influx_diffuse, d_influx_toa, d_albedo, d_temperature, d_solar_altitude,
                       d_solar_azimuth);
    return check_launch("atl_synth_pv_inputs");
}

}  // extern "C"
