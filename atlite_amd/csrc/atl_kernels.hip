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
#include "atl_wind_make.h"

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
