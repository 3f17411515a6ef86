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
#include <cmath>
#include <cstdlib>
#include <limits>
#include <type_traits>

#include "atl_internal.h"
#include "atl_math.h"

// tuning knobs of the fused kernel (A/B-tested with tools/build_variant.sh + tools/ab_bench.sh)
#ifndef ATL_FUSED_WAVES
#define ATL_FUSED_WAVES 3  // waves per SIMD the register allocation must allow (<= 168 VGPRs)
#endif
#ifndef ATL_PV_GROUP
#define ATL_PV_GROUP 1
#endif
#ifndef ATL_ROW_CACHE
#define ATL_ROW_CACHE 3
#endif

using namespace atl;

namespace {

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool dnan(double x) { return x != x; }
__device__ __forceinline__ double fill0(double x) { return dnan(x) ? 0.0 : x; }
// numpy clip/maximum/minimum semantics: NaN in either operand propagates
__device__ __forceinline__ double np_max(double a, double b) { return (a > b || dnan(a)) ? a : b; }
__device__ __forceinline__ double np_min(double a, double b) { return (a < b || dnan(a)) ? a : b; }
__device__ __forceinline__ double np_clip(double x, double lo, double hi) {
    return np_min(np_max(x, lo), hi);
}

// Loads of the lane's two cells.  c0 / c1 are SAFE cell indices (always inside the cube, equal to
// the lane's real cells when those exist, else cells 0 / 1), so the loads are unconditional:
// no branch, no per-load wait - the caller masks the result of invalid lanes afterwards.
// Every input byte is read exactly once -> nontemporal (+5..8 % measured); with VEC the two
// loads fuse into one global_load_dwordx4 nt.
template <bool VEC>
__device__ __forceinline__ double2 ld2(const double *__restrict__ p, int64_t base, int64_t c0, int64_t c1) {
    double2 r;
    if constexpr (VEC) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 t = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + base + c0));
        r.x = t.x;
        r.y = t.y;
    } else {
        r.x = __builtin_nontemporal_load(p + base + c0);
        r.y = __builtin_nontemporal_load(p + base + c1);
    }
    return r;
}

// streaming store of a result cube that is not read again by this kernel
template <bool VEC>
__device__ __forceinline__ void st2(double *__restrict__ p, int64_t off, bool v0, bool v1, double2 v) {
    // nontemporal: +3 % on the 24 B/cell wind series (measured, C3)
    if (v0) __builtin_nontemporal_store(v.x, p + off);
    if (VEC ? v0 : v1) __builtin_nontemporal_store(v.y, p + off + 1);
}

// ---------------------------------------------------------------------------------------
// converters
// ---------------------------------------------------------------------------------------
struct NoCell {};
struct NoCarry {};  // per-wave state a converter may keep across consecutive slots (pv night skip)
// converters may define batch_prefetch<VEC>(sb, send, c0, c1, carry); the others get this no-op
template <bool VEC, class Conv, class Carry>
__device__ __forceinline__ auto batch_prefetch(const Conv &conv, int64_t sb, int64_t send, int64_t c0, int64_t c1,
                                               Carry &carry, int) -> decltype(conv.template batch_prefetch<VEC>(sb, send, c0, c1, carry)) {
    conv.template batch_prefetch<VEC>(sb, send, c0, c1, carry);
}
template <bool VEC, class Conv, class Carry>
__device__ __forceinline__ void batch_prefetch(const Conv &, int64_t, int64_t, int64_t, int64_t, Carry &, long) {}

template <class C>
__device__ __forceinline__ C carry_init() {
    return C{};
}

// generic dense cube (aggregate_matrix on an arbitrary converted DataArray)
struct IdentityConv {
    const double *d;
    int64_t S;
    using Cell = NoCell;
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t, bool, bool, const double *) const { return {}; }
    static constexpr int kGroup = 8;  // slots whose loads are issued before any compute
    using Raw = double2;
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &, Carry &) const {
        return ld2<VEC>(d, slot * S, c0, c1);
    }
    __device__ __forceinline__ double2 compute(const Raw &r, bool, bool, const Cell &, const double *) const {
        return r;
    }
};

// runoff * height  (convert.py:1028-1034)
struct RunoffConv {
    const double *runoff;
    const double *height;  // (S) or nullptr
    int64_t S;
    struct Cell {
        double2 h;
    };
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.h.x = (height && v0) ? height[c0] : 1.0;
        c.h.y = (height && v1) ? height[c0 + 1] : 1.0;
        return c;
    }
    static constexpr int kGroup = 8;
    using Raw = double2;
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &, Carry &) const {
        return ld2<VEC>(runoff, slot * S, c0, c1);
    }
    __device__ __forceinline__ double2 compute(Raw r, bool, bool, const Cell &c, const double *) const {
        if (height) {
            r.x *= c.h.x;
            r.y *= c.h.y;
        }
        return r;
    }
};

// temperature family + heat-pump COP (convert.py:292-364)
struct ThermoConv {
    const double *var;
    int64_t S;
    double offset, sink_T, c0, c1, c2;
    int fillna0, quadratic;
    using Cell = NoCell;
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t, bool, bool, const double *) const { return {}; }
    __device__ __forceinline__ double f(double v) const {
        double x = v + offset;
        if (fillna0) x = fill0(x);
        if (quadratic) {
            const double d = sink_T - x;
            x = c0 + c1 * d + c2 * (d * d);
        }
        return x;
    }
    static constexpr int kGroup = 8;
    using Raw = double2;
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0_, int64_t c1_, const Cell &, Carry &) const {
        return ld2<VEC>(var, slot * S, c0_, c1_);
    }
    __device__ __forceinline__ double2 compute(const Raw &v, bool v0, bool v1, const Cell &, const double *) const {
        double2 r;
        r.x = v0 ? f(v.x) : 0.0;
        r.y = v1 ? f(v.y) : 0.0;
        return r;
    }
};

// heat demand: nan-skipping daily mean, degree-day transform (convert.py:405-418)
struct HeatConv {
    const double *temperature;
    const int64_t *day_ptr;  // device (D+1)
    int64_t S;
    double threshold_K, a, constant;
    int cooling;
    using Cell = NoCell;
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t, bool, bool, const double *) const { return {}; }
    static constexpr int kGroup = 1;  // a slot is a whole day: its own loop keeps 8 loads in flight
    struct Raw {
        double sx, sy;
        int nx, ny;
    };
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &, Carry &) const {
        const int64_t t0 = day_ptr[slot], t1 = day_ptr[slot + 1];
        Raw r{0.0, 0.0, 0, 0};
#pragma unroll 8
        for (int64_t t = t0; t < t1; ++t) {
            const double2 v = ld2<VEC>(temperature, t * S, c0, c1);
            if (!dnan(v.x)) {
                r.sx += v.x;
                ++r.nx;
            }
            if (!dnan(v.y)) {
                r.sy += v.y;
                ++r.ny;
            }
        }
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool, bool, const Cell &, const double *) const {
        // mean over an empty / all-NaN group is NaN (0/0), like xarray's resample().mean()
        const double mx = q.sx / double(q.nx), my = q.sy / double(q.ny);
        double hx = cooling ? a * (mx - threshold_K) : a * (threshold_K - mx);
        double hy = cooling ? a * (my - threshold_K) : a * (threshold_K - my);
        hx = np_max(hx, 0.0);
        hy = np_max(hy, 0.0);
        double2 r;
        r.x = constant + hx;
        r.y = constant + hy;
        return r;
    }
};

// wind: hub-height extrapolation + power curve (wind.py:76-112, convert.py:648-649)
//
// LDS table, built on the host (make_wind): n_pad = power of two > n_knots,
//   V[n_pad]  knots, padded with +inf          (search never needs a bounds check)
//   K[n_pad]  records {V[j], F[j], slope[j], 0} (slope[n-1] = 0)
// np.interp(x, V, F) (numpy arr_interp) for a FINITE table reduces to
//   xc = clamp(x, V[0], V[n-1]);  j = largest index with V[j] <= xc;
//   r  = fma(slope[j], xc - V[j], F[j])
// which returns F[j] exactly on knots, F[0] / F[n-1] outside the range (also for +-inf), NaN
// for NaN, and takes the upper one of repeated knots - all without a branch.  Tables holding
// non-finite values take interp_generic(), the literal transcription of arr_interp.
// METHOD: ATL_WIND_NONE / LOG / POWER fixed at compile time for finite tables (the hot
// instantiations carry no dead paths); METHOD = -1 is the generic converter: runtime method,
// any table (interp_generic).
template <int METHOD>
struct WindConvT {
    const double *wnd;
    const double *aux;
    int64_t S;
    int aux_static;
    int method;
    double to_height, from_height;
    double log_ratio;      // log(to/from)   (power law)
    const double *table;   // device: V[n_pad] | K[n_pad][4]
    int n_knots, n_pad;
    struct Cell {
        double2 aux;
        double lh, lf;  // lean_log(to_height), lean_log(from_height)
    };
    __device__ void block_init(double *lds) const {
        for (int i = threadIdx.x; i < 5 * n_pad; i += blockDim.x) lds[i] = table[i];
        if constexpr (METHOD == ATL_WIND_LOG) log_table_init(lds + 5 * n_pad);
    }
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.aux.x = 0.0;
        c.aux.y = 0.0;
        if (method != ATL_WIND_NONE && aux_static) {
            c.aux.x = v0 ? aux[c0] : 1.0;
            c.aux.y = v1 ? aux[c0 + 1] : 1.0;
        }
        // log(from) through the same routine as the per-cell roughness: z0 == from_height gives an
        // exact zero denominator, like the reference's log(from/z0) = log(1)
        c.lh = 0.0;
        c.lf = 0.0;
        if constexpr (METHOD == ATL_WIND_LOG) c.lf = log_core_tab(from_height, lds + 5 * n_pad);
        return c;
    }
    // wind.py:99-101 / :111, literally; used for the rare arguments the fast path excludes
    __device__ __noinline__ double hub_speed_literal(double v, double z) const {
        if (method == ATL_WIND_LOG) return v * (log(to_height / z) / log(from_height / z));
        if (method == ATL_WIND_POWER) return v * pow(to_height / from_height, z);
        return v;
    }
    // fast path: *rare is set when the literal formula must be used instead
    __device__ __forceinline__ double hub_speed_fast(double v, double z, const Cell &c, bool *rare,
                                                     const double *lds) const {
        if constexpr (METHOD == ATL_WIND_LOG) {
            // v * (log(to/z0) / log(from/z0))  =  v * (1 + log(to/from) / (log(from) - log(z0))):
            // one table-driven log and one reciprocal per cell.  log(from) goes through the same
            // routine, so z0 == from gives den == 0 exactly (-> rare path -> the literal formula).
            const double *ltab = lds + 5 * n_pad;
            const bool zok = z >= 0x1.0p-1022 && z < __builtin_inf();
#ifdef ATL_ABLATE_WIND_NOLOG
            const double lz = z;
#else
            const double lz = log_core_tab(zok ? z : 1.0, ltab);
#endif
            const double den = c.lf - lz;
            const bool tame = __builtin_fabs(den) > 0x1.0p-40;  // |den| < 1500 always
            *rare = !(zok && tame);
            return v * __builtin_fma(log_ratio, fast_rcp(den), 1.0);
        } else if constexpr (METHOD == ATL_WIND_POWER) {
            *rare = false;
            return v * exp(z * log_ratio);  // v * (to/from) ** shear
        } else {
            *rare = false;
            return v;
        }
    }
    __device__ __forceinline__ double interp(double x, const double *lds) const {
        const double *V = lds;
        const double *K = lds + n_pad;
        const double vmin = V[0], vmax = V[n_knots - 1];
        double xc = x > vmax ? vmax : x;
        xc = xc < vmin ? vmin : xc;  // NaN stays NaN
        int j = 0;
        for (int step = n_pad >> 1; step > 0; step >>= 1) {
            const int cand = j + step;
            j = (V[cand] <= xc) ? cand : j;
        }
        const double2 k0 = *reinterpret_cast<const double2 *>(K + 4 * j);
        const double sl = K[4 * j + 2];
        return __builtin_fma(sl, xc - k0.x, k0.y);
    }
    // literal numpy/_core/src/multiarray/compiled_base.c arr_interp (any table)
    __device__ __noinline__ double interp_generic(double x, const double *lds) const {
        const double *V = lds;
        const double *K = lds + n_pad;
        const int n = n_knots;
        if (dnan(x)) return x;
        if (x < V[0]) return K[1];
        if (x > V[n - 1]) return K[4 * (n - 1) + 1];
        int j = 0;
        for (int step = n_pad >> 1; step > 0; step >>= 1) {
            const int cand = j + step;
            if (V[cand] <= x) j = cand;
        }
        const double xj = K[4 * j], fj = K[4 * j + 1];
        if (j == n - 1) return fj;
        if (xj == x) return fj;
        const double slope = (K[4 * (j + 1) + 1] - fj) / (K[4 * (j + 1)] - xj);
        double r = slope * (x - xj) + fj;
        if (dnan(r)) {
            r = slope * (x - K[4 * (j + 1)]) + K[4 * (j + 1) + 1];
            if (dnan(r) && fj == K[4 * (j + 1) + 1]) r = fj;
        }
        return r;
    }
    static constexpr int kGroup = 4;
    struct Raw {
        double2 v, z;
    };
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &c, Carry &) const {
        Raw r;
        r.v = ld2<VEC>(wnd, slot * S, c0, c1);
        r.z = c.aux;
        if (METHOD != ATL_WIND_NONE && !aux_static) r.z = ld2<VEC>(aux, slot * S, c0, c1);
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c,
                                               const double *lds) const {
        const double2 v = q.v, z = q.z;
        double2 r;
        if constexpr (METHOD < 0) {
            r.x = interp_generic(hub_speed_literal(v.x, z.x), lds);
            r.y = interp_generic(hub_speed_literal(v.y, z.y), lds);
        } else {
            bool r0, r1;
            double h0 = hub_speed_fast(v.x, z.x, c, &r0, lds), h1 = hub_speed_fast(v.y, z.y, c, &r1, lds);
            if ((r0 && v0) || (r1 && v1)) {  // degenerate roughness: literal formula, out of line
                h0 = hub_speed_literal(v.x, z.x);
                h1 = hub_speed_literal(v.y, z.y);
            }
#ifdef ATL_ABLATE_WIND_NOINTERP
            r.x = h0;
            r.y = h1;
#else
            r.x = interp(h0, lds);
            r.y = interp(h1, lds);
#endif
        }
        r.x = v0 ? r.x : 0.0;
        r.y = v1 ? r.y : 0.0;
        return r;
    }
};

// solar PV, ERA5-shaped inputs with stored solar position
struct PvConst {
    double c_amb, c_irr, r_tmod, inv_r_irr, k1, k2, k3, k4, k5, k6, inv_eff, alt_thr, sin_alt_thr;
};

// per-cell orientation factors: sin/cos(slope), (1 +- cos(slope))/2, panel azimuth
struct PvOri {
    double ss, cs, hp, hm, saz;
};
// cos/sin of the panel azimuth: only the in-kernel solar position variant needs them
template <bool SP>
struct PvAz {
    double csaz, ssaz;
};
template <>
struct PvAz<false> {};

// irradiation on the tilted surface + Huld panel model, from sin/cos of the solar altitude and
// cos(surface_azimuth - sun_azimuth)   (irradiation.py:214-226, solar_panel_model.py:22-41)
__device__ __forceinline__ double pv_tail(double direct, double diffuse, double influx, double alb, double tmp,
                                          double sa, double ca, double cosd, const PvOri &o, const PvConst &k) {
    // orientation.py:114-117,188
    double cosinc = o.ss * ca * cosd + o.cs * sa;
    cosinc = np_max(cosinc, 0.0);
    const double kk = fast_div(cosinc, sa);
    const double direct_t = kk * direct;
    const double diffuse_t = o.hp * diffuse;
    const double ground_t = alb * influx * o.hm;
    const double G = fill0(direct_t) + fill0(diffuse_t) + fill0(ground_t);
    const double T_ = (k.c_amb * tmp + k.c_irr * G) - k.r_tmod;
    const double G_ = G * k.inv_r_irr;
    double eff = 0.0;
    if (G_ > 0.0) {
        const double l = lean_log(G_);
        const double l2 = l * l;
        eff = 1.0 + k.k1 * l + k.k2 * l2 + T_ * (k.k3 + k.k4 * l + k.k5 * l2) + k.k6 * (T_ * T_);
        eff = fill0(eff);
        eff = eff < 0.0 ? 0.0 : eff;
    }
    return G_ * eff * k.inv_eff;
}

__device__ __forceinline__ double pv_cell(double dir, double dif, double toa, double alb, double tmp,
                                          double alt, double az, const PvOri &o, const PvConst &k) {
    // irradiation.py:206-208
    const double direct = np_clip(dir, 0.0, toa);
    const double diffuse = np_clip(dif, 0.0, toa - direct);
    const double influx = direct + diffuse;
    // irradiation.py:251-252 (NaN compares false: a NaN altitude is not capped)
    const bool capped = (alt < k.alt_thr) || (influx <= 0.01);
#ifdef ATL_ABLATE_NOMATH  // experiment: keep the 7 streams, drop the physics
    return influx + alb + tmp + alt + az;
#endif
    if (capped) return 0.0;  // G = 0 -> G_ = 0, eff -> 0 : 0*0*inv
    double sa, ca;
    lean_sincos(alt, &sa, &ca);
    return pv_tail(direct, diffuse, influx, alb, tmp, sa, ca, lean_cos(o.saz - az), o, k);
}

// same, with the solar position computed from the separable tables instead of read:
// pv/solar_position.py:100-114.  sin(alt) = s directly, cos(alt) = sqrt(1-s^2),
// cos(az) = clip(.../cos(alt)), sin(az) = +-sqrt(1-cos^2 az) by the sign of the hour angle, so
// cos(surface_az - az) needs no inverse trig at all.  The cut alt < thr becomes s < sin(thr).
__device__ __forceinline__ double pv_cell_sp(double dir, double dif, double toa, double alb, double tmp,
                                             double sd, double cd, double sl, double cl, double h, double ch,
                                             const PvOri &o, const PvAz<true> &a, const PvConst &k) {
    const double direct = np_clip(dir, 0.0, toa);
    const double diffuse = np_clip(dif, 0.0, toa - direct);
    const double influx = direct + diffuse;
    const double s = np_clip(sd * sl + cd * cl * ch, -1.0, 1.0);  // :103-105
    const bool capped = (s < k.sin_alt_thr) || (influx <= 0.01);
    if (capped) return 0.0;
    const double ca = sqrt((1.0 - s) * (1.0 + s));
    const double num = sd * cl - cd * sl * ch;
    double q = fast_div(num, ca);
    if (!(ca > 0x1.0p-500)) q = num / ca;  // zenith / NaN: IEEE division like the reference
    const double caz = np_clip(q, -1.0, 1.0);  // :109-113
    double saz = sqrt((1.0 - caz) * (1.0 + caz));
    saz = (h <= 0.0) ? saz : -saz;  // :114  az = az if h <= 0 else 2 pi - az
    return pv_tail(direct, diffuse, influx, alb, tmp, s, ca, a.csaz * caz + a.ssaz * saz, o, k);
}

// SP: in-kernel solar position; PC: per-cell orientation (else the scalar orientation is read from
// the kernel arguments = SGPRs and costs no per-lane registers)
// SKIP: night early-out (stored solar position only): when every valid cell of the wave is below
// the altitude cut-off the other six streams are not read - the result is exactly +0.0 whatever
// they hold (pv_cell).  The altitude of the NEXT slot is prefetched together with the current slot's
// streams (Carry), so day-time slots still cost one memory round trip.
template <bool SP, bool PC = false, bool SKIP = false>
struct PvConvT {
    static_assert(!(SP && SKIP), "night skip is implemented for stored solar angles");
    atl_pv_inputs in;
    int64_t S;
    PvConst k;
    PvOri o;                     // scalar orientation
    PvAz<SP> oa;
    const double *cell_slope;    // (S) or nullptr
    const double *cell_azimuth;  // (S)
    struct SpCell {
        double sl0, cl0, sl1, cl1;  // sin/cos(lat) of the two cells
        int x0, x1;                 // grid column of the two cells
    };
    struct NoSp {};
    struct OriCell {
        PvOri o0, o1;
        PvAz<SP> a0, a1;
    };
    struct NoOri {};
    struct Cell : std::conditional_t<SP, SpCell, NoSp>, std::conditional_t<PC, OriCell, NoOri> {
        bool no_cell;  // SKIP: this lane owns no cell at all (tile padding)
    };
    __device__ void block_init(double *) const {}
    __device__ static PvOri make_ori(double slope, double azimuth) {
        PvOri r;
        lean_sincos(slope, &r.ss, &r.cs);
        r.hp = (1.0 + r.cs) / 2.0;
        r.hm = (1.0 - r.cs) / 2.0;
        r.saz = azimuth;
        return r;
    }
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.no_cell = !v0 && !v1;
        if constexpr (PC) {
            c.o0 = make_ori(v0 ? cell_slope[c0] : 0.0, v0 ? cell_azimuth[c0] : 0.0);
            c.o1 = make_ori(v1 ? cell_slope[c0 + 1] : 0.0, v1 ? cell_azimuth[c0 + 1] : 0.0);
            if constexpr (SP) {
                lean_sincos(c.o0.saz, &c.a0.ssaz, &c.a0.csaz);
                lean_sincos(c.o1.saz, &c.a1.ssaz, &c.a1.csaz);
            }
        }
        if constexpr (SP) {
            const int64_t a = v0 ? c0 : 0, b = v1 ? c0 + 1 : 0;
            const int64_t y0 = a / in.X, y1 = b / in.X;
            c.x0 = int(a - y0 * in.X);
            c.x1 = int(b - y1 * in.X);
            c.sl0 = in.d_sin_lat[y0];
            c.cl0 = in.d_cos_lat[y0];
            c.sl1 = in.d_sin_lat[y1];
            c.cl1 = in.d_cos_lat[y1];
        }
        return c;
    }
    static constexpr int kGroup = ATL_PV_GROUP;  // 7 x 16 B per lane per slot already; registers are the limit
    struct Raw {
        double2 dir, dif, toa, alb, tmp;
        double2 a, b;    // getter: altitude, azimuth   SP: hour angle, cos(hour angle)
        double sd, cd;   // SP: sin / cos declination of the slot
    };
    struct SkipCarry {
        double2 alt[kBatch];  // solar altitude of the batch's slots, prefetched one batch ahead
    };
    using Carry = std::conditional_t<SKIP, SkipCarry, NoCarry>;
    // called before the first batch and again right after a batch has been converted (i.e. while it
    // is being reduced): the next batch's altitudes are in flight behind the wave reduction
    template <bool VEC>
    __device__ __forceinline__ void batch_prefetch(int64_t sb, int64_t send, int64_t c0, int64_t c1, Carry &carry) const {
        if constexpr (SKIP) {
#pragma unroll
            for (int i = 0; i < kBatch; ++i)
                carry.alt[i] = ld2<VEC>(in.d_solar_altitude, min(sb + i, send - 1) * S, c0, c1);
        }
    }
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int i, int64_t c0, int64_t c1, const Cell &c, Carry &carry) const {
        const int64_t off = slot * S;
        Raw r;
        if constexpr (SKIP) {
            r.a = carry.alt[i];
            r.sd = r.cd = 0.0;
            // capped <=> alt < threshold (a NaN altitude is NOT capped); lanes that own no cell loaded
            // some other cell's altitude and vote "night" unconditionally
            const bool night = (r.a.x < k.alt_thr) && (r.a.y < k.alt_thr);
            if (__all(night || c.no_cell)) {
                const double2 z = {0.0, 0.0};
                r.dir = r.dif = r.toa = r.alb = r.tmp = r.b = z;
                return r;
            }
            r.dir = ld2<VEC>(in.d_influx_direct, off, c0, c1);
            r.dif = ld2<VEC>(in.d_influx_diffuse, off, c0, c1);
            r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
            r.alb = ld2<VEC>(in.d_albedo, off, c0, c1);
            r.tmp = ld2<VEC>(in.d_temperature, off, c0, c1);
            r.b = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
            return r;
        }
        r.dir = ld2<VEC>(in.d_influx_direct, off, c0, c1);
        r.dif = ld2<VEC>(in.d_influx_diffuse, off, c0, c1);
        r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
        r.alb = ld2<VEC>(in.d_albedo, off, c0, c1);
        r.tmp = ld2<VEC>(in.d_temperature, off, c0, c1);
        if constexpr (SP) {
            r.sd = in.d_sin_dec[slot];
            r.cd = in.d_cos_dec[slot];
            const int64_t hb = slot * in.X;
            r.a.x = in.d_hour_angle[hb + c.x0];
            r.a.y = in.d_hour_angle[hb + c.x1];
            r.b.x = in.d_cos_hour_angle[hb + c.x0];
            r.b.y = in.d_cos_hour_angle[hb + c.x1];
        } else {
            r.sd = r.cd = 0.0;
            r.a = ld2<VEC>(in.d_solar_altitude, off, c0, c1);
            r.b = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
        }
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c, const double *) const {
        const PvOri &o0 = [&]() -> const PvOri & { if constexpr (PC) return c.o0; else return o; }();
        const PvOri &o1 = [&]() -> const PvOri & { if constexpr (PC) return c.o1; else return o; }();
        double2 r;
        if constexpr (SP) {
            const PvAz<true> &a0 = [&]() -> const PvAz<true> & { if constexpr (PC) return c.a0; else return oa; }();
            const PvAz<true> &a1 = [&]() -> const PvAz<true> & { if constexpr (PC) return c.a1; else return oa; }();
            r.x = v0 ? pv_cell_sp(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.sd, q.cd, c.sl0, c.cl0, q.a.x, q.b.x, o0, a0, k) : 0.0;
            r.y = v1 ? pv_cell_sp(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.sd, q.cd, c.sl1, c.cl1, q.a.y, q.b.y, o1, a1, k) : 0.0;
        } else {
            r.x = v0 ? pv_cell(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.a.x, q.b.x, o0, k) : 0.0;
            r.y = v1 ? pv_cell(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.a.y, q.b.y, o1, k) : 0.0;
        }
        return r;
    }
};
using PvConv = PvConvT<false>;
using PvConvSP = PvConvT<true>;
template <class T>
struct pv_is_sp : std::false_type {};
template <bool PC, bool SK>
struct pv_is_sp<PvConvT<true, PC, SK>> : std::true_type {};

// ---------------------------------------------------------------------------------------
// general pv converter: every option of convert_pv / convert_irradiation / convert_solar_thermal
// (tracking modes, Hay-Davies, Reindl split, albedo from outflux, bofinger, irradiation
// quantities).  A literal transcription of the reference with full-precision libm - selected
// only when an option differs from the defaults the fast PvConvT path covers.
// ---------------------------------------------------------------------------------------
struct PvxOpt {
    int tracking, trigon, clearsky, irradiation, panel, has_influx, has_albedo;
    double bA, bB, bC, bD, bNOCT, bTstd, bTamb, bIntc, bta, bthr;
    double c0, c1, t_store;
    double r_irr;  // Huld division kept literal here
};

__device__ double pvx_cell(double dir, double dif, double infl, double toa, double albv, double outf, double tmp,
                           double rh, double alt, double az, double slope, double sazim, const PvConst &k,
                           const PvxOpt &o) {
    const double pi = 3.14159265358979323846;
    const double nan = __builtin_nan("");
    const double sa = sin(alt), ca = cos(alt);
    // ---- SurfaceOrientation (orientation.py:113-188) ---------------------------------------
    double surface_slope = slope, cosinc;
    if (o.tracking == ATL_TRACK_NONE) {
        cosinc = sin(slope) * ca * cos(sazim - az) + cos(slope) * sa;
    } else if (o.tracking == ATL_TRACK_HORIZONTAL) {
        const double rotation = atan((ca / sa) * sin(az - sazim));
        surface_slope = fabs(rotation);
        const double surface_azimuth = sazim + asin(sin(rotation) / sin(surface_slope));
        cosinc = cos(surface_slope) * sa + sin(surface_slope) * ca * cos(az - surface_azimuth);
    } else if (o.tracking == ATL_TRACK_TILTED_HORIZONTAL) {
        const double tilt = slope;
        double rotation = atan((ca * sin(az - sazim)) / (ca * cos(az - sazim) * sin(tilt) + sa * cos(tilt)));
        surface_slope = acos(cos(rotation) * cos(tilt));
        double ad = az - sazim;
        ad = ad > pi ? ad - 2 * pi : ad;
        ad = ad < -pi ? 2 * pi + ad : ad;
        rotation = (rotation < 0 && ad > 0) ? rotation + pi : rotation;
        rotation = (rotation > 0 && ad < 0) ? rotation - pi : rotation;
        cosinc = cos(rotation) * (sin(tilt) * ca * cos(az - sazim) + cos(tilt) * sa) + sin(rotation) * ca * sin(az - sazim);
    } else if (o.tracking == ATL_TRACK_VERTICAL) {
        cosinc = sin(slope) * ca + cos(slope) * sa;
    } else {
        cosinc = 1.0;
    }
    cosinc = np_max(cosinc, 0.0);
    // ---- direct / diffuse horizontal (irradiation.py:202-208, 13-73) ------------------------
    double direct, diffuse;
    if (o.has_influx) {
        const double influx = np_clip(infl, 0.0, toa);
        const double kk = influx / toa;
        double fraction;
        const double m1 = (kk > 0.0 && kk <= 0.3) ? 1.0 : 0.0, m2 = (kk > 0.3 && kk < 0.78) ? 1.0 : 0.0,
                     m3 = (kk >= 0.78) ? 1.0 : 0.0;
        if (o.clearsky == ATL_CLEARSKY_SIMPLE) {
            fraction = m1 * fmin(1.0, 1.020 - 0.254 * kk + 0.0123 * sa) +
                       m2 * fmin(0.97, fmax(0.1, 1.400 - 1.749 * kk + 0.177 * sa)) +
                       m3 * fmax(0.1, 0.486 * kk - 0.182 * sa);
        } else {
            fraction = m1 * fmin(1.0, 1.000 - 0.232 * kk + 0.0239 * sa - 0.000682 * tmp + 0.0195 * rh) +
                       m2 * fmin(0.97, fmax(0.1, 1.329 - 1.716 * kk + 0.267 * sa - 0.00357 * tmp + 0.106 * rh)) +
                       m3 * fmax(0.1, 0.426 * kk - 0.256 * sa + 0.00349 * tmp + 0.0734 * rh);
        }
        diffuse = influx * fraction;
        direct = influx - diffuse;
    } else {
        direct = np_clip(dir, 0.0, toa);
        diffuse = np_clip(dif, 0.0, toa - direct);
    }
    const double influx = direct + diffuse;
    // ---- albedo (irradiation.py:128-139) ---------------------------------------------------------
    double alb = albv;
    if (!o.has_albedo) {
        alb = fill0(outf / (influx != 0.0 ? influx : nan));
        alb = np_min(alb, 1.0);
    }
    // ---- tilted irradiation ---------------------------------------------------------------------
    double direct_t, diffuse_t, ground_t, total_t;
    if (o.trigon == ATL_TRIGON_SIMPLE) {
        const double kk = cosinc / sa;
        const double cs = (o.tracking != ATL_TRACK_DUAL) ? cos(surface_slope) : sa;
        direct_t = kk * direct;
        diffuse_t = (1.0 + cs) / 2.0 * diffuse;
        ground_t = alb * influx * ((1.0 - cs) / 2.0);
        total_t = fill0(direct_t) + fill0(diffuse_t) + fill0(ground_t);
    } else {
        const double f = fill0(sqrt(direct / influx));
        const double A = direct / toa;
        const double R_b = cosinc / sa;
        const double sh = sin(surface_slope / 2.0);
        diffuse_t = ((1.0 - A) * ((1 + cos(surface_slope)) / 2.0) * (1.0 + f * (sh * sh * sh)) + A * R_b) * diffuse;
        diffuse_t = fill0(np_max(diffuse_t, 0.0));
        direct_t = R_b * direct;
        ground_t = influx * alb * (1.0 - cos(surface_slope)) / 2.0;
        total_t = direct_t + diffuse_t + ground_t;
    }
    double G = o.irradiation == ATL_IRR_TOTAL    ? total_t
               : o.irradiation == ATL_IRR_DIRECT ? direct_t
               : o.irradiation == ATL_IRR_DIFFUSE ? diffuse_t
                                                  : ground_t;
    if ((alt < k.alt_thr) || (influx <= 0.01)) G = 0.0;  // :251-252
    // ---- panel ------------------------------------------------------------------------------------
    if (o.panel == ATL_PANEL_NONE) return G;
    if (o.panel == ATL_PANEL_HULD) {
        const double T_ = (k.c_amb * tmp + k.c_irr * G) - k.r_tmod;
        const double G_ = G / o.r_irr;
        const double l = log(G_ > 0.0 ? G_ : nan);
        double eff = 1.0 + k.k1 * l + k.k2 * (l * l) + T_ * (k.k3 + k.k4 * l + k.k5 * (l * l)) + k.k6 * (T_ * T_);
        eff = fill0(eff);
        eff = eff < 0.0 ? 0.0 : eff;
        return G_ * eff * k.inv_eff;
    }
    if (o.panel == ATL_PANEL_BOFINGER) {
        const double fraction = (o.bNOCT - o.bTamb) / o.bIntc;
        const double eta_ref = o.bA + o.bB * G + o.bC * log(G != 0.0 ? G : nan);
        const double eta = fill0(eta_ref * (1.0 + o.bD * (fraction * G + (tmp - o.bTstd))) /
                                 (1.0 + o.bD * fraction / o.bta * eta_ref * G));
        const double capacity = (o.bA + o.bB * 1000.0 + o.bC * log(1000.0)) * 1e3;
        const double power = G * eta * (k.inv_eff / capacity);
        return (G >= o.bthr) ? power : 0.0;
    }
    // solar thermal (convert.py:565-574)
    const double eta = o.c0 - o.c1 * fill0((o.t_store - tmp) / (G != 0.0 ? G : nan));
    const double output = G * eta;
    return output > 0.0 ? output : 0.0;
}

struct PvxConv {
    atl_pv_inputs in;
    int64_t S;
    PvConst k;
    PvxOpt o;
    double slope, azimuth;       // scalar orientation (radians)
    const double *cell_slope;    // (S) or nullptr
    const double *cell_azimuth;  // (S)
    struct Cell {
        double sl0, sl1, az0, az1;   // panel slope / azimuth of the two cells
        double slat0, clat0, slat1, clat1;
        int x0, x1;
    };
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.sl0 = c.sl1 = slope;
        c.az0 = c.az1 = azimuth;
        if (cell_slope) {
            c.sl0 = v0 ? cell_slope[c0] : 0.0;
            c.sl1 = v1 ? cell_slope[c0 + 1] : 0.0;
            c.az0 = v0 ? cell_azimuth[c0] : 0.0;
            c.az1 = v1 ? cell_azimuth[c0 + 1] : 0.0;
        }
        c.slat0 = c.clat0 = c.slat1 = c.clat1 = 0.0;
        c.x0 = c.x1 = 0;
        if (!in.d_solar_altitude) {
            const int64_t a = v0 ? c0 : 0, b = v1 ? c0 + 1 : 0;
            const int64_t y0 = a / in.X, y1 = b / in.X;
            c.x0 = int(a - y0 * in.X);
            c.x1 = int(b - y1 * in.X);
            c.slat0 = in.d_sin_lat[y0];
            c.clat0 = in.d_cos_lat[y0];
            c.slat1 = in.d_sin_lat[y1];
            c.clat1 = in.d_cos_lat[y1];
        }
        return c;
    }
    // pv/solar_position.py:100-114, literally
    __device__ static void solar(double sd, double cd, double sl, double cl, double h, double ch, double *alt,
                                 double *az) {
        const double a = asin(np_clip(sd * sl + cd * cl * ch, -1.0, 1.0));
        double z = acos(np_clip((sd * cl - cd * sl * ch) / cos(a), -1.0, 1.0));
        z = (h <= 0.0) ? z : 2.0 * 3.14159265358979323846 - z;
        *alt = a;
        *az = z;
    }
    static constexpr int kGroup = 1;
    struct Raw {
        double2 dir, dif, inf, toa, alb, ouf, tmp, hum, alt, az;
    };
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &c, Carry &) const {
        const int64_t off = slot * S;
        const double2 zero = {0.0, 0.0};
        Raw r;
        r.dir = in.d_influx_direct ? ld2<VEC>(in.d_influx_direct, off, c0, c1) : zero;
        r.dif = in.d_influx_diffuse ? ld2<VEC>(in.d_influx_diffuse, off, c0, c1) : zero;
        r.inf = in.d_influx ? ld2<VEC>(in.d_influx, off, c0, c1) : zero;
        r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
        r.alb = in.d_albedo ? ld2<VEC>(in.d_albedo, off, c0, c1) : zero;
        r.ouf = in.d_outflux ? ld2<VEC>(in.d_outflux, off, c0, c1) : zero;
        r.tmp = in.d_temperature ? ld2<VEC>(in.d_temperature, off, c0, c1) : zero;
        r.hum = in.d_humidity ? ld2<VEC>(in.d_humidity, off, c0, c1) : zero;
        if (in.d_solar_altitude) {
            r.alt = ld2<VEC>(in.d_solar_altitude, off, c0, c1);
            r.az = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
        } else {
            const double sd = in.d_sin_dec[slot], cd = in.d_cos_dec[slot];
            const int64_t hb = slot * in.X;
            solar(sd, cd, c.slat0, c.clat0, in.d_hour_angle[hb + c.x0], in.d_cos_hour_angle[hb + c.x0], &r.alt.x, &r.az.x);
            solar(sd, cd, c.slat1, c.clat1, in.d_hour_angle[hb + c.x1], in.d_cos_hour_angle[hb + c.x1], &r.alt.y, &r.az.y);
        }
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c, const double *) const {
        double2 r;
        r.x = v0 ? pvx_cell(q.dir.x, q.dif.x, q.inf.x, q.toa.x, q.alb.x, q.ouf.x, q.tmp.x, q.hum.x, q.alt.x, q.az.x, c.sl0, c.az0, k, o) : 0.0;
        r.y = v1 ? pvx_cell(q.dir.y, q.dif.y, q.inf.y, q.toa.y, q.alb.y, q.ouf.y, q.tmp.y, q.hum.y, q.alt.y, q.az.y, c.sl1, c.az1, k, o) : 0.0;
        return r;
    }
};

// ---------------------------------------------------------------------------------------
// kernel 1: per-cell series  out[slot, cell]
// grid.x over 512-cell blocks, grid.y over slot chunks of kSeriesSlots
// ---------------------------------------------------------------------------------------
constexpr int kSeriesSlots = 32;

template <class Conv, bool VEC>
__global__ __launch_bounds__(256) void k_cells_series(Conv conv, int64_t n_slots, int64_t S,
                                                      double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int64_t c0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 2;
    const bool v0 = c0 < S, v1 = c0 + 1 < S;
    const int64_t s0c = v0 ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const int64_t s0 = int64_t(blockIdx.y) * kSeriesSlots;
    const int64_t s1 = min(s0 + int64_t(kSeriesSlots), n_slots);
    constexpr int G = Conv::kGroup >= 4 ? 4 : Conv::kGroup;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    for (int64_t sg = s0; sg < s1; sg += G) {
        typename Conv::Raw raw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) raw[g] = conv.template load<VEC>(min(sg + g, s1 - 1), g, s0c, s1c, cell, carry);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double2 r = conv.compute(raw[g], v0, v1, cell, lds);
            if (sg + g < s1) st2<VEC>(out, (sg + g) * S + c0, v0, v1, r);
        }
    }
}

// ---------------------------------------------------------------------------------------
// kernel 2: per-cell time reduction.  psum/pcnt[chunk, cell] then k_chunk_reduce.
// ---------------------------------------------------------------------------------------
template <class Conv, bool VEC>
__global__ __launch_bounds__(256) void k_cells_timered(Conv conv, int64_t n_slots, int64_t S,
                                                       int64_t chunk_len, double *__restrict__ psum,
                                                       double *__restrict__ pcnt) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int64_t c0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 2;
    const bool v0 = c0 < S, v1 = c0 + 1 < S;
    const int64_t s0c = v0 ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const int64_t s0 = int64_t(blockIdx.y) * chunk_len;
    const int64_t s1 = min(s0 + chunk_len, n_slots);
    double2 acc = {0.0, 0.0}, cnt = {0.0, 0.0};
    constexpr int G = Conv::kGroup >= 4 ? 4 : Conv::kGroup;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    for (int64_t sg = s0; sg < s1; sg += G) {
        typename Conv::Raw raw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) raw[g] = conv.template load<VEC>(min(sg + g, s1 - 1), g, s0c, s1c, cell, carry);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double2 r = conv.compute(raw[g], v0, v1, cell, lds);
            const bool live = sg + g < s1;
            if (live && !dnan(r.x)) {
                acc.x += r.x;
                cnt.x += 1.0;
            }
            if (live && !dnan(r.y)) {
                acc.y += r.y;
                cnt.y += 1.0;
            }
        }
    }
    const int64_t o = int64_t(blockIdx.y) * S + c0;
    if (v0) {
        psum[o] = acc.x;
        pcnt[o] = cnt.x;
    }
    if (v1) {
        psum[o + 1] = acc.y;
        pcnt[o + 1] = cnt.y;
    }
}

__global__ __launch_bounds__(256) void k_chunk_reduce(const double *__restrict__ psum,
                                                      const double *__restrict__ pcnt, int64_t n_chunks,
                                                      int64_t S, int mean, double *__restrict__ out) {
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (c >= S) return;
    double s = 0.0, n = 0.0;
    for (int64_t k = 0; k < n_chunks; ++k) {
        s += psum[k * S + c];
        n += pcnt[k * S + c];
    }
    out[c] = mean ? s / n : s;  // nan-skipping mean of nothing is NaN; nan-skipping sum is 0
}

// ---------------------------------------------------------------------------------------
// kernel 3: fused convert + segment reduce
// ---------------------------------------------------------------------------------------
// ---- wave butterfly -----------------------------------------------------------------------
// c[i] (i = slot in batch) per lane -> every lane of the 8-lane group g holds sum over all 64
// lanes of c[g].  Stage 32 and 16 use the gfx950 lane-swap instructions (v_permlane32_swap /
// v_permlane16_swap: no selects, no LDS), stages 8..1 are DPP moves inside a row of 16 lanes.
// Deterministic: a fixed reduction tree.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double swap_add32(double a, double b) {
    // a kept by lanes 0-31, b kept by lanes 32-63:  lo: a[l] + a[l+32]   hi: b[l-32] + b[l]
    const u32x2 lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}

__device__ __forceinline__ double swap_add16(double a, double b) {
    // a kept by even rows of 16 lanes, b kept by odd rows
    const u32x2 lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_mov(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

constexpr int kDppRor8 = 0x128, kDppHalfMirror = 0x141, kDppQuad1032 = 0xB1, kDppQuad2301 = 0x4E;

__device__ __forceinline__ double butterfly8(const double (&c)[kBatch]) {
    const double d0 = swap_add32(c[0], c[4]), d1 = swap_add32(c[1], c[5]);
    const double d2 = swap_add32(c[2], c[6]), d3 = swap_add32(c[3], c[7]);
    const double e0 = swap_add16(d0, d2), e1 = swap_add16(d1, d3);
    // lanes 0-7 of a row keep e0, lanes 8-15 keep e1 (bank masks select the written lanes)
    const double u = dpp_mov<kDppRor8, 0x3>(e1, e0);  // lanes 0-7: e0[l+8]   lanes 8-15: e1[l]
    const double w = dpp_mov<kDppRor8, 0xC>(e0, e1);  // lanes 0-7: e0[l]     lanes 8-15: e1[l-8]
    double f = u + w;
    f += dpp_mov<kDppHalfMirror, 0xF>(f, f);
    f += dpp_mov<kDppQuad1032, 0xF>(f, f);
    f += dpp_mov<kDppQuad2301, 0xF>(f, f);
    return f;  // slot index held by lane l: 4*(l>>5) + 2*((l>>4)&1) + ((l>>3)&1) = (l >> 3)
}

// one partial row: weight the batch, reduce, store 8 consecutive slots
template <bool GUARD>
__device__ __forceinline__ void reduce_row(const double2 (&v)[kBatch], double2 w, bool a0, bool a1, int lane,
                                           int64_t sb, int64_t send, double *__restrict__ prow) {
    double c[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
        if constexpr (GUARD) {
            // structural zeros must not turn NaN/inf cells into NaN (scipy CSR skips them)
            const double t0 = a0 ? w.x * v[i].x : 0.0;
            const double t1 = a1 ? w.y * v[i].y : 0.0;
            c[i] = t0 + t1;
        } else {
            c[i] = __builtin_fma(w.y, v[i].y, w.x * v[i].x);  // w = 0 where absent, v finite
        }
    }
    const double f = butterfly8(c);
    const int g = lane >> 3;
    if ((lane & 7) == 0 && sb + g < send) prow[sb + g] = f;
}

constexpr int kRowCache = ATL_ROW_CACHE;  // partial rows of a tile whose weights stay in registers

template <class Conv, bool VEC>
__global__ __launch_bounds__(256, ATL_FUSED_WAVES) void k_fused_segred(Conv conv, PlanDev plan, int64_t slot0,
                                                      int64_t n_slots, int64_t S, int32_t chunk_slots,
                                                      int64_t n_units, double *__restrict__ partials,
                                                      int64_t ldp) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t unit = int64_t(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    const int32_t seg = int32_t(unit % plan.n_segs);
    const int64_t chunk = unit / plan.n_segs;
    // tile coordinates -> the lane's two adjacent cells
    // (columns sheared by (gy*X) mod 16 cells: tile rows start on 128-byte lines)
    const int32_t ty = seg / plan.ntx, tx = seg - ty * plan.ntx;
    const int64_t gy = int64_t(ty) * (kLanes >> plan.w2_log2) + (lane >> plan.w2_log2);
    const int64_t gx = (int64_t(tx) << (plan.w2_log2 + 1)) + ((lane & ((1 << plan.w2_log2) - 1)) << 1) -
                       ((gy * plan.X) & 15);
    const int64_t c0 = gy * plan.X + gx;
    const bool v0 = gy < plan.Y && gx >= 0 && gx < plan.X;
    const bool v1 = gy < plan.Y && gx + 1 >= 0 && gx + 1 < plan.X;
    const int64_t s0c = v0 ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices: loads never branch
    const int32_t p0 = plan.seg_ptr[seg], p1 = plan.seg_ptr[seg + 1];
    if (p0 == p1) return;  // no shape touches this tile: nothing to read
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    // weights of the first kRowCache partial rows: registers for the whole chunk
    double2 wc[kRowCache];
    unsigned present = 0;  // bit 2r / 2r+1: cell 0 / 1 structurally present in row r
#pragma unroll
    for (int r = 0; r < kRowCache; ++r) {
        wc[r].x = 0.0;
        wc[r].y = 0.0;
        if (p0 + r < p1) {
            const double2 w = *reinterpret_cast<const double2 *>(plan.prow_w + int64_t(p0 + r) * kSegCells + 2 * lane);
            const bool a0 = !dnan(w.x), a1 = !dnan(w.y);
            wc[r].x = a0 ? w.x : 0.0;
            wc[r].y = a1 ? w.y : 0.0;
            present |= (a0 ? 1u : 0u) << (2 * r) | (a1 ? 1u : 0u) << (2 * r + 1);
        }
    }
    // this launch covers output slots [slot0, slot0 + n_slots); partials are window-relative
    const int64_t sbeg = slot0 + chunk * chunk_slots;
    const int64_t send = min(sbeg + int64_t(chunk_slots), slot0 + n_slots);
    partials -= slot0;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    batch_prefetch<VEC>(conv, sbeg, send, s0c, s1c, carry, 0);
    for (int64_t sb = sbeg; sb < send; sb += kBatch) {
        double2 v[kBatch];
        bool finite = true;
        // kGroup slots are LOADED before any of them is converted, so a light converter keeps 8
        // independent 1-KiB loads in flight per wave; slots past the end of a ragged chunk re-load
        // its last slot (loads stay unconditional) and are zeroed afterwards.
        constexpr int G = Conv::kGroup;
#pragma unroll
        for (int i0 = 0; i0 < kBatch; i0 += G) {
            typename Conv::Raw raw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                raw[g] = conv.template load<VEC>(min(sb + i0 + g, send - 1), i0 + g, s0c, s1c, cell, carry);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i = i0 + g;
                const bool live = sb + i < send;
                v[i] = conv.compute(raw[g], v0, v1, cell, lds);
                v[i].x = live ? v[i].x : 0.0;
                v[i].y = live ? v[i].y : 0.0;
                // |x| < inf is false for NaN and +-inf
                finite = finite && (__builtin_fabs(v[i].x) < __builtin_inf()) && (__builtin_fabs(v[i].y) < __builtin_inf());
            }
        }
        if (sb + kBatch < send) batch_prefetch<VEC>(conv, sb + kBatch, send, s0c, s1c, carry, 0);
#ifdef ATL_ABLATE_NOREDUCE  // experiment: conversion only, one dummy store per batch
        {
            double acc = 0.0;
            for (int i = 0; i < kBatch; ++i) acc += v[i].x + v[i].y;
            if (acc == 1.2345e300) partials[sb] = acc;
            continue;
        }
#endif
        const bool all_finite = __all(finite);  // wave-uniform
#pragma unroll
        for (int r = 0; r < kRowCache; ++r) {
            if (p0 + r < p1) {
                double *prow = partials + int64_t(p0 + r) * ldp;
                if (all_finite)
                    reduce_row<false>(v, wc[r], true, true, lane, sb, send, prow);
                else
                    reduce_row<true>(v, wc[r], (present >> (2 * r)) & 1u, (present >> (2 * r + 1)) & 1u, lane, sb,
                                     send, prow);
            }
        }
        for (int32_t p = p0 + kRowCache; p < p1; ++p) {
            const double2 w = *reinterpret_cast<const double2 *>(plan.prow_w + int64_t(p) * kSegCells + 2 * lane);
            const bool a0 = !dnan(w.x), a1 = !dnan(w.y);
            double2 wz;
            wz.x = a0 ? w.x : 0.0;
            wz.y = a1 ? w.y : 0.0;
            reduce_row<true>(v, wz, a0, a1, lane, sb, send, partials + int64_t(p) * ldp);
        }
    }
}

// out[n, t] = sum over the shape's partial rows (ascending segment order)
__global__ __launch_bounds__(256) void k_combine(PlanDev plan, const double *__restrict__ partials,
                                                 int64_t ldp, int64_t n_slots, double *__restrict__ out,
                                                 int64_t ld_out) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    if (t >= n_slots) return;
    double s = 0.0;
    const int32_t q0 = plan.shape_ptr[n], q1 = plan.shape_ptr[n + 1];
    for (int32_t q = q0; q < q1; ++q) s += partials[int64_t(plan.shape_prow[q]) * ldp + t];
    if (plan.row_poison[n]) s = __builtin_nan("");
    out[n * ld_out + t] = s;
}

// nan-skipping sum / mean of each row of a (rows x len) matrix; one block per row
__global__ __launch_bounds__(256) void k_rows_timered(const double *__restrict__ in, int64_t ld,
                                                      int64_t len, int mean, double *__restrict__ out) {
    __shared__ double ss[256], sn[256];
    const double *row = in + int64_t(blockIdx.x) * ld;
    double s = 0.0, n = 0.0;
    for (int64_t t = threadIdx.x; t < len; t += 256) {
        const double v = row[t];
        if (!dnan(v)) {
            s += v;
            n += 1.0;
        }
    }
    ss[threadIdx.x] = s;
    sn[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) {
            ss[threadIdx.x] += ss[threadIdx.x + w];
            sn[threadIdx.x] += sn[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = mean ? ss[0] / sn[0] : ss[0];
}

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

__global__ __launch_bounds__(256) void k_synth_pv(atl_synth_solar s, int64_t T, int64_t S, double *dir,
                                                  double *dif, double *toa, double *alb, double *tmp,
                                                  double *altp, double *azp) {
    const int64_t n = T * S;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const int64_t t = i / S, c = i % S;
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
        altp[i] = alt;
        azp[i] = az;
        toa[i] = top;
        dir[i] = top * kt * fd;
        dif[i] = top * kt * (1.0 - fd);
        alb[i] = 0.05 + 0.30 * u3;
        tmp[i] = s.d_tseason[t] - 0.4 * (lat * (180.0 / M_PI) - 50.0) + 4.0 * (u4 - 0.5);
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
        default: r = fast_div(x, in[n + i]); break;
    }
    out[i] = r;
}

// ---------------------------------------------------------------------------------------
// host-side launch plumbing
// ---------------------------------------------------------------------------------------
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct KernelBracket {
    atl_ctx *ctx;
    explicit KernelBracket(atl_ctx *c) : ctx(c) {
        if (ctx->profiling) (void)hipEventRecord(ctx->ev_k0, ctx->stream);
    }
    ~KernelBracket() {
        if (ctx->profiling) {
            (void)hipEventRecord(ctx->ev_k1, ctx->stream);
            ctx->have_kernel_time = true;
        }
    }
};

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

// chunk of output slots walked by one wave of the fused kernel
int32_t pick_chunk_slots(const atl_ctx *ctx, int64_t n_slots, int64_t n_segs) {
    // aim for >= ~16 waves per CU worth of units, chunks a multiple of kBatch in [8, 64]
    int64_t chunk = 64;
    const int64_t want_units = int64_t(ctx->n_cu) * 64;
    while (chunk > kBatch && n_segs * ((n_slots + chunk - 1) / chunk) < want_units) chunk /= 2;
    return int32_t(chunk);
}

template <class Conv>
int run_cells(atl_ctx *ctx, const Conv &conv, bool vec, size_t lds_bytes, int64_t n_slots, int64_t S,
              int time_agg, double *d_out, const char *what) {
    ATL_REQUIRE(time_agg == ATL_TIME_NONE || time_agg == ATL_TIME_SUM || time_agg == ATL_TIME_MEAN,
                "%s: bad time_agg %d", what, time_agg);
    ATL_REQUIRE(d_out, "%s: d_out is NULL", what);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (S == 0 || (n_slots == 0 && time_agg == ATL_TIME_NONE)) return ATL_OK;
    const unsigned gx = unsigned((S + 511) / 512);
    vec = vec && aligned16(d_out);
    if (time_agg == ATL_TIME_NONE) {
        const dim3 grid(gx, unsigned((n_slots + kSeriesSlots - 1) / kSeriesSlots));
        KernelBracket kb(ctx);
        if (vec)
            hipLaunchKernelGGL((k_cells_series<Conv, true>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, d_out);
        else
            hipLaunchKernelGGL((k_cells_series<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, d_out);
        return check_launch(what);
    }
    // time-reduced: split the slot axis so that the grid fills the chip
    int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>((n_slots + 15) / 16,
                                                               (int64_t(ctx->n_cu) * 16 + gx - 1) / gx));
    const int64_t chunk_len = std::max<int64_t>(1, (n_slots + n_chunks - 1) / n_chunks);
    n_chunks = std::max<int64_t>(1, (n_slots + chunk_len - 1) / chunk_len);
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, size_t(2 * n_chunks * S) * sizeof(double), &scr);
    if (rc) return rc;
    double *psum = static_cast<double *>(scr), *pcnt = psum + n_chunks * S;
    {
        const dim3 grid(gx, unsigned(n_chunks));
        KernelBracket kb(ctx);
        if (vec)
            hipLaunchKernelGGL((k_cells_timered<Conv, true>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, chunk_len, psum, pcnt);
        else
            hipLaunchKernelGGL((k_cells_timered<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, chunk_len, psum, pcnt);
    }
    if ((rc = check_launch(what))) return rc;
    hipLaunchKernelGGL(k_chunk_reduce, dim3(unsigned((S + 255) / 256)), dim3(256), 0, ctx->stream, psum, pcnt,
                       n_chunks, S, time_agg == ATL_TIME_MEAN ? 1 : 0, d_out);
    return check_launch(what);
}

template <class Conv>
int run_fused(atl_ctx *ctx, const Conv &conv, bool vec, size_t lds_bytes, int64_t n_slots, int64_t S,
              const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out, const char *what) {
    ATL_REQUIRE(agg, "%s: agg is NULL", what);
    ATL_REQUIRE(agg->ctx == ctx, "%s: aggregation plan belongs to another context", what);
    ATL_REQUIRE(agg->dev.n_cells == S, "%s: matrix has %lld columns but the cutout has %lld cells", what,
                (long long)agg->dev.n_cells, (long long)S);
    ATL_REQUIRE(time_agg == ATL_TIME_NONE || time_agg == ATL_TIME_SUM || time_agg == ATL_TIME_MEAN,
                "%s: bad time_agg %d", what, time_agg);
    ATL_REQUIRE(d_out, "%s: d_out is NULL", what);
    ATL_REQUIRE(time_agg != ATL_TIME_NONE || ld_out >= n_slots, "%s: ld_out %lld < %lld", what,
                (long long)ld_out, (long long)n_slots);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    const PlanDev &plan = agg->dev;
    const int64_t N = plan.n_rows;
    if (N == 0) return ATL_OK;
    vec = vec && (plan.X % 2 == 0);  // the lane's cell pair must not straddle a grid row
    // Partial rows live in scratch as [P][window]; the slot axis is processed in windows so that the
    // scratch stays below ~1 GiB however dense the matrix is (P = tiles x shapes for a dense one).
    const int64_t P = plan.n_prows;
    int64_t window = std::max<int64_t>(n_slots, 1);
    int64_t budget = int64_t(1) << 27;  // doubles = 1 GiB
    if (const char *env = getenv("ATLITE_HIP_PARTIAL_BUDGET")) budget = std::max<int64_t>(1, atoll(env));  // tests
    const int64_t budget_slots = budget / std::max<int64_t>(P, 1);
    if (window > budget_slots) window = std::max<int64_t>(64, budget_slots / 64 * 64);
    const int64_t ldp = int64_t(align_up(size_t(window), 8));
    const int64_t lds_series = int64_t(align_up(size_t(std::max<int64_t>(n_slots, 1)), 8));
    size_t bytes_partials = align_up(size_t(std::max<int64_t>(P, 1) * ldp) * sizeof(double), 256);
    size_t bytes_series = time_agg == ATL_TIME_NONE ? 0 : align_up(size_t(N * lds_series) * sizeof(double), 256);
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, bytes_partials + bytes_series, &scr);
    if (rc) return rc;
    double *partials = static_cast<double *>(scr);
    double *series = time_agg == ATL_TIME_NONE
                         ? d_out
                         : reinterpret_cast<double *>(static_cast<char *>(scr) + bytes_partials);
    const int64_t ld_series = time_agg == ATL_TIME_NONE ? ld_out : lds_series;
    for (int64_t w0 = 0; w0 < n_slots; w0 += window) {
        const int64_t wn = std::min(window, n_slots - w0);
        if (P > 0) {
            const int32_t chunk_slots = pick_chunk_slots(ctx, wn, plan.n_segs);
            const int64_t n_chunks = (wn + chunk_slots - 1) / chunk_slots;
            const int64_t n_units = n_chunks * plan.n_segs;
            const dim3 grid(unsigned((n_units + kWavesPerBlock - 1) / kWavesPerBlock));
            KernelBracket kb(ctx);
            if (vec)
                hipLaunchKernelGGL((k_fused_segred<Conv, true>), grid, dim3(256), lds_bytes, ctx->stream, conv, plan,
                                   w0, wn, S, chunk_slots, n_units, partials, ldp);
            else
                hipLaunchKernelGGL((k_fused_segred<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                                   plan, w0, wn, S, chunk_slots, n_units, partials, ldp);
            if ((rc = check_launch(what))) return rc;
        }
        const dim3 grid(unsigned((wn + 255) / 256), unsigned(N));
        hipLaunchKernelGGL(k_combine, grid, dim3(256), 0, ctx->stream, plan, partials, ldp, wn, series + w0,
                           ld_series);
        if ((rc = check_launch(what))) return rc;
    }
    if (time_agg != ATL_TIME_NONE) {
        hipLaunchKernelGGL(k_rows_timered, dim3(unsigned(N)), dim3(256), 0, ctx->stream, series, ld_series,
                           n_slots, time_agg == ATL_TIME_MEAN ? 1 : 0, d_out);
        if ((rc = check_launch(what))) return rc;
    }
    return ATL_OK;
}

bool vec_ok(int64_t S, std::initializer_list<const void *> ptrs) {
    if (S % 2) return false;
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

// ---- converter construction + validation ---------------------------------------------
template <class PV>
int make_pv(const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PV *c, bool *vec) {
    ATL_REQUIRE(in && p, "atl_pv: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_pv: negative shape");
    ATL_REQUIRE(in->d_influx_direct && in->d_influx_diffuse && in->d_influx_toa,
                "atl_pv: need influx_direct, influx_diffuse and influx_toa (irradiation.py:209-213)");
    ATL_REQUIRE(in->d_albedo, "atl_pv: need albedo (irradiation.py:128-139)");
    ATL_REQUIRE(in->d_temperature, "atl_pv: need temperature");
    if (in->d_solar_altitude || in->d_solar_azimuth) {
        ATL_REQUIRE(in->d_solar_altitude && in->d_solar_azimuth,
                    "atl_pv: solar_altitude and solar_azimuth must be given together");
    } else {
        ATL_REQUIRE(in->d_sin_dec && in->d_cos_dec && in->d_hour_angle && in->d_cos_hour_angle && in->d_sin_lat &&
                        in->d_cos_lat,
                    "atl_pv: need either solar_altitude/solar_azimuth or the solar position tables");
        ATL_REQUIRE(in->X > 0 && S % in->X == 0, "atl_pv: X (%lld) must divide the number of cells (%lld)",
                    (long long)in->X, (long long)S);
    }
    ATL_REQUIRE((p->d_cell_slope == nullptr) == (p->d_cell_azimuth == nullptr),
                "atl_pv: d_cell_slope and d_cell_azimuth must be given together");
    c->in = *in;
    c->S = S;
    c->k = PvConst{p->c_temp_amb, p->c_temp_irrad, p->r_tmod, 1.0 / p->r_irradiance, p->k_1, p->k_2,
                   p->k_3,        p->k_4,          p->k_5,    p->k_6,                 p->inverter_efficiency,
                   p->altitude_threshold, sin(p->altitude_threshold)};
    c->o.ss = sin(p->slope);
    c->o.cs = cos(p->slope);
    c->o.hp = (1.0 + c->o.cs) / 2.0;
    c->o.hm = (1.0 - c->o.cs) / 2.0;
    c->o.saz = p->azimuth;
    if constexpr (pv_is_sp<PV>::value) {
        c->oa.csaz = cos(p->azimuth);
        c->oa.ssaz = sin(p->azimuth);
    }
    c->cell_slope = p->d_cell_slope;
    c->cell_azimuth = p->d_cell_azimuth;
    *vec = vec_ok(S, {in->d_influx_direct, in->d_influx_diffuse, in->d_influx_toa, in->d_albedo,
                      in->d_temperature, in->d_solar_altitude, in->d_solar_azimuth});
    return ATL_OK;
}

// f(converter instance) with the PvConvT instantiation for (stored / computed solar position,
// scalar / per-cell orientation)
template <class F>
int pv_dispatch(const atl_pv_inputs *in, const atl_pv_params *p, bool allow_skip, F &&f) {
    const bool sp = !(in->d_solar_altitude || in->d_solar_azimuth), pc = p->d_cell_slope != nullptr;
    if (sp) return pc ? f(PvConvT<true, true>()) : f(PvConvT<true, false>());
    if (p->night_skip && allow_skip) return pc ? f(PvConvT<false, true, true>()) : f(PvConvT<false, false, true>());
    return pc ? f(PvConvT<false, true>()) : f(PvConvT<false, false>());
}

bool pv_needs_general(const atl_pv_inputs *in, const atl_pv_params *p) {
    return p->tracking != ATL_TRACK_NONE || p->trigon_model != ATL_TRIGON_SIMPLE || p->irradiation != ATL_IRR_TOTAL ||
           p->panel_model != ATL_PANEL_HULD || in->d_influx != nullptr || in->d_albedo == nullptr;
}

int make_pvx(const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PvxConv *c, bool *vec) {
    ATL_REQUIRE(in && p, "atl_pv: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_pv: negative shape");
    ATL_REQUIRE(p->tracking >= ATL_TRACK_NONE && p->tracking <= ATL_TRACK_DUAL, "atl_pv: bad tracking code %d",
                p->tracking);
    ATL_REQUIRE(p->trigon_model == ATL_TRIGON_SIMPLE || p->trigon_model == ATL_TRIGON_OTHER,
                "atl_pv: bad trigon_model code %d", p->trigon_model);
    ATL_REQUIRE(p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND, "atl_pv: bad irradiation code %d",
                p->irradiation);
    ATL_REQUIRE(p->panel_model >= ATL_PANEL_HULD && p->panel_model <= ATL_PANEL_SOLAR_THERMAL,
                "atl_pv: bad panel_model code %d", p->panel_model);
    ATL_REQUIRE(in->d_influx_toa, "atl_pv: need influx_toa");
    if (in->d_influx) {
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || p->clearsky_model == ATL_CLEARSKY_ENHANCED,
                    "`clearsky model` must be chosen from 'simple' and 'enhanced'");
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || (in->d_temperature && in->d_humidity),
                    "atl_pv: the enhanced clearsky model needs temperature and humidity");
    } else {
        ATL_REQUIRE(in->d_influx_direct && in->d_influx_diffuse,
                    "Need either influx or influx_direct and influx_diffuse in the dataset. Check your cutout and "
                    "dataset module.");
    }
    ATL_REQUIRE(in->d_albedo || in->d_outflux,
                "Need either albedo or outflux as a variable in the dataset. Check your cutout and dataset module.");
    ATL_REQUIRE(p->panel_model == ATL_PANEL_NONE || in->d_temperature, "atl_pv: need temperature");
    if (in->d_solar_altitude || in->d_solar_azimuth) {
        ATL_REQUIRE(in->d_solar_altitude && in->d_solar_azimuth,
                    "atl_pv: solar_altitude and solar_azimuth must be given together");
    } else {
        ATL_REQUIRE(in->d_sin_dec && in->d_cos_dec && in->d_hour_angle && in->d_cos_hour_angle && in->d_sin_lat &&
                        in->d_cos_lat,
                    "atl_pv: need either solar_altitude/solar_azimuth or the solar position tables");
        ATL_REQUIRE(in->X > 0 && S % in->X == 0, "atl_pv: X (%lld) must divide the number of cells (%lld)",
                    (long long)in->X, (long long)S);
    }
    ATL_REQUIRE((p->d_cell_slope == nullptr) == (p->d_cell_azimuth == nullptr),
                "atl_pv: d_cell_slope and d_cell_azimuth must be given together");
    c->in = *in;
    c->S = S;
    c->k = PvConst{p->c_temp_amb, p->c_temp_irrad, p->r_tmod, 1.0 / p->r_irradiance, p->k_1, p->k_2,
                   p->k_3,        p->k_4,          p->k_5,    p->k_6,                 p->inverter_efficiency,
                   p->altitude_threshold, sin(p->altitude_threshold)};
    c->o = PvxOpt{p->tracking, p->trigon_model, p->clearsky_model, p->irradiation, p->panel_model,
                  in->d_influx ? 1 : 0, in->d_albedo ? 1 : 0,
                  p->bof_A, p->bof_B, p->bof_C, p->bof_D, p->bof_NOCT, p->bof_Tstd, p->bof_Tamb, p->bof_Intc, p->bof_ta,
                  p->bof_threshold, p->st_c0, p->st_c1, p->st_t_store_K, p->r_irradiance};
    c->slope = p->slope;
    c->azimuth = p->azimuth;
    c->cell_slope = p->d_cell_slope;
    c->cell_azimuth = p->d_cell_azimuth;
    *vec = vec_ok(S, {in->d_influx_direct, in->d_influx_diffuse, in->d_influx, in->d_influx_toa, in->d_albedo,
                      in->d_outflux, in->d_temperature, in->d_humidity, in->d_solar_altitude, in->d_solar_azimuth});
    return ATL_OK;
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
    ATL_REQUIRE(p->n_knots >= 1 && p->n_knots <= kMaxKnots && p->h_V && p->h_POWn,
                "atl_wind: power curve needs 1..%d knots", kMaxKnots);
    const int n = p->n_knots;
    int n_pad = 2;
    while (n_pad <= n) n_pad *= 2;  // power of two > n: V[n..n_pad) = +inf
    std::vector<double> tbl(size_t(5 * n_pad), 0.0);
    bool finite = true;
    for (int i = 0; i < n_pad; ++i) tbl[i] = std::numeric_limits<double>::infinity();
    for (int i = 0; i < n; ++i) {
        ATL_REQUIRE(i == 0 || p->h_V[i] >= p->h_V[i - 1],
                    "wind speed 'V' in the turbine config is expected to be increasing");
        tbl[i] = p->h_V[i];
        double *k = &tbl[size_t(n_pad) + 4 * size_t(i)];
        k[0] = p->h_V[i];
        k[1] = p->h_POWn[i];
        // slope as numpy precomputes it; only the upper one of repeated knots is ever selected
        k[2] = (i + 1 < n && p->h_V[i + 1] > p->h_V[i]) ? (p->h_POWn[i + 1] - p->h_POWn[i]) / (p->h_V[i + 1] - p->h_V[i])
                                                        : 0.0;
        finite = finite && std::isfinite(k[0]) && std::isfinite(k[1]) && std::isfinite(k[2]);
    }
    // stream-ordered after any earlier kernel that still reads the table
    ATL_HIP_TRY(hipMemcpyAsync(ctx->d_table, tbl.data(), tbl.size() * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    c->wnd = in->d_wnd;
    c->aux = in->d_aux;
    c->S = S;
    c->aux_static = in->aux_is_static;
    c->method = p->method;
    c->to_height = p->to_height;
    c->from_height = p->from_height;
    c->log_ratio = log(p->to_height / p->from_height);
    c->table = ctx->d_table;
    c->n_knots = n;
    c->n_pad = n_pad;
    *table_finite = finite;
    *lds_bytes = size_t(5 * n_pad + 2 * kLogTabN) * sizeof(double);
    *vec = vec_ok(S, {in->d_wnd, in->aux_is_static ? nullptr : in->d_aux});
    return ATL_OK;
}

int make_heat(const double *d_temperature, const atl_heat_params *p, int64_t T, int64_t S, HeatConv *c,
              bool *vec) {
    ATL_REQUIRE(d_temperature && p, "atl_heat_demand: temperature/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0 && p->n_days >= 0, "atl_heat_demand: negative shape");
    ATL_REQUIRE(p->n_days == 0 || p->d_day_ptr, "atl_heat_demand: d_day_ptr is NULL");
    c->temperature = d_temperature;
    c->day_ptr = p->d_day_ptr;
    c->S = S;
    c->threshold_K = p->threshold_K;
    c->a = p->a;
    c->constant = p->constant;
    c->cooling = p->cooling ? 1 : 0;
    *vec = vec_ok(S, {d_temperature});
    return ATL_OK;
}

template <int M>
WindConvT<M> wind_as(const WindConvT<-1> &g) {
    WindConvT<M> c;
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
    return c;
}

// run `f(converter)` with the instantiation matching (method, table finiteness)
template <class F>
int wind_dispatch(const WindConvT<-1> &g, bool finite, F &&f) {
    // the fast log-law path also needs positive, finite heights (their logs are taken once)
    const bool heights_ok = g.to_height > 0 && g.from_height > 0 && std::isfinite(g.to_height) &&
                            std::isfinite(g.from_height);
    if (!finite || (g.method == ATL_WIND_LOG && !heights_ok)) return f(g);
    switch (g.method) {
        case ATL_WIND_LOG: return f(wind_as<ATL_WIND_LOG>(g));
        case ATL_WIND_POWER: return f(wind_as<ATL_WIND_POWER>(g));
        default: return f(wind_as<ATL_WIND_NONE>(g));
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int atl_spmm_csr(atl_ctx *ctx, const atl_agg *agg, const double *d_dense, int64_t T, int64_t S,
                 int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && d_dense, "atl_spmm_csr: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_spmm_csr: negative shape");
    IdentityConv c{d_dense, S};
    return run_fused(ctx, c, vec_ok(S, {d_dense}), 0, T, S, agg, time_agg, d_out, ld_out, "atl_spmm_csr");
}

int atl_pv_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                   int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert: ctx/inputs/params is NULL");
    bool vec;
    if (pv_needs_general(in, p)) {
        PvxConv c;
        int rc = make_pvx(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
    }
    return pv_dispatch(in, p, false, [&](auto c) {  // night skip: fused (aggregating) kernel only
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
    });
}

int atl_pv_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                             int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert_aggregate: ctx/inputs/params is NULL");
    bool vec;
    if (pv_needs_general(in, p)) {
        PvxConv c;
        int rc = make_pvx(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    }
    return pv_dispatch(in, p, true, [&](auto c) {
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
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
    int rc = make_heat(d_temperature, p, T, S, &c, &vec);
    if (rc) return rc;
    return run_cells(ctx, c, vec, 0, p->n_days, S, time_agg, d_out, "atl_heat_demand_convert");
}

int atl_heat_demand_convert_aggregate(atl_ctx *ctx, const double *d_temperature, const atl_heat_params *p,
                                      int64_t T, int64_t S, const atl_agg *agg, int time_agg,
                                      double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx, "atl_heat_demand_convert_aggregate: ctx is NULL");
    HeatConv c;
    bool vec;
    int rc = make_heat(d_temperature, p, T, S, &c, &vec);
    if (rc) return rc;
    return run_fused(ctx, c, vec, 0, p->n_days, S, agg, time_agg, d_out, ld_out,
                     "atl_heat_demand_convert_aggregate");
}

int atl_thermo_convert(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T, int64_t S,
                       int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && d_var && p, "atl_thermo_convert: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_thermo_convert: negative shape");
    ThermoConv c{d_var, S, p->offset, p->sink_T, p->c0, p->c1, p->c2, p->fillna0, p->quadratic};
    return run_cells(ctx, c, vec_ok(S, {d_var}), 0, T, S, time_agg, d_out, "atl_thermo_convert");
}

int atl_thermo_convert_aggregate(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T,
                                 int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && d_var && p, "atl_thermo_convert_aggregate: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_thermo_convert_aggregate: negative shape");
    ThermoConv c{d_var, S, p->offset, p->sink_T, p->c0, p->c1, p->c2, p->fillna0, p->quadratic};
    return run_fused(ctx, c, vec_ok(S, {d_var}), 0, T, S, agg, time_agg, d_out, ld_out,
                     "atl_thermo_convert_aggregate");
}

int atl_runoff_convert(atl_ctx *ctx, const double *d_runoff, const double *d_height, int64_t T, int64_t S,
                       int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && d_runoff, "atl_runoff_convert: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_runoff_convert: negative shape");
    RunoffConv c{d_runoff, d_height, S};
    return run_cells(ctx, c, vec_ok(S, {d_runoff}), 0, T, S, time_agg, d_out, "atl_runoff_convert");
}

int atl_runoff_convert_aggregate(atl_ctx *ctx, const double *d_runoff, const double *d_height, int64_t T,
                                 int64_t S, const atl_agg *agg, int time_agg, double *d_out,
                                 int64_t ld_out) {
    ATL_REQUIRE(ctx && d_runoff, "atl_runoff_convert_aggregate: bad argument");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_runoff_convert_aggregate: negative shape");
    RunoffConv c{d_runoff, d_height, S};
    return run_fused(ctx, c, vec_ok(S, {d_runoff}), 0, T, S, agg, time_agg, d_out, ld_out,
                     "atl_runoff_convert_aggregate");
}

int atl_math_probe(atl_ctx *ctx, int fn, const double *d_in, int64_t n, double *d_out) {
    ATL_REQUIRE(ctx && d_in && d_out && n >= 0, "atl_math_probe: bad argument");
    ATL_REQUIRE(fn >= 0 && fn <= 5, "atl_math_probe: fn must be 0..5");
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
    hipLaunchKernelGGL(k_synth_pv, dim3(grid), dim3(256), 0, ctx->stream, *s, T, S, d_influx_direct,
                       d_influx_diffuse, d_influx_toa, d_albedo, d_temperature, d_solar_altitude,
                       d_solar_azimuth);
    return check_launch("atl_synth_pv_inputs");
}

}  // extern "C"
