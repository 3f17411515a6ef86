// Wind converter: hub-height extrapolation + power-curve interpolation.
// Reference: atlite/wind.py:76-112, atlite/convert.py:634-662 (np.interp).
// Part of libatlite_hip.so (gfx950); included by atl_kernels.hip inside its anonymous namespace.
#pragma once

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
// STEPS > 0: the table is padded to exactly 2^STEPS knots and the search is STEPS unrolled,
// branch-free probes, so the searches of a group's 8 cells interleave and hide each other's LDS
// latency; STEPS = 0 searches a table of run-time size in a loop (13 instructions and one exposed LDS
// round trip per probe - measured 52 of the ~125 instructions per cell before the specialisation).
// STEPS = kWindGrid: grid-aligned knots (integer / half-integer ... wind speeds, which is what nearly every
// shipped power curve has): the interval comes from one bucket lookup, no search (interp_grid, atl_math.h).
constexpr int kWindGrid = -1;
// STEPS = kWindIdentity: no power curve at all (atl_wind_params.n_knots == 0): the converter's output is the
// extrapolated wind speed itself - atlite.wind.extrapolate_wind_speed (wind.py:76-112) as an operation of its own
constexpr int kWindIdentity = -2;
// Out-of-line rare paths are FREE functions taking scalars by value: a __noinline__ member function
// needs `this`, i.e. the whole converter struct spilled to a scratch frame by every thread at kernel
// entry (80 B/lane of extra HBM writes in round 1's wind kernels).
// wind.py:99-101 / :111, literally; used for the rare arguments the fast path excludes
ATL_HD inline __noinline__ double wind_hub_speed_literal(int method, double to_height, double from_height, double v,
                                                         double z) {
    if (method == ATL_WIND_LOG) return v * (log(to_height / z) / log(from_height / z));
    if (method == ATL_WIND_POWER) return v * pow(to_height / from_height, z);
    return v;
}
// literal numpy arr_interp (any table): atl_math.h, shared with the host probe
ATL_HD inline __noinline__ double wind_interp_generic(const double *lds, int n_knots, int n_pad, double x) {
    return interp_literal(lds, n_knots, n_pad, x);
}

template <int METHOD, int STEPS = 0>
struct WindConvT {
    const double *wnd;
    const double *aux;
    int64_t S;
    int aux_static;
    int method;
    double to_height, from_height;
    double log_ratio;      // log(to/from)   (power law)
    const double *table;   // device: V[n_pad] | K[n_pad][4]   (grid mode: G[n_buckets][4])
    int n_knots, n_pad;
    int tab_doubles;       // size of the interpolation table in LDS (the log table follows it)
    double vmin, vmax, inv_w;  // grid mode: first / last knot, buckets per unit wind speed
    int b0;                    // grid mode: bucket index of the first knot
    struct Cell {
        double2 aux;
        double lh, lf;  // lean_log(to_height), lean_log(from_height)
    };
    __device__ void block_init(double *lds) const {
        for (int i = threadIdx.x; i < tab_doubles; i += blockDim.x) lds[i] = table[i];
        if constexpr (METHOD == ATL_WIND_LOG) log_table_init(lds + tab_doubles);
    }
    // cell_setup in two parts (k_cells_series_flat loads the cubes between them): the static roughness / shear needs no table
    static constexpr bool kEarlyLoad = true;
    ATL_HD Cell cell_early(int64_t c0, bool v0, bool v1) const {
        Cell c;
        c.aux.x = 0.0;
        c.aux.y = 0.0;
        if (method != ATL_WIND_NONE && aux_static) {
            c.aux.x = v0 ? aux[c0] : 1.0;
            c.aux.y = v1 ? aux[c0 + 1] : 1.0;
        }
        c.lh = 0.0;
        c.lf = 0.0;
        return c;
    }
    ATL_HD void cell_finish(Cell &c, const double *lds) const {
        // log(from) through the same routine as the per-cell roughness: z0 == from_height gives an
        // exact zero denominator, like the reference's log(from/z0) = log(1)
        if constexpr (METHOD == ATL_WIND_LOG) c.lf = log_core_tab(from_height, lds + tab_doubles);
    }
    ATL_HD Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c = cell_early(c0, v0, v1);
        cell_finish(c, lds);
        return c;
    }
    ATL_HD __forceinline__ double hub_speed_literal(double v, double z) const {
        return wind_hub_speed_literal(method, to_height, from_height, v, z);
    }
    // fast path: *rare is set when the literal formula must be used instead
    ATL_HD __forceinline__ double hub_speed_fast(double v, double z, const Cell &c, bool *rare,
                                                     const double *lds) const {
        if constexpr (METHOD == ATL_WIND_LOG) {
            // v * (log(to/z0) / log(from/z0))  =  v * (1 + log(to/from) / (log(from) - log(z0))):
            // one table-driven log and one reciprocal per cell.  log(from) goes through the same
            // routine, so z0 == from gives den == 0 exactly (-> rare path -> the literal formula).
            const double *ltab = lds + tab_doubles;
            const bool zok = z >= 0x1.0p-1022 && z < __builtin_inf();
            // no substitute for a bad z: log_core_tab builds its mantissa (and with it the table index) from
            // the fraction bits alone, so any bit pattern reads inside the table; the value is discarded (rare)
            const double lz = log_core_tab(z, ltab);
            const double den = c.lf - lz;
            // |den| < 1500 always.  Both logs carry ~1e-15 of absolute error: below 2^-20 the difference would
            // lose the 1e-10 the result must keep, so roughness within 1e-6 (relative) of the source height
            // takes the literal formula like the exactly-zero denominator does.
            const bool tame = __builtin_fabs(den) > 0x1.0p-20;
            *rare = !(zok && tame);
            // one Newton step on v_rcp_f64 (2^-25 -> ~2^-50): 1e-15 on a factor of order one
            double y = rcp_seed(den);
            y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
            return v * __builtin_fma(log_ratio, y, 1.0);
        } else if constexpr (METHOD == ATL_WIND_POWER) {
            *rare = false;
            return v * exp(z * log_ratio);  // v * (to/from) ** shear
        } else {
            *rare = false;
            return v;
        }
    }
    ATL_HD __forceinline__ double interp(double x, const double *lds) const {  // atl_math.h (shared with the host probe)
        if constexpr (STEPS == kWindIdentity)
            return x;
        else if constexpr (STEPS == kWindGrid)
            return interp_grid(lds, vmin, vmax, inv_w, b0, x);
        else
            return interp_padded<STEPS>(lds, n_knots, n_pad, x);
    }
    ATL_HD __forceinline__ double interp_generic(double x, const double *lds) const {
        return wind_interp_generic(lds, n_knots, n_pad, x);
    }
#ifndef ATL_WIND_GROUP
#define ATL_WIND_GROUP 4
#endif
    static constexpr int kGroup = ATL_WIND_GROUP;
    static constexpr int kMinChunk = 64;  // fused kernel: C3 aggregated 3.70 / 3.77 / 3.96 ms with chunks of 64 / 32 / 16 slots
    static constexpr int kCubes = 2;
    static constexpr bool kFlatSeries = true;  // per-cell series in flat order (k_cells_series_flat)
    static constexpr bool kShiftOk = true;     // line-aligned plans may re-address the cubes (S is the slot stride and nothing else)
    // dense tiles: the log-law converter has no 64 VGPRs to spare for a resident operand image at two waves per SIMD
    static constexpr int kDenseResident = (METHOD == ATL_WIND_LOG || (METHOD == ATL_WIND_POWER && STEPS == 0)) ? 0 : 1;
#ifdef ATL_WIND_WAVES
    static constexpr int kMinWaves = ATL_WIND_WAVES;
#endif
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
    ATL_HD __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c,
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
            r.x = interp(h0, lds);
            r.y = interp(h1, lds);
        }
        r.x = v0 ? r.x : 0.0;
        r.y = v1 ? r.y : 0.0;
        return r;
    }
};

