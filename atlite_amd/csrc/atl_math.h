// Lean fp64 elementary functions for the streaming kernels (gfx950).
//
// ocml's sin/cos/log carry Payne-Hanek big-argument paths and double-double arithmetic
// (~600 VALU instructions per pv cell); the kernels only need ~1 ulp on physically ranged
// arguments, so these are plain Cody-Waite reductions + the classic fdlibm minimax
// polynomials (constants from FreeBSD msun k_sin.c / k_cos.c / e_log.c, (c) 1993 Sun
// Microsystems, "Permission to use, copy, modify, and distribute this software is freely
// granted, provided that this notice is preserved").
//
// Accuracy (measured against numpy on the GPU, tests/test_gpu_math.py): <= 2 ulp for
// sincos on |x| < 2^20, absolute error < 2.3e-16 up to |x| < 2^30; |x| >= 2^30 returns NaN
// (numpy would still return a value there; solar angles that large are not physical).
// log: <= 1 ulp on positive normal arguments; zero / subnormal / negative / inf / NaN go to
// libm's log (log_rare).
#pragma once
#include <hip/hip_runtime.h>

// Every routine is __host__ __device__: the kernels use the device instantiation (identical code to a
// device-only build), atl_math_probe_host() runs the SAME source on the CPU, so `pytest -m "not gpu"`
// checks the polynomials, reductions and special-case handling without a GPU.  Only the hardware
// seeds differ on the host: 1/b for v_rcp_f64 (the Newton steps still run), memcpy for the
// hi/lo-word intrinsics.
#define ATL_HD __host__ __device__

namespace atl {

ATL_HD __forceinline__ double rcp_seed(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(b);  // v_rcp_f64: ~2^-25 relative
#else
    return 1.0 / b;
#endif
}
ATL_HD __forceinline__ int dbl_hi(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2hiint(x);
#else
    unsigned long long u;
    __builtin_memcpy(&u, &x, 8);
    return int(u >> 32);
#endif
}
ATL_HD __forceinline__ int dbl_lo(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2loint(x);
#else
    unsigned long long u;
    __builtin_memcpy(&u, &x, 8);
    return int(u & 0xffffffffu);
#endif
}
ATL_HD __forceinline__ double dbl_make(int hi, int lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double(hi, lo);
#else
    const unsigned long long u = (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo);
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
#endif
}

ATL_HD __forceinline__ double fast_rcp(double b) {
    double y = rcp_seed(b);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    return y;
}

// sqrt(x) for 0 <= x < 2^500 (and NaN, -0): the hardware reciprocal-square-root seed (~2^-26), one coupled
// Newton step on (g ~ sqrt x, h ~ 1 / (2 sqrt x)) and a final residual correction - <= 1 ulp, 8 instructions; the
// compiler's IEEE expansion of sqrt() (pre-scaling for denormals and huge arguments, two steps) is ~20.
// the seed: v_rsq_f64 on the device; the host build (same-source tests) degrades the exact value to the ~2^-24 the
// hardware seed is good for, so that the CPU suite exercises the iterations' convergence, not the libm's
ATL_HD __forceinline__ double rsq_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    double y = 1.0 / __builtin_sqrt(x);
    uint64_t u;
    __builtin_memcpy(&u, &y, 8);
    u &= ~uint64_t(0x0FFFFFFF);  // keep 24 mantissa bits (inf / NaN / zero unchanged in kind)
    __builtin_memcpy(&y, &u, 8);
    return y;
#endif
}

ATL_HD __forceinline__ double lean_sqrt(double x) {
    const double y = rsq_seed(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? x : g;  // 0 * inf; negative and NaN arguments come out NaN through the seed
}

// sqrt(x) and 1 / sqrt(x) together for 2^-500 < x < 2^500: the same coupled iteration, then a Newton step on the
// reciprocal - both <= 1-2 ulp; replaces sqrt() + a division (or a reciprocal) in the trackers' closed forms
ATL_HD __forceinline__ double lean_sqrt_rsqrt(double x, double *rs) {
    const double y = rsq_seed(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    // the reciprocal half gets a Newton step of its own (1 - x y^2): refining it with h * g would only halve its error
    const double y2 = h + h;
    const double e = __builtin_fma(-(x * y2), y2, 1.0);
    *rs = __builtin_fma(0.5 * y2, e, y2);
    return g;
}

// a / b to ~1 ulp for normal, well-scaled operands (no denormal / overflow fix-ups)
ATL_HD __forceinline__ double fast_div(double a, double b) {
    const double y = fast_rcp(b);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// a / b through the reciprocal path when both operands sit comfortably inside the normal range,
// IEEE division (zeros, infinities, NaN, denormals) otherwise: <= 1 ulp from the IEEE quotient,
// identical special-case behaviour.  The slow branch is out of the common path of a wave.
ATL_HD __forceinline__ double guarded_div(double a, double b) {
    const double ab = __builtin_fabs(b);
    const bool ok = ab > 0x1.0p-400 && ab < 0x1.0p400 && __builtin_fabs(a) < 0x1.0p400;
    double r = fast_div(ok ? a : 0.0, ok ? b : 1.0);
    if (__builtin_expect(!ok, 0)) r = a / b;
    return r;
}

// reduce x to r in [-pi/4, pi/4] (+ tiny slack), quadrant in *q
ATL_HD __forceinline__ double reduce_pio2(double x, int *q) {
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);  // 2/pi
    // pi/2 split into three doubles; every FMA is exact before its single rounding
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    r = __builtin_fma(-k, -1.49738490485916983506e-33, r);
#if defined(__HIP_DEVICE_COMPILE__)
    // |x| >= 2^30 / NaN: the conversion is out of range (v_cvt_i32_f64 saturates), and every caller then selects
    // NaN on its own range test, never a value derived from q - an unselected operand cannot poison a select.
    // (-fno-strict-float-cast-overflow would pin this down formally; it costs the wind kernels 5-7 %.)
    *q = int(k) & 3;
#else
    *q = (k > -2147483648.0 && k < 2147483648.0) ? int(k) & 3 : 0;  // C++: out-of-range / NaN conversion is undefined
#endif
    return r;
}

ATL_HD __forceinline__ double poly_sin(double r, double z) {
    double p = 1.58969099521155010221e-10;
    p = __builtin_fma(p, z, -2.50507602534068634195e-08);
    p = __builtin_fma(p, z, 2.75573137070700676789e-06);
    p = __builtin_fma(p, z, -1.98412698298579493134e-04);
    p = __builtin_fma(p, z, 8.33333333332248946124e-03);
    p = __builtin_fma(p, z, -1.66666666666666324348e-01);
    return __builtin_fma(r * z, p, r);
}

ATL_HD __forceinline__ double poly_cos(double z) {
    double p = -1.13596475577881948265e-11;
    p = __builtin_fma(p, z, 2.08757232129817482790e-09);
    p = __builtin_fma(p, z, -2.75573143513906633035e-07);
    p = __builtin_fma(p, z, 2.48015872894767294178e-05);
    p = __builtin_fma(p, z, -1.38888888888741095749e-03);
    p = __builtin_fma(p, z, 4.16666666666666019037e-02);
    // 1 - z/2 + z^2 p, evaluated so that the leading terms stay exact
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * z * p);
}

ATL_HD __forceinline__ void lean_sincos(double x, double *s, double *c) {
    int q;
    const double r = reduce_pio2(x, &q);
    const double z = r * r;
    const double ps = poly_sin(r, z), pc = poly_cos(z);
    double ss = (q & 1) ? pc : ps;
    double cc = (q & 1) ? ps : pc;
    ss = (q & 2) ? -ss : ss;
    cc = ((q + 1) & 2) ? -cc : cc;
    const bool ok = __builtin_fabs(x) < 0x1.0p30;  // false for NaN / inf too
    *s = ok ? ss : __builtin_nan("");
    *c = ok ? cc : __builtin_nan("");
}

// the same without the |x| < 2^30 guard: callers that have checked the range themselves
ATL_HD __forceinline__ void sincos_core(double x, double *s, double *c) {
    int q;
    const double r = reduce_pio2(x, &q);
    const double z = r * r;
    const double ps = poly_sin(r, z), pc = poly_cos(z);
    const double ss = (q & 1) ? pc : ps;
    const double cc = (q & 1) ? ps : pc;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
ATL_HD __forceinline__ double cos_core(double x) {
    int q;
    const double r = reduce_pio2(x, &q);
    const double z = r * r;
    // both polynomials, one select: a branch on the quadrant diverges in every wave and costs the same
    const double ps = poly_sin(r, z), pc = poly_cos(z);
    const double cc = (q & 1) ? ps : pc;
    return ((q + 1) & 2) ? -cc : cc;
}
ATL_HD __forceinline__ double lean_cos(double x) {
    int q;
    const double r = reduce_pio2(x, &q);
    const double z = r * r;
    double cc = (q & 1) ? poly_sin(r, z) : poly_cos(z);
    cc = ((q + 1) & 2) ? -cc : cc;
    return __builtin_fabs(x) < 0x1.0p30 ? cc : __builtin_nan("");
}

ATL_HD __forceinline__ double lean_sin(double x) {
    int q;
    const double r = reduce_pio2(x, &q);
    const double z = r * r;
    double ss = (q & 1) ? poly_cos(z) : poly_sin(r, z);
    ss = (q & 2) ? -ss : ss;
    return __builtin_fabs(x) < 0x1.0p30 ? ss : __builtin_nan("");
}

// log of a positive, normal, finite double: no special cases, integer exponent split
ATL_HD __forceinline__ double log_core(double x) {
    // x = 2^e * m, m in [sqrt(1/2), sqrt(2)): shift the mantissa window by sqrt(2)/2 like fdlibm
    unsigned hi = unsigned(dbl_hi(x));
    const unsigned lo = unsigned(dbl_lo(x));
    hi += 0x3ff00000u - 0x3fe6a09eu;
    const int e = int(hi >> 20) - 0x3ff;
    hi = (hi & 0x000fffffu) + 0x3fe6a09eu;
    const double m = dbl_make(int(hi), int(lo));
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);
    const double z = s * s, w = z * z;
    double t1 = __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);
    t1 = __builtin_fma(w, t1, 3.999999999940941908e-01) * w;
    double t2 = __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);
    t2 = __builtin_fma(w, t2, 2.857142874366239149e-01);
    t2 = __builtin_fma(w, t2, 6.666666666666735130e-01) * z;
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = double(e);
    return dk * 6.93147180369123816490e-01 -
           ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// ---- table-driven log (for kernels that already use LDS) ------------------------------------------
// x = 2^e * m, m in [sqrt(1/2), sqrt(2)); c = round(128 m)/128 is exact in fp64 and the table holds
// {fl(1/c), fl(log c)} for c = 90/128 .. 182/128 (c = 1: {1, 0} exactly, so arguments near 1 keep full
// RELATIVE accuracy); r = m/c - 1, |r| <= 1/180, log1p(r) by a degree-7 Taylor polynomial
// (truncation < 4e-18).  No reciprocal, ~17 VALU slots + one 16-byte LDS read.  <= 2 ulp measured.
constexpr int kLogTabLo = 90, kLogTabHi = 182, kLogTabN = kLogTabHi - kLogTabLo + 1;  // 93 entries

// fills tab[2*kLogTabN] (LDS); call from all threads of the block, then __syncthreads()
ATL_HD __forceinline__ void log_table_entry(double *tab, int i) {
    const double c = double(kLogTabLo + i) * 0x1.0p-7;
    tab[2 * i] = 1.0 / c;  // IEEE division, once per block
    tab[2 * i + 1] = (kLogTabLo + i == 128) ? 0.0 : log_core(c);
}
__device__ __forceinline__ void log_table_init(double *tab) {
    for (int i = threadIdx.x; i < kLogTabN; i += blockDim.x) log_table_entry(tab, i);
}

ATL_HD __forceinline__ double log_core_tab(double x, const double *tab) {
    unsigned hi = unsigned(dbl_hi(x));
    const unsigned lo = unsigned(dbl_lo(x));
    hi += 0x3ff00000u - 0x3fe6a09eu;
    const int e = int(hi >> 20) - 0x3ff;
    hi = (hi & 0x000fffffu) + 0x3fe6a09eu;
    const double m = dbl_make(int(hi), int(lo));           // [sqrt(1/2), sqrt(2))
    const double rm = __builtin_rint(m * 128.0);
    const int i = int(rm) - kLogTabLo;                               // 0 .. kLogTabN-1
    const double2 t = *reinterpret_cast<const double2 *>(tab + 2 * i);
    // m - c is exact (Sterbenz), so r carries only the RELATIVE error of 1/c: bins next to c = 1,
    // where log(m) is small, stay accurate
    const double r = (m - rm * 0x1.0p-7) * t.x;
    double p = 1.0 / 7.0;
    p = __builtin_fma(p, r, -1.0 / 6.0);
    p = __builtin_fma(p, r, 0.2);
    p = __builtin_fma(p, r, -0.25);
    p = __builtin_fma(p, r, 1.0 / 3.0);
    p = __builtin_fma(p, r, -0.5);
    const double l1p = __builtin_fma(p * r, r, r);                   // r - r^2/2 + ...
    const double dk = double(e);
    return __builtin_fma(dk, 6.93147180369123816490e-01, t.y + (__builtin_fma(dk, 1.90821492927058770002e-10, l1p)));
}

// zero, subnormal, negative, inf, NaN: full libm, kept out of line so that the hot loops carry
// only a never-taken branch
ATL_HD inline __noinline__ double log_rare(double x) { return log(x); }

ATL_HD __forceinline__ double lean_log(double x) {
    const bool ok = x >= 0x1.0p-1022 && x < __builtin_inf();
    double r = log_core(ok ? x : 1.0);
    if (__builtin_expect(!ok, 0)) r = log_rare(x);
    return r;
}

// ---- np.interp on a padded knot table (wind power curves) ---------------------------------------------
// Table layout (built on the host by wind_table_build, atl_internal.h): V[n_pad] knots padded with +inf,
// then K[n_pad][4] records {V[j], F[j], slope[j], 0}.  For a FINITE table
//   xc = clamp(x, V[0], V[n-1]);  j = largest index with V[j] <= xc;  r = fma(slope[j], xc - V[j], F[j])
// reproduces numpy's arr_interp exactly: F[j] on knots, F[0] / F[n-1] outside the range (+-inf too), NaN
// for NaN, the upper one of repeated knots.  STEPS > 0: the table has exactly 2^STEPS entries and the
// search is STEPS unrolled probes; STEPS = 0: run-time size.
template <int STEPS>
ATL_HD __forceinline__ double interp_padded(const double *tab, int n_knots, int n_pad, double x) {
    const double *V = tab;
    const double *K = tab + n_pad;
    const double vmin = V[0], vmax = V[n_knots - 1];
    double xc = x > vmax ? vmax : x;
    xc = xc < vmin ? vmin : xc;  // NaN stays NaN
    int j = 0;
    if constexpr (STEPS > 0) {
#pragma unroll
        for (int s = STEPS - 1; s >= 0; --s) {
            const int cand = j + (1 << s);
            j = (V[cand] <= xc) ? cand : j;
        }
    } else {
        for (int step = n_pad >> 1; step > 0; step >>= 1) {
            const int cand = j + step;
            j = (V[cand] <= xc) ? cand : j;
        }
    }
    const double2 k0 = *reinterpret_cast<const double2 *>(K + 4 * j);
    const double sl = K[4 * j + 2];
    return __builtin_fma(sl, xc - k0.x, k0.y);
}

// GRID-ALIGNED knot tables (every knot a multiple of w = 2^-k, V[0] >= 0, finite; built on the host by
// wind_grid_build, atl_internal.h): one record {V[j], F[j], slope[j], 0} per bucket [b w, (b+1) w), j = the
// largest index with V[j] <= b w.  Since no knot lies strictly inside a bucket that j is also the largest
// index with V[j] <= xc for every xc of the bucket, and xc * inv_w is exact (inv_w is a power of two), so
// the lookup selects the same interval as the search above and returns the same bits - with one LDS round
// trip instead of STEPS dependent ones (most shipped power curves have integer or half-integer knots).
ATL_HD __forceinline__ double interp_grid(const double *tab, double vmin, double vmax, double inv_w, int b0, double x) {
    // clamp with min / max (they drop a NaN operand, so xc is always a number and the bucket index is always
    // inside the table) and put the NaN back at the end: 5 instructions instead of 9 for the select-based clamp
    const double xc = __builtin_fmax(__builtin_fmin(x, vmax), vmin);
    const int b = int(xc * inv_w);  // exact product, >= 0: truncation is floor
    const double *rec = (tab - 4 * b0) + 4 * b;  // (tab - 4 b0) is loop-invariant
    const double2 k0 = *reinterpret_cast<const double2 *>(rec);
    const double r = __builtin_fma(rec[2], xc - k0.x, k0.y);
    return x != x ? x : r;
}

// literal numpy/_core/src/multiarray/compiled_base.c arr_interp, for tables that hold non-finite values
// (same table layout; the slope is formed on the fly like numpy does when it has not precomputed it)
ATL_HD __forceinline__ double interp_literal(const double *tab, int n_knots, int n_pad, double x) {
    const double *V = tab;
    const double *K = tab + n_pad;
    const int n = n_knots;
    if (x != x) return x;
    if (x < V[0]) return K[1];
    if (x > V[n - 1]) return K[4 * (n - 1) + 1];
    int j = 0;
    for (int step = n_pad >> 1; step > 0; step >>= 1) {
        const int cand = j + step;
        if (cand < n && V[cand] <= x) j = cand;  // cand < n: a real +inf knot must not run into the +inf padding
    }
    const double xj = K[4 * j], fj = K[4 * j + 1];
    if (j == n - 1) return fj;
    if (xj == x) return fj;
    const double slope = (K[4 * (j + 1) + 1] - fj) / (K[4 * (j + 1)] - xj);
    double r = slope * (x - xj) + fj;
    if (r != r) {
        r = slope * (x - K[4 * (j + 1)]) + K[4 * (j + 1) + 1];
        if (r != r && fj == K[4 * (j + 1) + 1]) r = fj;
    }
    return r;
}

}  // namespace atl
