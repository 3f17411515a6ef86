// Device helpers shared by all converters: numpy-compatible NaN semantics, unconditional nontemporal
// loads / stores of a lane's two cells, per-wave carry state.
// Part of libatlite_hip.so (gfx950); included through atl_kernel_templates.h by every kernel file, inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
ATL_HD __forceinline__ bool dnan(double x) { return x != x; }
ATL_HD __forceinline__ double fill0(double x) { return dnan(x) ? 0.0 : x; }
// numpy clip/maximum/minimum semantics: NaN in either operand propagates
ATL_HD __forceinline__ double np_max(double a, double b) { return (a > b || dnan(a)) ? a : b; }
ATL_HD __forceinline__ double np_min(double a, double b) { return (a < b || dnan(a)) ? a : b; }
ATL_HD __forceinline__ double np_clip(double x, double lo, double hi) {
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
#ifdef ATL_FLAT_LOADS  // experiment: generic-pointer loads (flat_load_*: count against LDS waits too)
    if constexpr (VEC) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 t = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + base + c0));
        r.x = t.x;
        r.y = t.y;
    } else {
        r.x = __builtin_nontemporal_load(p + base + c0);
        r.y = __builtin_nontemporal_load(p + base + c1);
    }
#else
    // The cubes live in device memory: tell the compiler (pointers inside the by-value converter
    // structs are generic to it), so it emits global_load_* instead of flat_load_* - flat loads also
    // tick the LDS counter, which makes every LDS table read of the wind kernel wait for them.
    typedef __attribute__((address_space(1))) const double gdouble;
    if constexpr (VEC) {
        // the pair as ONE 16-byte access that is only promised 8-byte alignment (global_load_dwordx4 takes it): with an
        // odd cell count every other slot's rows start 8 bytes off a 16-byte boundary
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        typedef f64x2 f64x2_a8 __attribute__((aligned(8)));
        typedef __attribute__((address_space(1))) const f64x2_a8 gf64x2;
        const f64x2 t = __builtin_nontemporal_load((gf64x2 *)(p + base + c0));
        r.x = t.x;
        r.y = t.y;
    } else {
        r.x = __builtin_nontemporal_load((gdouble *)(p + base + c0));
        r.y = __builtin_nontemporal_load((gdouble *)(p + base + c1));
    }
#endif
    return r;
}

// streaming store of a result cube that is not read again by this kernel
template <bool VEC>
__device__ __forceinline__ void st2(double *__restrict__ p, int64_t off, bool v0, bool v1, double2 v) {
    // nontemporal: +3 % on the 24 B/cell wind series (measured, C3)
#ifndef ATL_SPLIT_STORES
    if constexpr (VEC) {
        // one 16-byte store per lane: a wave writes whole 128-byte lines with a single instruction
        // instead of two half-filled ones (wind series 5.96-6.1 -> 5.9 ms, measured A/B); 8-byte alignment suffices
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        typedef f64x2 f64x2_a8 __attribute__((aligned(8)));
        typedef __attribute__((address_space(1))) f64x2_a8 gf64x2;
        f64x2 t;
        t.x = v.x;
        t.y = v.y;
        // (temporal stores instead: wind series 6.0 -> 6.4 ms, pv series 4.2 -> 4.4 ms)
#if defined(ATL_ST_POLICY) && defined(__HIP_DEVICE_COMPILE__)
        // experiment (tools/build_variant.sh): the cache policy bits of the result store spelled out.  nt keeps the line
        // in the XCD's L2 until it is evicted; sc1 / sc0 sc1 write through and drop it (MI355X_MICROARCH.md, store flavours)
        if (v0 && v1) {
            double *q = p + off;
#if ATL_ST_POLICY == 1
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(t) : "memory");
#elif ATL_ST_POLICY == 2
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(t) : "memory");
#elif ATL_ST_POLICY == 3
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(t) : "memory");
#elif ATL_ST_POLICY == 4
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(q), "v"(t) : "memory");
#else
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(q), "v"(t) : "memory");
#endif
        }
#else
        if (v0 && v1) __builtin_nontemporal_store(t, (gf64x2 *)(p + off));
#endif
        // the lane that owns the LAST cell of an odd cell count: its second value belongs to nobody (it would land on
        // the next slot's first cell)
        else if (v0) __builtin_nontemporal_store(v.x, p + off);
        // ... and the lane whose FIRST cell lies before the slot (blocks on a slot's own line grid, k_cells_series_flat)
        else if (v1) __builtin_nontemporal_store(v.y, p + off + 1);
        return;
    }
#endif
    if (v0) __builtin_nontemporal_store(v.x, p + off);
    if (VEC ? v0 : v1) __builtin_nontemporal_store(v.y, p + off + 1);
}

