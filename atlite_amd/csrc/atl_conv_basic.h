// Light converters: identity (generic aggregate_matrix), runoff, temperature family / COP, heat and
// cooling demand.  Reference: atlite/convert.py:292-418, 475-490, 1028-1034; aggregate.py:16-35.
// Part of libatlite_hip.so (gfx950); included through atl_kernel_templates.h by every kernel file, inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------------------
// converters
// ---------------------------------------------------------------------------------------
struct NoCell {};
struct NoCarry {};  // per-wave state a converter may keep across consecutive slots (none does at present)
template <class C>
__device__ __forceinline__ C carry_init() {
    return C{};
}

// generic dense cube (aggregate_matrix on an arbitrary converted DataArray)
struct IdentityConv {
    static constexpr bool kFlatSeries = true;  // per-cell series in flat order (k_cells_series_flat): the per-cell setup is (next to) nothing
    static constexpr bool kShiftOk = true;     // line-aligned plans may re-address the cube (S is the slot stride and nothing else)
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
    static constexpr bool kFlatSeries = true;  // per-cell series in flat order (k_cells_series_flat): the per-cell setup is (next to) nothing
    static constexpr bool kShiftOk = true;     // line-aligned plans may re-address the cube (S is the slot stride and nothing else)
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
    static constexpr bool kFlatSeries = true;  // per-cell series in flat order (k_cells_series_flat): the per-cell setup is (next to) nothing
    static constexpr bool kShiftOk = true;     // line-aligned plans may re-address the cube (S is the slot stride and nothing else)
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

