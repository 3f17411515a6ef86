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
#include <cstring>
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

#include "atl_device_util.h"
#include "atl_conv_basic.h"
#include "atl_conv_wind.h"
#include "atl_conv_pv.h"

// ---------------------------------------------------------------------------------------
// kernel 1: per-cell series  out[slot, cell]
// grid.x over 512-cell blocks, grid.y over slot chunks of kSeriesSlots
// ---------------------------------------------------------------------------------------
#ifndef ATL_SERIES_SLOTS
#define ATL_SERIES_SLOTS 32
#endif
constexpr int kSeriesSlots = ATL_SERIES_SLOTS;

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
            // lanes without a cell loaded a real cell's data (safe indices) and are masked by the store
            const double2 r = conv.compute(raw[g], true, true, cell, lds);
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
            const double2 r = conv.compute(raw[g], true, true, cell, lds);  // masked when psum / pcnt are stored
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
    out[c] = mean == 1 ? s / n : s;  // nan-skipping mean of nothing is NaN; nan-skipping sum is 0
    if (mean == 2) out[S + c] = n;   // ATL_TIME_SUM_COUNT: [sum | count]
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
            // structural zeros must not turn NaN/inf cells into NaN (scipy CSR skips them); same
            // expression as the unguarded path, so a row gives the same bits whichever path it takes
            const double t0 = a0 ? w.x * v[i].x : 0.0;
            c[i] = a1 ? __builtin_fma(w.y, v[i].y, t0) : t0;
        } else {
            c[i] = __builtin_fma(w.y, v[i].y, w.x * v[i].x);  // w = 0 where absent, v finite
        }
    }
    const double f = butterfly8(c);
    const int g = lane >> 3;
    if ((lane & 7) == 0 && sb + g < send) prow[sb + g] = f;
}

constexpr int kRowCache = ATL_ROW_CACHE;  // partial rows of a tile whose weights stay in registers

// register budget of the fused kernel: ATL_FUSED_WAVES waves per SIMD unless the converter asks for
// more registers (the general pv kernel is a long literal transcription and would spill)
template <class Conv, class = void>
struct conv_min_waves : std::integral_constant<int, ATL_FUSED_WAVES> {};
template <class Conv>
struct conv_min_waves<Conv, std::void_t<decltype(Conv::kMinWaves)>> : std::integral_constant<int, Conv::kMinWaves> {};
template <class Conv>
constexpr int min_waves() {
    return conv_min_waves<Conv>::value;
}

template <class Conv, bool VEC>
__global__ __launch_bounds__(256, min_waves<Conv>()) void k_fused_segred(Conv conv, PlanDev plan, int64_t slot0,
                                                      int64_t n_slots, int64_t S, int32_t chunk_slots,
                                                      int64_t n_units, double *__restrict__ partials,
                                                      int64_t ldp, int32_t conv_lds_doubles) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // per-wave LDS area behind the converter's tables: the weights of the tile's first kRowCache partial
    // rows (LDS instead of 4 VGPRs per row for the whole chunk: the register budget decides the occupancy)
    double *wlds = lds + conv_lds_doubles + (threadIdx.x >> 6) * (kRowCache * kSegCells);
#ifndef ATL_XCD_MAP
    // Linear order: consecutive blocks (= consecutive XCDs, block b runs on XCD b % 8) take consecutive
    // tile groups of the same time chunk, so the chip as a whole streams contiguous memory.
    const int64_t unit = int64_t(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    const int32_t seg = int32_t(unit % plan.n_segs);
    const int64_t chunk = unit / plan.n_segs;
#else
    // XCD-affine order (-DATL_XCD_MAP; measured: no gain - C2 3.44 vs 3.41 ms, C4 shard 6.12 vs
    // 6.13 ms, runoff 0.97 vs 0.97 ms): a group of 4 tiles always lands on the same XCD for every time
    // chunk, so its weights sit in one XCD's L2 instead of eight.  There is no reuse to win - the
    // weights are register-cached per 64-slot chunk and make up < 1 % of the traffic.
    const int64_t n_groups = (int64_t(plan.n_segs) + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t per_xcd = (n_groups + 7) / 8;
    const int64_t j = int64_t(blockIdx.x) >> 3;
    const int64_t group = int64_t(blockIdx.x & 7) + 8 * (j % per_xcd);
    const int64_t chunk = j / per_xcd;
    const int64_t seg64 = group * kWavesPerBlock + (threadIdx.x >> 6);
    if (group >= n_groups || seg64 >= plan.n_segs || chunk * chunk_slots >= n_slots) return;
    const int32_t seg = int32_t(seg64);
#endif
    // tile coordinates -> the lane's two adjacent cells (atl_internal.h: tile_lane_cells)
    const TileLane tl = tile_lane_cells(plan.X, plan.Y, plan.ntx, plan.w2_log2, seg, lane);
    const int64_t c0 = tl.c0;
    const bool v0 = tl.v0, v1 = tl.v1;
    const int64_t s0c = v0 ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices: loads never branch
    const int32_t p0 = plan.seg_ptr[seg], p1 = plan.seg_ptr[seg + 1];
    if (p0 == p1) return;  // no shape touches this tile: nothing to read
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    // weights of the first kRowCache partial rows: in this wave's LDS area for the whole chunk (each lane
    // writes and later reads only its own 16 bytes: no barrier needed)
    unsigned present = 0;  // bit 2r / 2r+1: cell 0 / 1 structurally present in row r
#pragma unroll
    for (int r = 0; r < kRowCache; ++r) {
        double2 wz = {0.0, 0.0};
        if (p0 + r < p1) {
            const double2 w = *reinterpret_cast<const double2 *>(plan.prow_w + int64_t(p0 + r) * kSegCells + 2 * lane);
            const bool a0 = !dnan(w.x), a1 = !dnan(w.y);
            wz.x = a0 ? w.x : 0.0;
            wz.y = a1 ? w.y : 0.0;
            present |= (a0 ? 1u : 0u) << (2 * r) | (a1 ? 1u : 0u) << (2 * r + 1);
        }
        *reinterpret_cast<double2 *>(wlds + r * kSegCells + 2 * lane) = wz;
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
                const double2 wr = *reinterpret_cast<const double2 *>(wlds + r * kSegCells + 2 * lane);
                if (all_finite)
                    reduce_row<false>(v, wr, true, true, lane, sb, send, prow);
                else
                    reduce_row<true>(v, wr, (present >> (2 * r)) & 1u, (present >> (2 * r + 1)) & 1u, lane, sb,
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
    if (threadIdx.x == 0) {
        out[blockIdx.x] = mean == 1 ? ss[0] / sn[0] : ss[0];
        if (mean == 2) out[gridDim.x + blockIdx.x] = sn[0];  // ATL_TIME_SUM_COUNT: [sum | count]
    }
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
    size_t slot = 0;
    explicit KernelBracket(atl_ctx *c) : ctx(c) {
        if (ctx->profiling) {
            slot = size_t(ctx->ring_count % int64_t(ctx->ev_ring.size() / 2));
            (void)hipEventRecord(ctx->ev_ring[2 * slot], ctx->stream);
        }
    }
    ~KernelBracket() {
        if (ctx->profiling) {
            (void)hipEventRecord(ctx->ev_ring[2 * slot + 1], ctx->stream);
            ++ctx->ring_count;
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
    if (const char *e = getenv("ATLITE_HIP_CHUNK")) {  // experiments: any multiple of kBatch
        const int64_t v = atoll(e);
        if (v >= kBatch && v % kBatch == 0) return int32_t(v);
    }
    const int64_t want_units = int64_t(ctx->n_cu) * 64;
    while (chunk > kBatch && n_segs * ((n_slots + chunk - 1) / chunk) < want_units) chunk /= 2;
    return int32_t(chunk);
}

template <class Conv>
int run_cells(atl_ctx *ctx, const Conv &conv, bool vec, size_t lds_bytes, int64_t n_slots, int64_t S,
              int time_agg, double *d_out, const char *what) {
    ATL_REQUIRE(time_agg >= ATL_TIME_NONE && time_agg <= ATL_TIME_SUM_COUNT, "%s: bad time_agg %d", what, time_agg);
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
                       n_chunks, S, time_agg == ATL_TIME_MEAN ? 1 : time_agg == ATL_TIME_SUM_COUNT ? 2 : 0, d_out);
    return check_launch(what);
}

template <class Conv>
int run_fused(atl_ctx *ctx, const Conv &conv, bool vec, size_t lds_bytes, int64_t n_slots, int64_t S,
              const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out, const char *what) {
    ATL_REQUIRE(agg, "%s: agg is NULL", what);
    ATL_REQUIRE(agg->ctx == ctx, "%s: aggregation plan belongs to another context", what);
    ATL_REQUIRE(agg->dev.n_cells == S, "%s: matrix has %lld columns but the cutout has %lld cells", what,
                (long long)agg->dev.n_cells, (long long)S);
    ATL_REQUIRE(time_agg >= ATL_TIME_NONE && time_agg <= ATL_TIME_SUM_COUNT, "%s: bad time_agg %d", what, time_agg);
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
#ifndef ATL_XCD_MAP
            const dim3 grid(unsigned((n_units + kWavesPerBlock - 1) / kWavesPerBlock));
#else
            const int64_t n_groups = (int64_t(plan.n_segs) + kWavesPerBlock - 1) / kWavesPerBlock;
            const dim3 grid(unsigned(8 * ((n_groups + 7) / 8) * n_chunks));
#endif
            // dynamic LDS: the converter's tables, then kRowCache weight rows per wave
            const size_t conv_lds = align_up(lds_bytes, 16);
            const size_t lds_total = conv_lds + size_t(kWavesPerBlock) * kRowCache * kSegCells * sizeof(double);
            const int32_t conv_lds_doubles = int32_t(conv_lds / sizeof(double));
            KernelBracket kb(ctx);
            if (vec)
                hipLaunchKernelGGL((k_fused_segred<Conv, true>), grid, dim3(256), lds_total, ctx->stream, conv, plan,
                                   w0, wn, S, chunk_slots, n_units, partials, ldp, conv_lds_doubles);
            else
                hipLaunchKernelGGL((k_fused_segred<Conv, false>), grid, dim3(256), lds_total, ctx->stream, conv,
                                   plan, w0, wn, S, chunk_slots, n_units, partials, ldp, conv_lds_doubles);
            if ((rc = check_launch(what))) return rc;
        }
        const dim3 grid(unsigned((wn + 255) / 256), unsigned(N));
        hipLaunchKernelGGL(k_combine, grid, dim3(256), 0, ctx->stream, plan, partials, ldp, wn, series + w0,
                           ld_series);
        if ((rc = check_launch(what))) return rc;
    }
    if (time_agg != ATL_TIME_NONE) {
        hipLaunchKernelGGL(k_rows_timered, dim3(unsigned(N)), dim3(256), 0, ctx->stream, series, ld_series,
                           n_slots, time_agg == ATL_TIME_MEAN ? 1 : time_agg == ATL_TIME_SUM_COUNT ? 2 : 0, d_out);
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
    c->k = pv_const_of(p);
    c->o.ss = sin(p->slope);
    c->o.cs = cos(p->slope);
    c->o.hp = (1.0 + c->o.cs) / 2.0;
    c->o.hm = (1.0 - c->o.cs) / 2.0;
    c->o.saz = p->azimuth;
    c->o.slope = p->slope;
    {
        const double sh = sin(p->slope / 2.0);
        c->o.sh3 = sh * sh * sh;
    }
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
    switch (p->tracking) {  // pv_needs_general() admits trackers only with stored angles, Huld, one orientation
        case ATL_TRACK_HORIZONTAL: return f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_HORIZONTAL>());
        case ATL_TRACK_TILTED_HORIZONTAL: return f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_TILTED_HORIZONTAL>());
        case ATL_TRACK_VERTICAL: return f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_VERTICAL>());
        case ATL_TRACK_DUAL: return f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_DUAL>());
        default: break;
    }
    if (p->trigon_model == ATL_TRIGON_OTHER)  // ... and Hay-Davies only with stored angles + Huld
        return pc ? f(PvConvT<false, true, false, kTailHuldHayDavies>()) : f(PvConvT<false, false, false, kTailHuldHayDavies>());
    if (p->panel_model == ATL_PANEL_SOLAR_THERMAL)
        return pc ? f(PvConvT<false, true, false, kTailThermal>()) : f(PvConvT<false, false, false, kTailThermal>());
    if (p->panel_model == ATL_PANEL_NONE)
        return pc ? f(PvConvT<false, true, false, kTailIrradiation>()) : f(PvConvT<false, false, false, kTailIrradiation>());
    if (sp) return pc ? f(PvConvT<true, true>()) : f(PvConvT<true, false>());
    if (p->night_skip && allow_skip) return pc ? f(PvConvT<false, true, true>()) : f(PvConvT<false, false, true>());
    return pc ? f(PvConvT<false, true>()) : f(PvConvT<false, false>());
}

// f(converter instance) with the PvxConvT instantiation for (tracking, trigon model)
template <class F>
int pvx_dispatch(const atl_pv_params *p, F &&f) {
    const bool other = p->trigon_model == ATL_TRIGON_OTHER;
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL:
            return other ? f(PvxConvT<ATL_TRACK_HORIZONTAL, ATL_TRIGON_OTHER>()) : f(PvxConvT<ATL_TRACK_HORIZONTAL, ATL_TRIGON_SIMPLE>());
        case ATL_TRACK_TILTED_HORIZONTAL:
            return other ? f(PvxConvT<ATL_TRACK_TILTED_HORIZONTAL, ATL_TRIGON_OTHER>())
                         : f(PvxConvT<ATL_TRACK_TILTED_HORIZONTAL, ATL_TRIGON_SIMPLE>());
        case ATL_TRACK_VERTICAL:
            return other ? f(PvxConvT<ATL_TRACK_VERTICAL, ATL_TRIGON_OTHER>()) : f(PvxConvT<ATL_TRACK_VERTICAL, ATL_TRIGON_SIMPLE>());
        case ATL_TRACK_DUAL:
            return other ? f(PvxConvT<ATL_TRACK_DUAL, ATL_TRIGON_OTHER>()) : f(PvxConvT<ATL_TRACK_DUAL, ATL_TRIGON_SIMPLE>());
        default:  // ATL_TRACK_NONE; out-of-range codes are rejected by make_pvx
            return other ? f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_OTHER>()) : f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_SIMPLE>());
    }
}

bool pv_needs_general(const atl_pv_inputs *in, const atl_pv_params *p) {
    if (in->d_influx != nullptr || in->d_albedo == nullptr) return true;
    if (p->tracking != ATL_TRACK_NONE)  // trackers: fast family for pv() with one orientation for the grid
        return !(p->tracking >= ATL_TRACK_HORIZONTAL && p->tracking <= ATL_TRACK_DUAL &&
                 p->trigon_model == ATL_TRIGON_SIMPLE && p->panel_model == ATL_PANEL_HULD &&
                 p->irradiation == ATL_IRR_TOTAL && in->d_solar_altitude != nullptr && in->d_temperature != nullptr &&
                 p->d_cell_slope == nullptr);
    if (p->trigon_model != ATL_TRIGON_SIMPLE)  // Hay-Davies: fast family only for pv() itself
        return !(p->trigon_model == ATL_TRIGON_OTHER && p->panel_model == ATL_PANEL_HULD &&
                 p->irradiation == ATL_IRR_TOTAL && in->d_solar_altitude != nullptr && in->d_temperature != nullptr);
    // fixed panel, simple trigon model, direct / diffuse / albedo cubes: the fast kernel family, with the
    // Huld panel, the solar thermal collector or the plain irradiation as its tail
    const bool stored = in->d_solar_altitude != nullptr, all7 = stored && in->d_temperature != nullptr;
    switch (p->panel_model) {
        case ATL_PANEL_HULD: return p->irradiation != ATL_IRR_TOTAL;
        case ATL_PANEL_SOLAR_THERMAL: return !(all7 && p->irradiation == ATL_IRR_TOTAL);
        case ATL_PANEL_NONE:
            return !(all7 && p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND);
        default: return true;  // bofinger
    }
}

template <class PVX>
int make_pvx(const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PVX *c, bool *vec) {
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
    c->k = pv_const_of(p);
    c->o = pvx_opt_of(p, in->d_influx != nullptr, in->d_albedo != nullptr);
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
    c->S = S;
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
    if (!finite || (g.method == ATL_WIND_LOG && !heights_ok)) return f(g);
    // unrolled knot search for the usual table sizes (make_wind pads to 16 / 32 / 128 knots)
    auto sized = [&](auto method) {
        constexpr int M = decltype(method)::value;
        if (g.inv_w > 0.0) return f(wind_as<M, kWindGrid>(g));
        if (g.n_pad == 16) return f(wind_as<M, 4>(g));
        if (g.n_pad == 32) return f(wind_as<M, 5>(g));
        if (g.n_pad == 128) return f(wind_as<M, 7>(g));
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
#ifdef ATL_NO_PV  // experimental builds (tools/build_variant.sh): leave a kernel family out to compile in seconds
    set_error("built without the pv kernels");
    return ATL_E_UNSUPPORTED;
#else
    if (pv_needs_general(in, p)) {
#ifdef ATL_NO_PVX
        set_error("built without the general pv kernel");
        return ATL_E_UNSUPPORTED;
#else
        return pvx_dispatch(p, [&](auto c) {
            int rc = make_pvx(in, p, T, S, &c, &vec);
            if (rc) return rc;
            return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
        });
#endif
    }
    return pv_dispatch(in, p, false, [&](auto c) {  // night skip: fused (aggregating) kernel only
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
    });
#endif
}

int atl_pv_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                             int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert_aggregate: ctx/inputs/params is NULL");
    bool vec;
#ifdef ATL_NO_PV
    set_error("built without the pv kernels");
    return ATL_E_UNSUPPORTED;
#else
    if (pv_needs_general(in, p)) {
#ifdef ATL_NO_PVX
        set_error("built without the general pv kernel");
        return ATL_E_UNSUPPORTED;
#else
        return pvx_dispatch(p, [&](auto c) {
            int rc = make_pvx(in, p, T, S, &c, &vec);
            if (rc) return rc;
            return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
        });
#endif
    }
    return pv_dispatch(in, p, true, [&](auto c) {
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
#endif
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
