// Kernel templates and launch plumbing shared by the translation units of libatlite_hip.so
// (atl_kernels.hip: wind / heat / runoff / thermo / spmm + tooling; atl_kernels_pv.hip: the fast pv
// family; atl_kernels_pvx.hip: the general pv kernel).  Everything here lives in the including file's
// anonymous namespace: the build compiles the three units in parallel.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>

#include "atl_internal.h"
#include "atl_math.h"

// tuning knobs of the fused kernel (A/B-tested with variants built by tools/build_variant.sh, loaded through $ATLITE_HIP_LIB)
#ifndef ATL_FUSED_WAVES
#define ATL_FUSED_WAVES 3  // waves per SIMD the register allocation must allow (<= 168 VGPRs)
#endif
#ifndef ATL_PV_GROUP
#define ATL_PV_GROUP 1
#endif
#ifndef ATL_ROW_CACHE
#define ATL_ROW_CACHE 8
#endif


using namespace atl;

namespace {

#include "atl_device_util.h"
#include "atl_conv_basic.h"

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
#ifndef ATL_SERIES_GROUP
#define ATL_SERIES_GROUP 4
#endif
    constexpr int G = Conv::kGroup >= ATL_SERIES_GROUP ? ATL_SERIES_GROUP : Conv::kGroup;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    for (int64_t sg = s0; sg < s1; sg += G) {
        typename Conv::Raw raw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) raw[g] = conv.template load<VEC>(min(sg + g, s1 - 1), g, s0c, s1c, cell, carry);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // lanes without a cell loaded a real cell's data (safe indices) and are masked by the store; the converter is
            // told which of the pair's cells exist, so that what lies beside the last cell (the next slot, slot padding)
            // cannot steer the pair onto a converter's rare path
            const double2 r = conv.compute(raw[g], v0, v1, cell, lds);
            if (sg + g < s1) st2<VEC>(out, (sg + g) * S + c0, v0, v1, r);
        }
    }
}

// ---------------------------------------------------------------------------------------
// kernel 1f: per-cell series in FLAT order (round 4).  A block converts kFlatChunks x 512 consecutive cells of ONE slot,
// blockIdx.x fastest: the chip as a whole walks every cube front to back instead of every block walking its cells
// through 32 slots a whole slot apart.  For a stream that also WRITES a cube that order is worth 10-20 % on this chip
// (tools/probes/mix_probe.hip: c = a + b over 11 GB cubes, 5.18 ms flat against 6.0-6.3 ms slot-walking; read-only
// streams do not care).  For converters whose per-cell setup is trivial (kFlatSeries): it is redone per cell and slot here.
// ---------------------------------------------------------------------------------------
template <class Conv, class = void>
struct conv_flat_series : std::false_type {};
template <class Conv>
struct conv_flat_series<Conv, std::void_t<decltype(Conv::kFlatSeries)>> : std::integral_constant<bool, Conv::kFlatSeries> {};

// converters whose per-cell setup splits into a part that needs no LDS (cell_early) and one that reads the block's tables
// (cell_finish): the flat kernel issues the cubes' loads BEFORE it fills the tables, so both latencies overlap
template <class Conv, class = void>
struct conv_early_load : std::false_type {};
template <class Conv>
struct conv_early_load<Conv, std::void_t<decltype(Conv::kEarlyLoad)>> : std::integral_constant<bool, Conv::kEarlyLoad> {};

// (Round 6 tried blocks of 2 / 4 / 8 x 512 cells - all loads of a block issued before its LDS tables are built, the tables paid for
//  once per 2-8 x 12 KiB of traffic: C3 wind series 5.58 ms with 512 cells per block, 5.66 / 5.91 / 5.70 ms with 1024 / 2048 / 4096
//  on one box (gpurun_out/r06_m).  The order in which the chip sweeps the cubes matters more than the table build; removed.)
template <class Conv>
__global__ __launch_bounds__(256) void k_cells_series_flat(Conv conv, int64_t S, uint32_t n_chunks, double *__restrict__ out, int32_t shift,
                                                           int64_t slot0) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t slot_u = blockIdx.x / n_chunks;  // blocks in (slot, chunk) order: every stream advances front to back
    const int64_t slot = slot0 + slot_u;            // (slot0: the first slot of this launch's range)
    // slots that do not start on a 128-byte line (the converter's slot stride conv.S is not a multiple of 16 cells: a
    // caller's contiguous cubes on an odd grid): the chunks of THIS slot start o cells early, on its line grid - every
    // lane's 16 bytes aligned, every wave's KiB eight whole lines (slot 0 has o = 0: nothing is read before the cube)
    const int64_t o = shift ? (slot * conv.S) & 15 : 0;
    const int64_t c0 = int64_t(blockIdx.x - slot_u * n_chunks) * 512 + int64_t(threadIdx.x) * 2 - o;
    const bool v0 = c0 >= 0 && c0 < S, v1 = c0 + 1 >= 0 && c0 + 1 < S;
    const int64_t s0c = (v0 || v1) ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    typename Conv::Raw raw;
    typename Conv::Cell cell;
    if constexpr (conv_early_load<Conv>::value) {
        // (s_setprio 3 around these two lines - the waves that still have to issue their loads win arbitration over those building
        //  tables - changed nothing measurable: round 6, gpurun_out/r06_final1)
        cell = conv.cell_early(c0, v0, v1);
        raw = conv.template load<true>(slot, 0, s0c, s1c, cell, carry);
        conv.block_init(lds);
        __syncthreads();
        conv.cell_finish(cell, lds);
    } else {
        conv.block_init(lds);
        __syncthreads();
        cell = conv.cell_setup(c0, v0, v1, lds);
        raw = conv.template load<true>(slot, 0, s0c, s1c, cell, carry);
    }
    const double2 r = conv.compute(raw, v0, v1, cell, lds);
    st2<true>(out, slot * S + c0, v0, v1, r);
}

// (Round 5 tried the same walk with a PERSISTENT grid - n_CU x 8 or 16 blocks taking units b, b + G, ..., the converter's LDS
//  tables built once per block, the next unit's loads in flight during the conversion: 7.0 / 6.8 ms against this kernel's
//  5.6 ms on C3, gpurun_out/r05_d.  One short-lived block per unit lets the hardware's dispatcher keep the sweep tight;
//  removed again.)
// ---------------------------------------------------------------------------------------
// kernel 2: per-cell time reduction.  psum/pcnt[chunk, cell] then k_chunk_reduce.
// ---------------------------------------------------------------------------------------
// Alignment classes (p > 1; cubes whose slots do not start on 128-byte lines - a caller's contiguous cubes, S % 16 != 0; the
// per-cell counterpart of a line-aligned plan): chunk blockIdx.y belongs to class r = blockIdx.y % p and walks the slots
// r, r + p, r + 2 p, ... of its range, which all start o_r = (r * stride) % 16 cells into a line - so the block's cells
// start o_r cells early, on THAT line grid, for the whole walk.  psum / pcnt keep one row per chunk; every cell is in every
// class, so k_chunk_reduce needs nothing new.
struct ClassWalk {
    int64_t o;        // cells the class's slots start into their line
    int64_t first;    // the class's first slot (r)
    int64_t step;     // p
    int64_t s0, s1;   // this chunk's range of the class's slots (indices i: slot = first + step * i)
};
__device__ __forceinline__ ClassWalk class_walk(int32_t p, int64_t stride, int64_t n_slots, int64_t chunk_len) {
    ClassWalk w;
    const int64_t r = p > 1 ? int64_t(blockIdx.y % unsigned(p)) : 0, j = p > 1 ? int64_t(blockIdx.y / unsigned(p)) : int64_t(blockIdx.y);
    w.o = p > 1 ? (r * stride) & 15 : 0;
    w.first = r;
    w.step = p > 1 ? p : 1;
    const int64_t n = p > 1 ? (n_slots - r + p - 1) / p : n_slots;  // slots of the class
    w.s0 = j * chunk_len;
    w.s1 = min(w.s0 + chunk_len, n);
    return w;
}

template <class Conv, bool VEC>
__global__ __launch_bounds__(256) void k_cells_timered(Conv conv, int64_t n_slots, int64_t S,
                                                       int64_t chunk_len, double *__restrict__ psum,
                                                       double *__restrict__ pcnt, int32_t classes) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const ClassWalk w = class_walk(classes, conv.S, n_slots, chunk_len);
    const int64_t c0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 2 - w.o;
    const bool v0 = c0 >= 0 && c0 < S, v1 = c0 + 1 >= 0 && c0 + 1 < S;
    const int64_t s0c = (v0 || v1) ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);  // safe indices
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const int64_t s0 = w.s0, s1 = w.s1;
    double2 acc = {0.0, 0.0}, cnt = {0.0, 0.0};
    constexpr int G = Conv::kGroup >= 4 ? 4 : Conv::kGroup;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    for (int64_t sg = s0; sg < s1; sg += G) {
        typename Conv::Raw raw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) raw[g] = conv.template load<VEC>(w.first + w.step * min(sg + g, s1 - 1), g, s0c, s1c, cell, carry);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double2 r = conv.compute(raw[g], v0, v1, cell, lds);  // (invalid cells: masked when psum / pcnt are stored)
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

// converter traits of the early-out kernels (k_cells_night, k_fused_segred_night) and of the MFMA-carrying instantiation
template <class Conv, class = void>
struct conv_min_waves : std::integral_constant<int, ATL_FUSED_WAVES> {};
template <class Conv>
struct conv_min_waves<Conv, std::void_t<decltype(Conv::kMinWaves)>> : std::integral_constant<int, Conv::kMinWaves> {};
template <class Conv>
constexpr int min_waves() {
    return conv_min_waves<Conv>::value;
}

// ... of the per-cell early-out kernel (k_cells_night), where a converter asks for a budget of its own (kMinWavesCells)
template <class Conv, class = void>
struct conv_min_waves_cells : std::integral_constant<int, conv_min_waves<Conv>::value> {};
template <class Conv>
struct conv_min_waves_cells<Conv, std::void_t<decltype(Conv::kMinWavesCells)>> : std::integral_constant<int, Conv::kMinWavesCells> {};

// shortest chunk of slots a wave of the fused kernel walks (kMinChunk; default: two batches).  A converter whose
// per-wave setup is heavy asks for longer ones.
template <class Conv, class = void>
struct conv_min_chunk : std::integral_constant<int, 16> {};
template <class Conv>
struct conv_min_chunk<Conv, std::void_t<decltype(Conv::kMinChunk)>> : std::integral_constant<int, Conv::kMinChunk> {};

// cubes a converter streams per slot (kCubes; default 1): with the partial rows per tile it decides how short a chunk may
// get - every unit re-reads its tile's weight rows (1 KiB each), which a light converter does not amortise over two batches
template <class Conv, class = void>
struct conv_cubes : std::integral_constant<int, 1> {};
template <class Conv>
struct conv_cubes<Conv, std::void_t<decltype(Conv::kCubes)>> : std::integral_constant<int, Conv::kCubes> {};

// MFMA groups (16 partial rows each) of a dense tile whose operand image a wave keeps IN REGISTERS for its whole chunk
// (kDenseResident; default 1 = 64 VGPRs; two groups = 128 VGPRs spill at the two waves per SIMD the kernel is built for).  Until round 4 every 16-slot sweep
// re-read the image from L2 - 16 KiB per group against the 16 KiB of cube data a one-cube converter streams per sweep.
template <class Conv, class = void>
struct conv_dense_resident : std::integral_constant<int, 1> {};
template <class Conv>
struct conv_dense_resident<Conv, std::void_t<decltype(Conv::kDenseResident)>> : std::integral_constant<int, Conv::kDenseResident> {};
template <class Conv, class = void>
struct conv_dense_ok : std::true_type {};
template <class Conv>
struct conv_dense_ok<Conv, std::void_t<decltype(Conv::kDenseOk)>> : std::integral_constant<bool, Conv::kDenseOk> {};
// converters that exist for vectorised launches only (kVecOnly): run_cells / run_fused answer kNeedScalar when a
// launch cannot be vectorised (odd cell count or row length, unaligned cubes) and the caller takes its generic
// fallback (the pv family's rarely used members -> the general pv kernel)
template <class Conv, class = void>
struct conv_vec_only : std::false_type {};
template <class Conv>
struct conv_vec_only<Conv, std::void_t<decltype(Conv::kVecOnly)>> : std::integral_constant<bool, Conv::kVecOnly> {};
constexpr int kNeedScalar = 1;  // internal status, never returned through the C ABI
// converters whose fused kernel parks the converted values of a batch in the wave's LDS rows instead of registers
// (kStageValues): 32 VGPRs less at the conversion's peak - what lets the in-kernel solar position run 4 waves per SIMD
// without scratch; the weight cache shrinks to kRowCacheNight rows so that four workgroups still share a CU's 160 KiB
template <class Conv, class = void>
struct conv_stage_values : std::false_type {};
template <class Conv>
struct conv_stage_values<Conv, std::void_t<decltype(Conv::kStageValues)>> : std::integral_constant<bool, Conv::kStageValues> {};
// converters whose early-out can read a precomputed DAY MAP instead of loading and voting on the keys (Conv::kDayMap;
// the map pointer travels in conv.in.d_day_map: k_day_map / k_fused_segred_night<..., MAP = true>)
template <class Conv, class = void>
struct conv_day_map : std::false_type {};
template <class Conv>
struct conv_day_map<Conv, std::void_t<decltype(Conv::kDayMap)>> : std::integral_constant<bool, Conv::kDayMap> {};
template <class Conv, class = void>
struct conv_night_pipe : std::false_type {};
template <class Conv>
struct conv_night_pipe<Conv, std::void_t<decltype(Conv::kNightPipe)>> : std::integral_constant<bool, Conv::kNightPipe> {};

// ---------------------------------------------------------------------------------------
// kernels 1b / 2b: per-cell series / time reduction with the converter's early-out (pv night skip)
// ---------------------------------------------------------------------------------------
// A wave owns 128 consecutive cells; the keys (solar altitudes) of eight slots are loaded ahead, voted on (wave-
// uniform day mask), and only the day slots read their other streams and are converted; night slots contribute
// (write) exactly +0.0.  Same converter interface as k_fused_segred_night (key_load / key_is_zero / rest_load /
// compute_keyed).  Used for pv capacity-factor maps and per-cell series: 40 % fewer bytes on a year of data.
template <class Conv, bool VEC, bool SERIES>
__global__ __launch_bounds__(256, conv_min_waves_cells<Conv>::value) void k_cells_night(Conv conv, int64_t n_slots, int64_t S, int64_t chunk_len,
                                                                        double *__restrict__ out_a, double *__restrict__ out_b,
                                                                        int32_t conv_lds_doubles, int64_t X, int64_t Y, int32_t ntx, int32_t classes) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    // the wave's key rows: each lane parks the keys of a batch here and reads its own back inside the (rolled) slot
    // loop - eight unrolled conversions are 55 KB of code, the whole instruction cache
    double *vl = lds + conv_lds_doubles + wave * (kBatch * kSegCells) + 2 * lane;
    // the wave's 128 cells: a 16 x 8 tile of the grid when its row length is known (the fused kernels' tiles: the day /
    // night line crosses a compact tile in an eighth of the slots it takes to cross a 128-cell strip: 11.8 instead of
    // 13.7 GB read on C2), else 128 consecutive cells
    // (SERIES: chunks of consecutive slots, classes = 1; time-reduced: see class_walk above)
    const ClassWalk w = class_walk(SERIES ? 1 : classes, conv.S, n_slots, chunk_len);
    const auto slot_of = [&](int64_t i) { return w.first + w.step * i; };
    int64_t c0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 2 - w.o;
    bool v0 = c0 >= 0 && c0 < S, v1 = c0 + 1 >= 0 && c0 + 1 < S;
    if (X > 0) {
        const int64_t seg = int64_t(blockIdx.x) * 4 + wave;
        if (seg >= int64_t(ntx) * ((Y + 7) / 8)) return;
        const TileLane tl = tile_lane_cells(X, Y, ntx, 3, int32_t(seg), lane, w.o);
        c0 = tl.c0;
        v0 = tl.v0;
        v1 = tl.v1;
    }
    const bool own = v0 || v1;  // (on a shifted line grid a lane may own its second cell only)
    const int64_t s0c = own ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const int64_t s0 = w.s0, s1 = w.s1;
    double2 acc = {0.0, 0.0};
    int cnt0 = 0, cnt1 = 0;  // slots per chunk fit an int
    double2 key[kBatch];
    const bool any = s0 < s1;  // an empty time axis still writes its (0, 0) partials
#pragma unroll
    for (int i = 0; i < kBatch; ++i) key[i] = (own && any) ? conv.template key_load<VEC>(slot_of(min(s0 + i, s1 - 1)), s0c, s1c, cell) : double2{0.0, 0.0};
    for (int64_t sb = s0; sb < s1; sb += kBatch) {
        unsigned day = 0;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const bool d = (sb + i < s1) && !__all(conv.key_is_zero(key[i], slot_of(min(sb + i, s1 - 1)), cell) || !own);
            day |= d ? 1u << i : 0u;
            *reinterpret_cast<double2 *>(vl + i * kSegCells) = key[i];
        }
        const int nb = int(min(int64_t(kBatch), s1 - sb));
        if (sb + kBatch < s1) {  // next batch's keys: in flight behind this batch's conversions
#pragma unroll
            for (int i = 0; i < kBatch; ++i)
                key[i] = own ? conv.template key_load<VEC>(slot_of(min(sb + kBatch + i, s1 - 1)), s0c, s1c, cell) : double2{0.0, 0.0};
        }
        if constexpr (SERIES) {
#pragma unroll 1
            for (int i = 0; i < nb; ++i) {
                double2 r = {0.0, 0.0};
                if ((day >> i) & 1u) {  // wave-uniform
                    const typename Conv::Raw A = conv.template rest_load<VEC>(slot_of(sb + i), s0c, s1c, cell);
                    r = conv.compute_keyed(A, *reinterpret_cast<const double2 *>(vl + i * kSegCells), v0, v1, cell, lds);
                }
                st2<VEC>(out_a, slot_of(sb + i) * S + c0, v0, v1, r);
            }
        } else {
            unsigned m = day;
            while (m) {
                const int i = __builtin_ctz(m);
                m &= m - 1;
                const typename Conv::Raw A = conv.template rest_load<VEC>(slot_of(sb + i), s0c, s1c, cell);
                const double2 r = conv.compute_keyed(A, *reinterpret_cast<const double2 *>(vl + i * kSegCells), v0, v1, cell, lds);
                if (!dnan(r.x)) {
                    acc.x += r.x;
                    ++cnt0;
                }
                if (!dnan(r.y)) {
                    acc.y += r.y;
                    ++cnt1;
                }
            }
            const int night = nb - __builtin_popcount(day);  // night slots are valid zeros
            cnt0 += night;
            cnt1 += night;
        }
    }
    if constexpr (!SERIES) {
        const int64_t o = int64_t(blockIdx.y) * S + c0;
        if (v0) {
            out_a[o] = acc.x;
            out_b[o] = double(cnt0);
        }
        if (v1) {
            out_a[o + 1] = acc.y;
            out_b[o + 1] = double(cnt1);
        }
    }
}

// The early-out per-cell SERIES in flat order (see k_cells_series_flat): a wave converts its 128 cells of ONE slot and
// ends - key first, the other streams only when a covered cell is in daylight.  No batch of keys in flight ahead: the
// chip's occupancy hides the two dependent latencies, and the streams advance front to back together.
template <class Conv, class = void>
struct conv_flat_night : std::false_type {};
template <class Conv>
struct conv_flat_night<Conv, std::void_t<decltype(Conv::kFlatNightSeries)>> : std::integral_constant<bool, Conv::kFlatNightSeries> {};

template <class Conv>
__global__ __launch_bounds__(256) void k_cells_series_flat_night(Conv conv, int64_t S, uint32_t n_chunks, double *__restrict__ out, int64_t X,
                                                                 int64_t Y, int32_t ntx, int32_t shift, int64_t slot0) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const uint32_t slot_u = blockIdx.x / n_chunks;
    const int64_t slot = slot0 + slot_u;
    const int64_t chunk = blockIdx.x - slot_u * n_chunks;
    const int64_t o = shift ? (slot * conv.S) & 15 : 0;  // this slot's offset inside its 128-byte line (see k_cells_series_flat)
    int64_t c0 = (chunk * 256 + threadIdx.x) * 2 - o;
    bool v0 = c0 >= 0 && c0 < S, v1 = c0 + 1 >= 0 && c0 + 1 < S;
    if (X > 0) {  // 16 x 8 tiles, as k_cells_night - their rows on THIS slot's line grid
        const int64_t seg = chunk * 4 + wave;
        if (seg >= int64_t(ntx) * ((Y + 7) / 8)) return;
        const TileLane tl = tile_lane_cells(X, Y, ntx, 3, int32_t(seg), lane, o);
        c0 = tl.c0;
        v0 = tl.v0;
        v1 = tl.v1;
    }
    const int64_t s0c = (v0 || v1) ? c0 : 0, s1c = v1 ? c0 + 1 : (S > 1 ? 1 : 0);
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const bool any = v0 || v1;  // (a lane may own its second cell only: the first lies before the slot, on a shifted line grid)
    const double2 key = any ? conv.template key_load<true>(slot, s0c, s1c, cell) : double2{0.0, 0.0};
    double2 r = {0.0, 0.0};
    if (!__all(conv.key_is_zero(key, slot, cell) || !any)) {  // wave-uniform
        const typename Conv::Raw A = conv.template rest_load<true>(slot, s0c, s1c, cell);
        r = conv.compute_keyed(A, key, v0, v1, cell, lds);
    }
    st2<true>(out, slot * S + c0, v0, v1, r);
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

#ifndef ATL_ROW_CACHE_NIGHT
#define ATL_ROW_CACHE_NIGHT 2
#endif
constexpr int kRowCacheNight = ATL_ROW_CACHE_NIGHT;  // ... in the kernels that also keep a batch's values in LDS rows (4 waves per SIMD)
constexpr int kRowCache = ATL_ROW_CACHE;  // partial rows of a tile whose weights sit in the wave's LDS area for the chunk
constexpr int kRowCacheDense = 3;        // ... in the instantiation that also carries the MFMA path and its LDS value rows
constexpr int kDenseSlots = 2 * kBatch;  // slots a dense tile of k_fused_segred contracts per MFMA sweep (= the instruction's 16 columns)
template <bool DENSE>
constexpr int row_cache() {
    return DENSE ? kRowCacheDense : kRowCache;
}

// all partial rows of the tile for one converted batch: the first kRowCache rows with their weights from the
// wave's LDS area, the others from global memory (guarded path)
template <int ROWS = kRowCache>
__device__ __forceinline__ void reduce_batch(const double2 (&v)[kBatch], bool finite, const PlanDev &plan, int32_t p0,
                                             int32_t p1, const double *wlds, unsigned present, int lane, int64_t sb,
                                             int64_t send, double *__restrict__ partials, int64_t ldp) {
    const bool all_finite = __all(finite);  // wave-uniform
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
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
    for (int32_t p = p0 + ROWS; p < p1; ++p) {
        const double2 w = *reinterpret_cast<const double2 *>(plan.prow_w + int64_t(p) * kSegCells + 2 * lane);
        const bool a0 = !dnan(w.x), a1 = !dnan(w.y);
        double2 wz;
        wz.x = a0 ? w.x : 0.0;
        wz.y = a1 ? w.y : 0.0;
        // (an all-finite batch needs no guards here either: a zero weight times a finite value is a zero - the same
        // shortcut the cached rows take; 16 selects less per row and batch, which the early-out kernels with their two
        // cached rows feel)
        if (all_finite)
            reduce_row<false>(v, wz, true, true, lane, sb, send, partials + int64_t(p) * ldp);
        else
            reduce_row<true>(v, wz, a0, a1, lane, sb, send, partials + int64_t(p) * ldp);
    }
}


// register budget of the fused kernel: ATL_FUSED_WAVES waves per SIMD unless the converter asks for
// more registers (the general pv kernel is a long literal transcription and would spill)
// ---- dense tiles: the contraction on the matrix cores ------------------------------------------------------
// out[row][slot] = sum over the tile's 128 cells of weight[row][cell] * value[cell][slot] for 16 rows x 8 slots
// per group: 32 v_mfma_f64_16x16x4_f64 (A = 16 rows x 4 cells from the plan's operand image, one coalesced 512-byte
// load; B = 4 cells x 16 slots from the wave's LDS value rows, slots 8-15 are copies nobody stores; layouts
// verified by tools/probes/mfma_f64_layout.hip: A[i][k] lane 16 k + i, B[k][j] lane 16 k + j, D[4 r + l / 16][l % 16]
// in register r).  No shuffles at all, ~4x fewer issue cycles per partial row than the butterfly.  Only taken when
// every value of the batch is finite: 0 * NaN would leak through structural zeros (the guarded VALU path runs then).
typedef double d4 __attribute__((ext_vector_type(4)));

// NB = slots per batch: 8 (B columns 8-15 repeat 0-7, nobody stores them) or 16 (k_fused_segred's dense tiles: every
// column of the instruction is a slot, half the MFMAs and half the A-fragment loads per slot)
template <int U = 8, int NB = kBatch>  // U: K-steps whose operands are in flight together (4 VGPRs each)
__device__ __forceinline__ void reduce_dense_mfma(const double *vl, const double *__restrict__ wm, int G, int n_rows,
                                                  int32_t p0, int lane, int64_t sb, int64_t send,
                                                  double *__restrict__ partials, int64_t ldp, int g0 = 0) {
    static_assert(NB == 8 || NB == 16, "the fp64 MFMA has 16 columns");
    typedef __attribute__((address_space(1))) const double gdouble;
    const int j = lane & 15, kq = lane >> 4, jj = NB == 16 ? j : (j & 7);
    const double *brow = vl + jj * kSegCells + kq;
    for (int g = g0; g < G; ++g) {
        // two accumulators (even / odd K-steps): back-to-back MFMAs on ONE accumulator wait for each other
        d4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        gdouble *a = (gdouble *)(wm + (int64_t(g) * 32) * 64 + lane);
        // U K-steps per trip of a loop that is NOT unrolled: U operand pairs in flight, not all 32
#pragma unroll 1
        for (int k8 = 0; k8 < 32; k8 += U) {
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(k8 + u) * 64], brow[4 * ((k8 + u) ^ jj)], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(k8 + u + 1) * 64], brow[4 * ((k8 + u + 1) ^ jj)], acc1, 0, 0, 0);
            }
        }
        acc += acc1;
        if (j < NB && sb + j < send) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kMfmaRows * g + 4 * r + kq;
                if (row < n_rows) partials[int64_t(p0 + row) * ldp + sb + j] = acc[r];
            }
        }
    }
}

// The same contraction with the operand image of the tile's first GR groups held in registers (areg[g][k] = the lane's A
// value of group g, K-step k): no global load inside the sweep, and one B-fragment read from LDS serves both groups -
// their two accumulator chains interleave, so no second accumulator is needed there.  Groups past GR stream their image
// as before (reduce_dense_mfma from group GR on).
template <int GR>
__device__ __forceinline__ void reduce_dense_resident(const double *vl, const double (&areg)[GR][32], int G, int n_rows, int32_t p0,
                                                      int lane, int64_t sb, int64_t send, double *__restrict__ partials,
                                                      int64_t ldp) {
    const int j = lane & 15, kq = lane >> 4;
    const double *brow = vl + j * kSegCells + kq;
    const auto store = [&](int g, const d4 &acc) {
        if (sb + j < send) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kMfmaRows * g + 4 * r + kq;
                if (row < n_rows) partials[int64_t(p0 + row) * ldp + sb + j] = acc[r];
            }
        }
    };
    if (GR >= 2 && G >= 2) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const double b = brow[4 * (k ^ j)];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[0][k], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[GR >= 2 ? 1 : 0][k], b, acc1, 0, 0, 0);
        }
        store(0, acc0);
        store(1, acc1);
    } else {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[0][k], brow[4 * (k ^ j)], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[0][k + 1], brow[4 * ((k + 1) ^ j)], acc1, 0, 0, 0);
        }
        acc0 += acc1;
        store(0, acc0);
    }
}

// one batch of the fused kernel: kBatch slots from sb on -> v, and whether every value of it is finite.
// kGroup slots are LOADED before any of them is converted, so a light converter keeps 8 independent 1-KiB loads in
// flight per wave; slots past the end of a ragged chunk re-load its last slot (loads stay unconditional) and are zeroed
// afterwards.  Used by the dense-tile loop only: routed through this function (or a lambda) the SPARSE loop of the wind
// kernel took 168 VGPRs + 160 B of scratch instead of 106 VGPRs (C3 aggregated 3.65 -> 7.25 ms) and the pv kernel lost
// 3 % - the sparse loop keeps the same statements inline.
template <bool VEC, class Conv>
__device__ __forceinline__ void convert_batch(const Conv &conv, int64_t sb, int64_t send, bool covered, bool v0, bool v1,
                                              int64_t s0c, int64_t s1c, const typename Conv::Cell &cell,
                                              typename Conv::Carry &carry, const double *lds, double *vl, int row0, int lane,
                                              bool &finite) {
    // (the values go straight to the wave's swizzled LDS value rows row0 .. row0 + 7: held in registers until the end of
    // the batch they cost 32 VGPRs the resident operand image needs)
    constexpr int G = Conv::kGroup;
#pragma unroll
    for (int i0 = 0; i0 < kBatch; i0 += G) {
        typename Conv::Raw raw[G] = {};
        if (covered) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                raw[g] = conv.template load<VEC>(min(sb + i0 + g, send - 1), i0 + g, s0c, s1c, cell, carry);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g;
            const bool live = covered && sb + i < send;  // an uncovered lane converts zeros: whatever comes out is dropped
            double2 r = conv.compute(raw[g], v0, v1, cell, lds);
            r.x = live ? r.x : 0.0;
            r.y = live ? r.y : 0.0;
            // |x| < inf is false for NaN and +-inf
            finite = finite && (__builtin_fabs(r.x) < __builtin_inf()) && (__builtin_fabs(r.y) < __builtin_inf());
            *reinterpret_cast<double2 *>(vl + vrow_pair<true>(row0 + i, lane)) = r;
        }
    }
}

// converters whose cubes a line-aligned plan (PlanDev::shift_classes, atl_agg_create_aligned) may re-address: `S` is
// nothing but the distance between two slots, and a slot index means nothing else (no per-time tables, no day groups)
template <class Conv, class = void>
struct conv_shift_ok : std::false_type {};
template <class Conv>
struct conv_shift_ok<Conv, std::void_t<decltype(Conv::kShiftOk)>> : std::integral_constant<bool, Conv::kShiftOk> {};

// The lane's cells of a unit.  Ordinary plans: the tile's own (tile_lane_cells).  Line-aligned plans: the tile belongs
// to alignment class r, whose tiling has its rows on the line grid of c + o_r; its VIRTUAL slot i is real slot r + p i,
// and with the converter's slot stride set to p S the element (virtual slot i, cell c) sits at i (p S) + [r S + c]:
// `ld` is that bracket for the lane's first cell - 16-byte aligned per lane, 128-byte aligned per tile row, in every slot.
struct UnitCells {
    int64_t c0;        // the lane's first cell (cell_setup; the static per-cell fields)
    bool v0, v1;
    int64_t ld0, ld1;  // what load / key_load / rest_load index the cubes with (safe: loads never branch)
    int64_t slots;     // slots of this unit's class (ordinary plans: no limit)
};
__device__ __forceinline__ UnitCells unit_cells(const PlanDev &plan, int32_t seg, int lane, int64_t S, int64_t n_real) {
    UnitCells u;
    if (plan.shift_classes == 0) {
        const TileLane tl = tile_lane_cells(plan.X, plan.Y, plan.ntx, plan.w2_log2, seg, lane);
        u.c0 = tl.c0;
        u.v0 = tl.v0;
        u.v1 = tl.v1;
        u.ld0 = u.v0 ? u.c0 : 0;
        u.ld1 = u.v1 ? u.c0 + 1 : (S > 1 ? 1 : 0);
        u.slots = int64_t(1) << 62;
        return u;
    }
    const int32_t r = seg / plan.shift_tiles;  // wave-uniform
    const int64_t lo = int64_t(r) * S;         // real slot r starts here
    const int64_t o = lo & 15;
    const TileLane tl = tile_lane_cells(plan.X, plan.Y, plan.ntx, plan.w2_log2, seg - r * plan.shift_tiles, lane, o);
    u.c0 = tl.c0;
    u.v0 = tl.v0;
    u.v1 = tl.v1;
    // (a pair with one cell outside the slot reads the neighbouring slot's edge cell - class 0 is unshifted, so never
    //  before the cube; past the last slot only where the unshifted kernels would as well: vec_ok)
    u.ld0 = (u.v0 || u.v1) ? lo + u.c0 : lo - o;
    u.ld1 = u.ld0 + 1;
    u.slots = (n_real - r + plan.shift_classes - 1) / plan.shift_classes;
    return u;
}

template <class Conv, bool VEC, bool DENSE>
__global__ __launch_bounds__(kWavesPerBlock * 64, DENSE && min_waves<Conv>() > 2 ? 2 : min_waves<Conv>()) void k_fused_segred(Conv conv, PlanDev plan, int64_t slot0,
                                                      int64_t n_slots, int64_t S, int32_t chunk_slots,
                                                      int64_t n_units, double *__restrict__ partials,
                                                      int64_t ldp, int32_t conv_lds_doubles, int64_t n_real) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // the wave index is uniform: say so, and the unit / tile / chunk / slot arithmetic, the loop control and the
    // partial-row pointers live in SGPRs (they took ~16 VGPRs and 64-bit VALU multiplies per slot)
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    // per-wave LDS area behind the converter's tables: the weights of the tile's first kRowCache partial
    // rows (LDS instead of 4 VGPRs per row for the whole chunk: the register budget decides the occupancy)
    constexpr bool STAGE = conv_stage_values<Conv>::value && !DENSE;
    constexpr int ROWS = STAGE ? kRowCacheNight : row_cache<DENSE>();
    constexpr int kWaveLds = (ROWS + (STAGE ? kBatch : 0)) * kSegCells;
    double *wlds = lds + conv_lds_doubles + wave * kWaveLds;
    [[maybe_unused]] double *vstage = wlds + ROWS * kSegCells + 2 * lane;  // STAGE: this lane's 16 bytes of the wave's value rows
    // Linear order: consecutive blocks (= consecutive XCDs, block b runs on XCD b % 8) take consecutive
    // tile groups of the same time chunk, so the chip as a whole streams contiguous memory.  (An XCD-affine order - a group of
    // tiles always on the same XCD, its weights in one L2 - was measured in round 2 and removed in round 6: no gain, C2 3.44 vs
    // 3.41 ms; the weights are register-cached per chunk and < 1 % of the traffic.)
    const int64_t unit = int64_t(blockIdx.x) * kWavesPerBlock + wave;
    if (unit >= n_units) return;
    const int32_t seg = int32_t(unit % plan.n_segs);
    const int64_t chunk = unit / plan.n_segs;
    // tile coordinates -> the lane's two adjacent cells (atl_internal.h: tile_lane_cells; unit_cells above)
    const UnitCells uc = unit_cells(plan, seg, lane, S, n_real);
    const int64_t c0 = uc.c0;
    const bool v0 = uc.v0, v1 = uc.v1;
    const int64_t s0c = uc.ld0, s1c = uc.ld1;  // safe indices: loads never branch
    const int32_t p0 = plan.seg_ptr[seg], p1 = plan.seg_ptr[seg + 1];
    if (p0 == p1) return;  // no shape touches this tile: nothing to read
    // dense tile: its first 16 G rows go through the matrix cores (reduce_dense_mfma), the LDS value rows sit behind
    // the weight caches of the block (allocated only for plans that have dense tiles)
    // (DENSE = the plan has dense tiles: a separate instantiation, so that the common one keeps its registers)
    const int n_mfma = DENSE ? mfma_groups(p1 - p0) : 0;  // groups of 16 rows
    const double *wm = n_mfma ? plan.prow_wm + plan.seg_wm[seg] : nullptr;
    double *vl = lds + conv_lds_doubles + kWavesPerBlock * (ROWS * kSegCells) + wave * (kDenseSlots * kSegCells);
    // lanes whose cells carry no weight in any partial row do not load (PlanDev::seg_mask)
    const bool covered = (plan.seg_mask[seg] >> lane) & 1u;
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    // weights of the first kRowCache partial rows: in this wave's LDS area for the whole chunk (each lane
    // writes and later reads only its own 16 bytes: no barrier needed)
    unsigned present = 0;  // bit 2r / 2r+1: cell 0 / 1 structurally present in row r
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
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
    const int64_t send = min(min(sbeg + int64_t(chunk_slots), slot0 + n_slots), uc.slots);
    if (sbeg >= send) return;  // (a class of a line-aligned plan one slot shorter than the longest)
    partials -= slot0;
    typename Conv::Carry carry = carry_init<typename Conv::Carry>();
    if constexpr (DENSE) {
        if (n_mfma > 0) {
            // Dense tile: TWO batches per pass, so that all 16 columns of v_mfma_f64_16x16x4_f64 are slots (with 8 the
            // instruction's other half multiplies copies): both batches are converted into the wave's 16 LDS value rows,
            // then one MFMA sweep contracts them with the tile's operand image - half the MFMAs and half the operand
            // loads per slot.  Rows past the MFMA groups, and everything when a value is not finite (0 * NaN would
            // leak through structural zeros), go through the butterfly, batch by batch, from the rows.
            const int32_t pm = p0 + min(kMfmaRows * n_mfma, p1 - p0);
            // the operand image of the first GR groups: in registers for the whole chunk
            constexpr int GR = conv_dense_resident<Conv>::value;  // (0: every sweep streams the image, as until round 3)
            double areg[GR > 0 ? GR : 1][32];
            if constexpr (GR > 0) {
                typedef __attribute__((address_space(1))) const double gdouble;
#pragma unroll
                for (int g = 0; g < GR; ++g) {
                    gdouble *a = (gdouble *)(wm + (int64_t(min(g, n_mfma - 1)) * 32) * 64 + lane);
#pragma unroll
                    for (int k = 0; k < 32; ++k) areg[g][k] = a[k * 64];
                }
            }
            for (int64_t sb = sbeg; sb < send; sb += kDenseSlots) {
                bool finite = true;
#pragma unroll 1
                for (int h = 0; h < kDenseSlots / kBatch; ++h)
                    convert_batch<VEC>(conv, sb + h * kBatch, send, covered, v0, v1, s0c, s1c, cell, carry, lds, vl, h * kBatch, lane, finite);
                const bool all_finite = __all(finite);
                if (all_finite) {
                    if constexpr (GR > 0) reduce_dense_resident<GR>(vl, areg, n_mfma, p1 - p0, p0, lane, sb, send, partials, ldp);
                    if (n_mfma > GR) reduce_dense_mfma<8, kDenseSlots>(vl, wm, n_mfma, p1 - p0, p0, lane, sb, send, partials, ldp, GR);
                }
                if (!all_finite || pm < p1) {
#pragma unroll 1
                    for (int h = 0; h < kDenseSlots / kBatch; ++h) {
                        if (sb + h * kBatch >= send) break;
                        double2 v[kBatch];
#pragma unroll
                        for (int i = 0; i < kBatch; ++i) v[i] = *reinterpret_cast<const double2 *>(vl + vrow_pair<true>(h * kBatch + i, lane));
                        reduce_batch<0>(v, all_finite, plan, all_finite ? pm : p0, p1, wlds, 0u, lane, sb + h * kBatch, send, partials, ldp);
                    }
                }
            }
            return;
        }
    }
    for (int64_t sb = sbeg; sb < send; sb += kBatch) {
        double2 v[kBatch];
        bool finite = true;
        {  // convert_batch's statements, inline (see there)
            constexpr int G = Conv::kGroup;
#pragma unroll
            for (int i0 = 0; i0 < kBatch; i0 += G) {
                typename Conv::Raw raw[G] = {};
                if (covered) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        raw[g] = conv.template load<VEC>(min(sb + i0 + g, send - 1), i0 + g, s0c, s1c, cell, carry);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int i = i0 + g;
                    const bool live = covered && sb + i < send;
                    v[i] = conv.compute(raw[g], v0, v1, cell, lds);
                    v[i].x = live ? v[i].x : 0.0;
                    v[i].y = live ? v[i].y : 0.0;
                    finite = finite && (__builtin_fabs(v[i].x) < __builtin_inf()) && (__builtin_fabs(v[i].y) < __builtin_inf());
                    if constexpr (STAGE) *reinterpret_cast<double2 *>(vstage + i * kSegCells) = v[i];
                }
            }
            if constexpr (STAGE) {
                // the values went to the wave's LDS rows as they were converted and come back for the reduction: the
                // compiler must not forward them through registers (each lane reads its own 16 bytes: no barrier)
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int i = 0; i < kBatch; ++i) v[i] = *reinterpret_cast<const double2 *>(vstage + i * kSegCells);
            }
        }
        reduce_batch<ROWS>(v, finite, plan, p0, p1, wlds, present, lane, sb, send, partials, ldp);  // (sparse tile)
    }
}

// ---------------------------------------------------------------------------------------
// kernel 3b: fused convert + segment reduce with a per-slot early-out (pv night skip)
// ---------------------------------------------------------------------------------------
// Same tiles, chunks, partial rows and reduction as k_fused_segred - bit-identical output - but the slots of
// a batch that convert to +0.0 for the whole tile (every cell below the altitude cut-off) are neither read nor
// converted.  The batch's key values (altitudes) are prefetched one batch ahead, voted on (wave-uniform day
// mask) and parked in the wave's LDS rows; a loop over the set bits of the mask converts the day slots and
// overwrites their rows, night rows hold +0.0, and the reduction reads the eight rows back.  With the values in
// LDS and a loop instead of eight unrolled slots the kernel needs < 128 VGPRs: 4 waves per SIMD, for which
// the LDS budget (160 KiB / 16 waves) leaves kRowCacheNight = 2 weight rows per wave.  Converters opt in with
// kNightPipe and provide key_load / key_is_zero / rest_load / compute_keyed.
template <class Conv, bool VEC, bool DENSE, bool MAP = false>
__global__ __launch_bounds__(kWavesPerBlock * 64, DENSE ? 2 : min_waves<Conv>()) void k_fused_segred_night(Conv conv, PlanDev plan, int64_t slot0,
                                                      int64_t n_slots, int64_t S, int32_t chunk_slots,
                                                      int64_t n_units, double *__restrict__ partials,
                                                      int64_t ldp, int32_t conv_lds_doubles, int64_t n_real) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    constexpr int kWaveLds = (kRowCacheNight + kBatch) * kSegCells;
    double *wlds = lds + conv_lds_doubles + wave * kWaveLds;
    double *vl = wlds + kRowCacheNight * kSegCells;  // the wave's value rows (swizzled when DENSE: vrow_pair)
    const int64_t unit = int64_t(blockIdx.x) * kWavesPerBlock + wave;
    if (unit >= n_units) return;
    const int32_t seg = int32_t(unit % plan.n_segs);
    const int64_t chunk = unit / plan.n_segs;
    const UnitCells uc = unit_cells(plan, seg, lane, S, n_real);
    const int64_t c0 = uc.c0;
    const bool v0 = uc.v0, v1 = uc.v1;
    const int64_t s0c = uc.ld0, s1c = uc.ld1;
    const int32_t p0 = plan.seg_ptr[seg], p1 = plan.seg_ptr[seg + 1];
    if (p0 == p1) return;
    const int G = DENSE ? mfma_groups(p1 - p0) : 0;
    const double *wm = G ? plan.prow_wm + plan.seg_wm[seg] : nullptr;
    const typename Conv::Cell cell = conv.cell_setup(c0, v0, v1, lds);
    const bool covered = (plan.seg_mask[seg] >> lane) & 1u;  // lanes without a weighted cell do not load and vote "night"
    unsigned present = 0;
#pragma unroll
    for (int r = 0; r < kRowCacheNight; ++r) {
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
    const int64_t sbeg = slot0 + chunk * chunk_slots;
    const int64_t send = min(min(sbeg + int64_t(chunk_slots), slot0 + n_slots), uc.slots);
    if (sbeg >= send) return;
    partials -= slot0;
    if constexpr (MAP) {
        // The day bits of this tile were computed once for the (plan, altitude cube, cut-off) - k_day_map, the same vote - :
        // a batch reads eight bytes through the scalar cache instead of eight altitude pairs per lane; a dark slot reads
        // NOTHING, a day slot issues all of its streams at once (no key-then-rest dependency).
        // LINE-granular since round 6: a slot's byte has one bit per 128-byte line of the tile (lanes 8 j .. 8 j + 7 = the 16
        // consecutive cells of one line of every cube, whatever the tile's shape) - the lanes of a line without a cell above the
        // cut-off do not load: in the slots in which the terminator crosses a tile only its lit lines are fetched (round 5
        // fetched the whole tile: 1.09 x the day cells' bytes).  A dark lane's cells convert to +0.0 (irradiation.py:251-252),
        // which is what it contributes without having loaded anything: the same bits.
        static_assert(!DENSE, "day map: sparse tiles");
        const uint8_t *mrow = conv.in.d_day_map + int64_t(seg) * conv.in.day_map_ld;
        const unsigned my_line = unsigned(lane) >> 3;
        for (int64_t sb = sbeg; sb < send; sb += kBatch) {
            const int64_t rel = sb - slot0;  // the map is indexed by the call's own slots (a multiple of kBatch: run_fused)
            const uint2 w2 = *reinterpret_cast<const uint2 *>(mrow + rel);
            uint64_t lines = uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(w2.x)))) |
                             (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(w2.y)))) << 32);
            if (send - sb < kBatch) lines &= (uint64_t(1) << (8 * int(send - sb))) - 1u;
            // slot i is a day slot iff its byte is not zero: fold every byte onto its lowest bit, gather the eight bits
            uint64_t f = lines | (lines >> 4);
            f |= f >> 2;
            f |= f >> 1;
            const unsigned day = unsigned(((f & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
            // (round 6 tried writing the zeros of a batch that is dark for the whole tile - a third of the batches - straight to the
            //  partial rows, without value rows or a reduction: no gain, 1.919 vs 1.920 ms on one box, gpurun_out/r06_r; removed)
#pragma unroll
            for (int i = 0; i < kBatch; ++i) *reinterpret_cast<double2 *>(vl + vrow_pair<false>(i, lane)) = double2{0.0, 0.0};
            bool finite = true;
            unsigned m = day;
            // (one register set: a day slot's loads wait for the previous conversion.  A second set - the next day slot's seven
            //  streams in flight during the conversion - fits the 128 registers and was 3 % SLOWER on C2, 1.984 vs 1.919 ms
            //  (round 5, gpurun_out/r05_d): the kernel is issue-bound; removed in round 6)
            typename Conv::Carry carry{};
            while (m) {
                const int p1s = __builtin_ctz(m);
                m &= m - 1;
                const bool lit = covered && ((unsigned(lines >> (8 * p1s)) >> my_line) & 1u);
                typename Conv::Raw A1 = {};
                if (lit) A1 = conv.template load<VEC>(sb + p1s, p1s, s0c, s1c, cell, carry);
                double2 r = conv.compute(A1, v0, v1, cell, lds);
                r.x = lit ? r.x : 0.0;
                r.y = lit ? r.y : 0.0;
                finite = finite && (__builtin_fabs(r.x) < __builtin_inf()) && (__builtin_fabs(r.y) < __builtin_inf());
                *reinterpret_cast<double2 *>(vl + vrow_pair<false>(p1s, lane)) = r;
            }
            double2 v[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) v[i] = *reinterpret_cast<const double2 *>(vl + vrow_pair<false>(i, lane));
            reduce_batch<kRowCacheNight>(v, finite, plan, p0, p1, wlds, present, lane, sb, send, partials, ldp);
        }
        return;
    }
    double2 key[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) key[i] = covered ? conv.template key_load<VEC>(min(sbeg + i, send - 1), s0c, s1c, cell) : double2{0.0, 0.0};
    for (int64_t sb = sbeg; sb < send; sb += kBatch) {
        // votes: bit i of day = some cell of the tile is converted in slot sb + i
        unsigned day = 0;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const bool d = (sb + i < send) && !__all(conv.key_is_zero(key[i], min(sb + i, send - 1), cell) || !covered);
            day |= d ? 1u << i : 0u;
            *reinterpret_cast<double2 *>(vl + vrow_pair<DENSE>(i, lane)) = d ? key[i] : double2{0.0, 0.0};
        }
        bool finite = true;
        if (day != 0) {
            unsigned m = day;
            // one register set per wave and no software pipelining: what hides the latency is the FOURTH wave per
            // SIMD the smaller footprint allows (measured on C2: pipelined day slots with two register sets at 3
            // waves 2.32 ms, this loop at 3 waves 2.19 ms, at 4 waves 2.13 ms)
            while (m) {
                const int p = __builtin_ctz(m);
                m &= m - 1;
                typename Conv::Raw A = {};
                if (covered) A = conv.template rest_load<VEC>(sb + p, s0c, s1c, cell);
                const double2 kv = *reinterpret_cast<const double2 *>(vl + vrow_pair<DENSE>(p, lane));
                double2 r = conv.compute_keyed(A, kv, v0, v1, cell, lds);
                r.x = covered ? r.x : 0.0;
                r.y = covered ? r.y : 0.0;
                finite = finite && (__builtin_fabs(r.x) < __builtin_inf()) && (__builtin_fabs(r.y) < __builtin_inf());
                *reinterpret_cast<double2 *>(vl + vrow_pair<DENSE>(p, lane)) = r;
            }
        }
        const bool dense = DENSE && G > 0 && __all(finite);  // wave-uniform
        if (dense) reduce_dense_mfma(vl, wm, G, p1 - p0, p0, lane, sb, send, partials, ldp);
        // next batch's keys: in flight behind the (VALU) reduction; a dense tile's MFMA phase runs before the
        // prefetch so that its operands and the 32 key registers are not live together
        if (sb + kBatch < send) {
#pragma unroll
            for (int i = 0; i < kBatch; ++i) key[i] = covered ? conv.template key_load<VEC>(min(sb + kBatch + i, send - 1), s0c, s1c, cell) : double2{0.0, 0.0};
        }
        if (!dense || kMfmaRows * G < p1 - p0) {
            double2 v[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) v[i] = *reinterpret_cast<const double2 *>(vl + vrow_pair<DENSE>(i, lane));
            if (G > 0)
                reduce_batch<0>(v, finite, plan, dense ? p0 + kMfmaRows * G : p0, p1, wlds, 0u, lane, sb, send, partials, ldp);
            else
                reduce_batch<kRowCacheNight>(v, finite, plan, p0, p1, wlds, present, lane, sb, send, partials, ldp);
        }
    }
}

// The early-out's votes, once: map[tile * ld + t] = one bit per 128-byte line of the tile (bit j: lanes 8 j .. 8 j + 7), set iff in
// slot t some covered cell of that line has a key that does not convert to +0.0 - literally the test k_fused_segred_night
// makes on the keys it loads (same converter functions, same lanes, same coverage mask); the slot is a day slot of the tile
// iff the byte is not zero.  One wave per (tile, run of 64 slots); bytes past the last slot's are 0.
template <class Conv, bool VEC>
__global__ __launch_bounds__(256) void k_day_map(Conv conv, PlanDev plan, int64_t n_slots, int64_t S, int64_t n_units,
                                                 uint8_t *__restrict__ map, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    conv.block_init(lds);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int64_t unit = int64_t(blockIdx.x) * 4 + wave;
    if (unit >= n_units) return;
    const int32_t seg = int32_t(unit % plan.n_segs);
    const int64_t run = unit / plan.n_segs;
    const UnitCells uc = unit_cells(plan, seg, lane, S, n_slots);
    const typename Conv::Cell cell = conv.cell_setup(uc.c0, uc.v0, uc.v1, lds);
    const bool covered = (plan.seg_mask[seg] >> lane) & 1u;
    const int64_t sbeg = run * 64, send = min(sbeg + 64, n_slots);
    for (int64_t sb = sbeg; sb < send; sb += kBatch) {
        double2 key[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
            key[i] = covered ? conv.template key_load<VEC>(min(sb + i, send - 1), uc.ld0, uc.ld1, cell) : double2{0.0, 0.0};
        uint64_t lines = 0;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const bool up = (sb + i < send) && covered && !conv.key_is_zero(key[i], min(sb + i, send - 1), cell);
            uint64_t f = __ballot(up);  // lane bits -> one bit per group of eight lanes
            f |= f >> 4;
            f |= f >> 2;
            f |= f >> 1;
            lines |= (((f & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * i);
        }
        if (lane == 0) *reinterpret_cast<uint64_t *>(map + int64_t(seg) * ld + sb) = lines;  // (ld and sb are multiples of 8)
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

// ... of a line-aligned plan: stacked row r N + n, virtual slot w0 + t  ->  out[n, r + p (w0 + t)]
__global__ __launch_bounds__(256) void k_combine_aligned(PlanDev plan, const double *__restrict__ partials, int64_t ldp, int64_t w0, int64_t wn,
                                                         int64_t n_real, double *__restrict__ out, int64_t ld_out) {
    const int64_t tv = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    const int64_t r = row / plan.shift_rows, n = row - r * plan.shift_rows;
    const int64_t t = r + int64_t(plan.shift_classes) * (w0 + tv);
    if (tv >= wn || t >= n_real) return;
    double s = 0.0;
    const int32_t q0 = plan.shape_ptr[row], q1 = plan.shape_ptr[row + 1];
    for (int32_t q = q0; q < q1; ++q) s += partials[int64_t(plan.shape_prow[q]) * ldp + tv];
    if (plan.row_poison[row]) s = __builtin_nan("");
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
// host-side launch plumbing
// ---------------------------------------------------------------------------------------
// $ATLITE_HIP_NO_VEC: every launch through the unvectorised instantiations (tests)
inline bool no_vec() { return getenv("ATLITE_HIP_NO_VEC") != nullptr; }  // read per call: tests switch it in-process
inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

struct KernelBracket {
    atl_ctx *ctx;
    size_t slot = 0;
    explicit KernelBracket(atl_ctx *c) : ctx(c) {
        if (ctx->profiling && !ctx->capturing) {
            slot = size_t(ctx->ring_count % int64_t(ctx->ev_ring.size() / 2));
            (void)hipEventRecord(ctx->ev_ring[2 * slot], ctx->stream);
        }
    }
    ~KernelBracket() {
        if (ctx->profiling && !ctx->capturing) {
            (void)hipEventRecord(ctx->ev_ring[2 * slot + 1], ctx->stream);
            ++ctx->ring_count;
        }
    }
};

inline bool debug_occupancy() {
    static const bool on = getenv("ATLITE_HIP_DEBUG_OCCUPANCY") != nullptr;
    return on;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

// chunk of output slots walked by one wave of the fused kernel
int32_t pick_chunk_slots(const atl_ctx *ctx, int64_t n_slots, int64_t n_segs, int64_t min_chunk) {
    // aim for >= ~16 waves per CU worth of units, chunks a multiple of kBatch in [8, 64]
    // (longer chunks - fewer re-reads of a tile's weight rows, 4-5 % of the one-cube converters' traffic - are slower for every
    //  converter: 128 / 256 / 512 slots, C5 heat demand 0.853 -> 0.971 / 1.003 / 1.006 ms, runoff 0.858 -> 0.961 / 0.902 / 0.994, C2 pv 3.11 -> 3.20 /
    //  3.28 / 3.40: tools/jobs/r06_chunk.sh)
    int64_t chunk = 64;
    if (const char *e = getenv("ATLITE_HIP_CHUNK")) {  // experiments: any multiple of kBatch
        const int64_t v = atoll(e);
        if (v >= kBatch && v % kBatch == 0) return int32_t(v);
    }
    // ~12 waves per CU are resident, so the units run in rounds of ~3000: the last, partly filled round is dead time
    // (measured on C2, 325 tiles: 64-slot chunks = 14 rounds 2.99 ms, 16-slot chunks = 58 rounds 2.94 ms).  Halve the
    // chunk down to two batches while there are fewer than ~40 rounds; below that only when the chip would not fill.
    const auto units = [&](int64_t c) { return n_segs * ((n_slots + c - 1) / c); };
    const int64_t rounds40 = int64_t(ctx->n_cu) * 12 * 40;
    while (chunk > min_chunk && units(chunk) < rounds40) chunk /= 2;
    const int64_t want_units = int64_t(ctx->n_cu) * 64;
    while (chunk > kBatch && units(chunk) < want_units) chunk /= 2;
    return int32_t(chunk);
}

constexpr size_t kCellsNightLds = 4 * kBatch * kSegCells * sizeof(double);  // k_cells_night: key rows of the block's four waves

// slots per block of the per-cell kernels that walk a slot range: as long as possible while the grid still fills the chip
// (16 blocks per CU), whole batches
inline int64_t slot_chunk_len(const atl_ctx *ctx, int64_t n_slots, unsigned gx) {
    const int64_t per_cu = 16;
    const int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>((n_slots + 15) / 16, (int64_t(ctx->n_cu) * per_cu + gx - 1) / gx));
    const int64_t len = std::max<int64_t>(1, (n_slots + n_chunks - 1) / n_chunks);
    return (len + kBatch - 1) / kBatch * kBatch;
}

template <class Conv>
int run_cells(atl_ctx *ctx, const Conv &conv, bool vec, size_t lds_bytes, int64_t n_slots, int64_t S,
              int time_agg, double *d_out, const char *what, int64_t row_len = 0) {
    ATL_REQUIRE(time_agg >= ATL_TIME_NONE && time_agg <= ATL_TIME_SUM_COUNT, "%s: bad time_agg %d", what, time_agg);
    ATL_REQUIRE(d_out, "%s: d_out is NULL", what);
    ATL_REQUIRE(slot_stride_of(ctx, S) >= S, "%s: slot stride %lld is smaller than the %lld cells of a slot", what,
                (long long)slot_stride_of(ctx, S), (long long)S);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (S == 0 || (n_slots == 0 && time_agg == ATL_TIME_NONE)) return ATL_OK;
    unsigned gx = unsigned((S + 511) / 512);
    const unsigned gx_cells = gx;  // the slot chunking follows the cell count alone: the same summation order in every variant
    vec = vec && aligned8(d_out);
    // row_len = X of the (Y, X) grid, when the caller knows it: the early-out kernels then walk 16 x 8 tiles
    int64_t tX = 0, tY = 0;
    int32_t ntx = 0;
    if constexpr (conv_night_pipe<Conv>::value) {
        // (only when the slots are 128-byte aligned - stride % 16 == 0: a 16-cell tile row off the line grid costs two
        // lines; the 128-cell strips lose one line in nine)
        if (row_len > 0 && S % row_len == 0 && S / row_len < (int64_t(1) << 30) && slot_stride_of(ctx, S) % 16 == 0) {
            tX = row_len;
            tY = S / row_len;
            ntx = int32_t(tile_columns(tX, tY, 3));
            gx = unsigned((int64_t(ntx) * ((tY + 7) / 8) + 3) / 4);
        }
    }
    constexpr bool kScalarToo = !conv_vec_only<Conv>::value;  // the unvectorised instantiations exist
    if (!kScalarToo && !vec) return kNeedScalar;
    if (time_agg == ATL_TIME_NONE) {
        const dim3 grid(gx, unsigned((n_slots + kSeriesSlots - 1) / kSeriesSlots));
        KernelBracket kb(ctx);
        // the flat kernels' grids are (chunks of a slot) x (slots) blocks in ONE dimension: a launch carries at most 2^30 blocks,
        // longer series go in consecutive slot ranges (slot0) - the converters they serve have no slot-walking vectorised
        // instantiation since round 6 (47 kernels less)
        const auto flat_ranges = [&](unsigned gxf, auto &&launch) {
            const int64_t per = std::max<int64_t>(1, ((int64_t(1) << 30) - 1) / std::max(1u, gxf));
            for (int64_t a0 = 0; a0 < n_slots; a0 += per) launch(a0, std::min(per, n_slots - a0));
        };
        if constexpr (conv_night_pipe<Conv>::value) {  // long slot ranges: the keys are fetched one batch ahead
            const int64_t len = slot_chunk_len(ctx, n_slots, gx_cells);
            const dim3 gridn(gx, unsigned((n_slots + len - 1) / len));
            if constexpr (conv_flat_night<Conv>::value) {
                if (vec) {
                    // slots off the line grid (a caller's contiguous cubes, S % 16 != 0): one block lives for one slot, so its
                    // tiles / strips can sit on THAT slot's line grid (o = (slot * stride) % 16 cells; ATLITE_HIP_SERIES_NO_SHIFT: as before)
                    const int32_t shift = slot_stride_of(ctx, S) % 16 != 0 && !getenv("ATLITE_HIP_SERIES_NO_SHIFT");
                    int64_t fX = tX, fY = tY;
                    int32_t fntx = ntx;
                    unsigned gxf = gx;
                    if (shift) {
                        gxf = unsigned((S + 15 + 511) / 512);
                        if (row_len > 0 && S % row_len == 0 && S / row_len < (int64_t(1) << 30)) {
                            fX = row_len;
                            fY = S / row_len;
                            fntx = int32_t(tile_columns(fX, fY, 3, true));
                            gxf = unsigned((int64_t(fntx) * ((fY + 7) / 8) + 3) / 4);
                        }
                    }
                    flat_ranges(gxf, [&](int64_t a0, int64_t nb) {
                        hipLaunchKernelGGL((k_cells_series_flat_night<Conv>), dim3(unsigned(int64_t(gxf) * nb)), dim3(256), lds_bytes, ctx->stream, conv,
                                           S, uint32_t(gxf), d_out, fX, fY, fntx, shift, a0);
                    });
                } else if constexpr (kScalarToo) {
                    hipLaunchKernelGGL((k_cells_night<Conv, false, true>), gridn, dim3(256), lds_bytes + kCellsNightLds, ctx->stream, conv, n_slots, S,
                                       len, d_out, static_cast<double *>(nullptr), int32_t(lds_bytes / sizeof(double)), tX, tY, ntx, 1);
                }
            } else if (vec) {
                hipLaunchKernelGGL((k_cells_night<Conv, true, true>), gridn, dim3(256), lds_bytes + kCellsNightLds, ctx->stream, conv, n_slots, S,
                                   len, d_out, static_cast<double *>(nullptr), int32_t(lds_bytes / sizeof(double)), tX, tY, ntx, 1);
            } else if constexpr (kScalarToo) {
                hipLaunchKernelGGL((k_cells_night<Conv, false, true>), gridn, dim3(256), lds_bytes + kCellsNightLds, ctx->stream, conv, n_slots, S,
                                   len, d_out, static_cast<double *>(nullptr), int32_t(lds_bytes / sizeof(double)), tX, tY, ntx, 1);
            }
        } else if constexpr (conv_flat_series<Conv>::value) {  // flat order: see k_cells_series_flat
            if (vec) {
                const int32_t shift = slot_stride_of(ctx, S) % 16 != 0 && !getenv("ATLITE_HIP_SERIES_NO_SHIFT");
                const unsigned gxs = shift ? unsigned((S + 15 + 511) / 512) : gx_cells;
                flat_ranges(gxs, [&](int64_t a0, int64_t nb) {
                    hipLaunchKernelGGL((k_cells_series_flat<Conv>), dim3(unsigned(int64_t(gxs) * nb)), dim3(256), lds_bytes, ctx->stream, conv, S,
                                       uint32_t(gxs), d_out, shift, a0);
                });
            } else if constexpr (kScalarToo) {
                hipLaunchKernelGGL((k_cells_series<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv, n_slots, S, d_out);
            }
        } else if (vec) {  // (the early-out converters do not instantiate the plain series kernel)
            hipLaunchKernelGGL((k_cells_series<Conv, true>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, d_out);
        } else if constexpr (kScalarToo) {
            hipLaunchKernelGGL((k_cells_series<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, d_out);
        }
        return check_launch(what);
    }
    // time-reduced: split the slot axis so that the grid fills the chip
    int64_t chunk_len = slot_chunk_len(ctx, n_slots, gx_cells);
    int64_t n_chunks = std::max<int64_t>(1, (n_slots + chunk_len - 1) / chunk_len);
    // cubes whose slots do not start on 128-byte lines (a caller's contiguous cubes, S % 16 != 0): the chunks split by the
    // alignment class of their slots, each class on its own line grid (class_walk) - about as many chunks as before
    int32_t classes = 1;
    if constexpr (conv_shift_ok<Conv>::value) {
        const int64_t stride = slot_stride_of(ctx, S);
        // (the unvectorised instantiations walk the same classes: the same summation order, the same bits)
        if (stride % 16 != 0 && S >= 16 && !getenv("ATLITE_HIP_SERIES_NO_SHIFT")) {
            int64_t g = stride % 16, b = 16;
            while (g) {  // gcd(stride, 16)
                const int64_t t = b % g;
                b = g;
                g = t;
            }
            classes = int32_t(16 / b);
            const int64_t per_class = (n_slots + classes - 1) / classes;           // slots of the longest class
            const int64_t cpc = std::max<int64_t>(1, (n_chunks + classes - 1) / classes);  // chunks per class
            chunk_len = std::max<int64_t>(kBatch, ((per_class + cpc - 1) / cpc + kBatch - 1) / kBatch * kBatch);  // (an empty time axis: one empty chunk per class)
            n_chunks = classes * std::max<int64_t>(1, (per_class + chunk_len - 1) / chunk_len);
            gx = unsigned((S + 15 + 511) / 512);
            if (tX > 0 || (conv_night_pipe<Conv>::value && row_len > 0 && S % row_len == 0 && S / row_len < (int64_t(1) << 30))) {
                tX = row_len;  // the early-out kernel keeps its 16 x 8 tiles, their rows on the class's line grid
                tY = S / row_len;
                ntx = int32_t(tile_columns(tX, tY, 3, true));
                gx = unsigned((int64_t(ntx) * ((tY + 7) / 8) + 3) / 4);
            }
        }
    }
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, size_t(2 * n_chunks * S) * sizeof(double), &scr);
    if (rc) return rc;
    double *psum = static_cast<double *>(scr), *pcnt = psum + n_chunks * S;
    {
        const dim3 grid(gx, unsigned(n_chunks));
        KernelBracket kb(ctx);
        if constexpr (conv_night_pipe<Conv>::value) {
            if (vec)
                hipLaunchKernelGGL((k_cells_night<Conv, true, false>), grid, dim3(256), lds_bytes + kCellsNightLds, ctx->stream, conv, n_slots, S,
                                   chunk_len, psum, pcnt, int32_t(lds_bytes / sizeof(double)), tX, tY, ntx, classes);
            else if constexpr (kScalarToo)
                hipLaunchKernelGGL((k_cells_night<Conv, false, false>), grid, dim3(256), lds_bytes + kCellsNightLds, ctx->stream, conv, n_slots, S,
                                   chunk_len, psum, pcnt, int32_t(lds_bytes / sizeof(double)), tX, tY, ntx, classes);
        } else if (vec) {
            hipLaunchKernelGGL((k_cells_timered<Conv, true>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, chunk_len, psum, pcnt, classes);
        } else if constexpr (kScalarToo) {
            hipLaunchKernelGGL((k_cells_timered<Conv, false>), grid, dim3(256), lds_bytes, ctx->stream, conv,
                               n_slots, S, chunk_len, psum, pcnt, classes);
        }
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
    const bool aligned = agg->dev.shift_classes > 0;  // line-aligned plan (atl_agg_create_aligned)
    ATL_REQUIRE(agg->dev.n_cells == S, "%s: matrix has %lld columns but the cutout has %lld cells", what,
                (long long)agg->dev.n_cells, (long long)S);
    if (aligned) {
        // refusals of a line-aligned plan are ATL_E_UNSUPPORTED, not ATL_E_INVALID: the caller (the gateway) takes the
        // ordinary plan instead - told apart by the code, not by the wording
        if (!conv_shift_ok<Conv>::value) {
            set_error("%s: this conversion cannot run on a line-aligned plan (atl_agg_create_aligned)", what);
            return ATL_E_UNSUPPORTED;
        }
        if (slot_stride_of(ctx, S) != S) {
            set_error("%s: a line-aligned plan is for contiguous cubes (slot stride %lld, %lld cells)", what,
                      (long long)slot_stride_of(ctx, S), (long long)S);
            return ATL_E_UNSUPPORTED;
        }
        if (!vec) {
            set_error("%s: a line-aligned plan needs the vectorised kernels (8-byte aligned cubes that do not end on a page boundary)", what);
            return ATL_E_UNSUPPORTED;
        }
        if constexpr (conv_shift_ok<Conv>::value) {
            if (conv.S != agg->dev.shift_classes * S) {  // the converter's slot stride: p slots of the contiguous cubes
                Conv strided = conv;
                strided.S = agg->dev.shift_classes * S;
                return run_fused(ctx, strided, vec, lds_bytes, n_slots, S, agg, time_agg, d_out, ld_out, what);
            }
        }
    }
    ATL_REQUIRE(time_agg >= ATL_TIME_NONE && time_agg <= ATL_TIME_SUM_COUNT, "%s: bad time_agg %d", what, time_agg);
    ATL_REQUIRE(d_out, "%s: d_out is NULL", what);
    ATL_REQUIRE(time_agg != ATL_TIME_NONE || ld_out >= n_slots, "%s: ld_out %lld < %lld", what,
                (long long)ld_out, (long long)n_slots);
    ATL_REQUIRE(slot_stride_of(ctx, S) >= S, "%s: slot stride %lld is smaller than the %lld cells of a slot", what,
                (long long)slot_stride_of(ctx, S), (long long)S);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    const PlanDev &plan = agg->dev;
    const int64_t N = aligned ? plan.shift_rows : plan.n_rows;
    if (N == 0) return ATL_OK;
    // line-aligned plan: the kernels walk VIRTUAL slots (slot i of class r = real slot r + p i); k_combine_aligned puts them back
    const int64_t n_real = n_slots;
    if (aligned) n_slots = (n_slots + plan.shift_classes - 1) / plan.shift_classes;
    // (a lane's cell pair may straddle two grid rows - odd row lengths: cells are owned by flat index, tile_lane_cells)
    constexpr bool kScalarToo = !conv_vec_only<Conv>::value;  // the unvectorised instantiations exist
    if (!kScalarToo && !vec) return kNeedScalar;
    // Partial rows live in scratch as [P][window]; the slot axis is processed in windows so that the
    // scratch stays below ~1 GiB however dense the matrix is (P = tiles x shapes for a dense one).
    const int64_t P = plan.n_prows;
    int64_t window = std::max<int64_t>(n_slots, 1);
    int64_t budget = int64_t(1) << 27;  // doubles = 1 GiB
    if (const char *env = getenv("ATLITE_HIP_PARTIAL_BUDGET")) budget = std::max<int64_t>(1, atoll(env));  // tests
    const int64_t budget_slots = budget / std::max<int64_t>(P, 1);
    if (window > budget_slots) window = std::max<int64_t>(64, budget_slots / 64 * 64);
    const int64_t ldp = int64_t(align_up(size_t(window), 8));
    const int64_t lds_series = int64_t(align_up(size_t(std::max<int64_t>(n_real, 1)), 8));
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
            const bool dense_plan = plan.prow_wm != nullptr && vec && conv_dense_ok<Conv>::value;
            // shortest chunk: the weight rows a unit reads (P / tiles KiB) against the KiB of cube data per slot, so that they
            // stay a few % of its traffic (profiles/r03_interleave_probe.txt section 8: runoff wants 64 slots from 4 rows per
            // tile, pv 16 up to 8 rows); never below the converter's own floor
            int64_t min_chunk = 2 * kBatch;
            while (min_chunk < 64 && 2 * min_chunk * conv_cubes<Conv>::value * int64_t(plan.n_segs) <= 24 * P) min_chunk *= 2;
            min_chunk = std::max<int64_t>(min_chunk, conv_min_chunk<Conv>::value);
            const int32_t chunk_slots = pick_chunk_slots(ctx, wn, plan.n_segs, min_chunk);
            const int64_t n_chunks = (wn + chunk_slots - 1) / chunk_slots;
            const int64_t n_units = n_chunks * plan.n_segs;
            const dim3 grid(unsigned((n_units + kWavesPerBlock - 1) / kWavesPerBlock));
            // dynamic LDS: the converter's tables, then kRowCache weight rows per wave
            const size_t conv_lds = align_up(lds_bytes, 16);
            const int32_t conv_lds_doubles = int32_t(conv_lds / sizeof(double));
            KernelBracket kb(ctx);
            // the MFMA-carrying instantiation exists for the vectorised kernels of converters that opt in (kDenseOk,
            // default yes); everything else reduces dense tiles on the butterfly path - correct, just slower there
            const bool dense = dense_plan;
            auto launch = [&](auto kern, size_t lds_sz) {
                if (debug_occupancy()) {  // $ATLITE_HIP_DEBUG_OCCUPANCY: what the runtime says fits on a CU
                    int nb = -1;
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, kWavesPerBlock * 64, lds_sz);
                    fprintf(stderr, "[atlite-hip] %s: fused kernel %d blocks/CU x %d waves, %zu B LDS/block, grid %u, chunk %d slots\n",
                            what, nb, kWavesPerBlock, lds_sz, grid.x, int(chunk_slots));
                }
                hipLaunchKernelGGL(kern, grid, dim3(kWavesPerBlock * 64), lds_sz, ctx->stream, conv, plan, w0, wn, S,
                                   chunk_slots, n_units, partials, ldp, conv_lds_doubles, n_real);
            };
            // (converters that park a batch's values in LDS - kStageValues - take the early-out kernels' layout: two cached
            // weight rows + eight value rows per wave, 40 KiB per workgroup, four workgroups per CU)
            const bool stage = conv_stage_values<Conv>::value && !dense;
            const size_t lds_base = conv_lds + size_t(kWavesPerBlock) * (dense ? kRowCacheDense + kDenseSlots : stage ? kRowCacheNight + kBatch : kRowCache) * kSegCells * sizeof(double);
            if constexpr (conv_night_pipe<Conv>::value) {
                const size_t lds_night = conv_lds + size_t(kWavesPerBlock) * (kRowCacheNight + kBatch) * kSegCells * sizeof(double);
                bool mapped = false;
                if constexpr (conv_day_map<Conv>::value) {
                    if (conv.in.d_day_map && vec && !dense && !aligned && w0 == 0 && wn == n_slots) {
                        launch(k_fused_segred_night<Conv, true, false, true>, lds_night);
                        mapped = true;
                    }
                }
                if (mapped) {
                } else if (!vec) {
                    if constexpr (kScalarToo) launch(k_fused_segred_night<Conv, false, false>, lds_night);
                } else if constexpr (conv_dense_ok<Conv>::value)
                    dense ? launch(k_fused_segred_night<Conv, true, true>, lds_night) : launch(k_fused_segred_night<Conv, true, false>, lds_night);
                else
                    launch(k_fused_segred_night<Conv, true, false>, lds_night);
            } else if (!vec) {
                if constexpr (kScalarToo) launch(k_fused_segred<Conv, false, false>, lds_base);
            } else if constexpr (conv_dense_ok<Conv>::value) {
                dense ? launch(k_fused_segred<Conv, true, true>, lds_base) : launch(k_fused_segred<Conv, true, false>, lds_base);
            } else {
                launch(k_fused_segred<Conv, true, false>, lds_base);
            }
            if ((rc = check_launch(what))) return rc;
        }
        if (aligned) {
            hipLaunchKernelGGL(k_combine_aligned, dim3(unsigned((wn + 255) / 256), unsigned(plan.n_rows)), dim3(256), 0, ctx->stream, plan, partials, ldp,
                               w0, wn, n_real, series, ld_series);
        } else {
            const dim3 grid(unsigned((wn + 255) / 256), unsigned(N));
            hipLaunchKernelGGL(k_combine, grid, dim3(256), 0, ctx->stream, plan, partials, ldp, wn, series + w0, ld_series);
        }
        if ((rc = check_launch(what))) return rc;
    }
    if (time_agg != ATL_TIME_NONE) {
        hipLaunchKernelGGL(k_rows_timered, dim3(unsigned(N)), dim3(256), 0, ctx->stream, series, ld_series,
                           n_real, time_agg == ATL_TIME_MEAN ? 1 : time_agg == ATL_TIME_SUM_COUNT ? 2 : 0, d_out);
        if ((rc = check_launch(what))) return rc;
    }
    return ATL_OK;
}

// Can the (T, S) fp64 cubes at `ptrs` be read 16 bytes per lane (the lane's two adjacent cells)?  Almost always:
//  * the accesses need not be 16-byte aligned - ld2 / st2 promise 8 bytes, which global_load / store_dwordx4 take; with
//    an odd cell count every other slot's rows sit 8 bytes off;
//  * with an ODD cell count the lane that owns a slot's last cell reads 8 bytes past it: the next slot's first cell or,
//    in the last slot, 8 bytes past the cube.  Those bytes share the last element's 4 KiB page unless the cube ENDS on
//    a page boundary - then, and only then, the launch takes the unvectorised instantiation (stores never overrun: st2).
// (ld = cells between slots: the ld_cells argument of the *_ld entry points.)
bool vec_ok(int64_t T, int64_t S, int64_t ld, std::initializer_list<const void *> ptrs) {
    if (no_vec()) return false;
    for (const void *p : ptrs) {
        if (!p) continue;
        if (!aligned8(p)) return false;
        // the final slot's last cell sits (T - 1) * ld + S cells in, whatever the stride is (an unpadded pool of n cubes,
        // ld = n * S, ends its last cube as abruptly as a contiguous cube does; behind a padded slot's last cell the
        // address is 8 bytes off a 16-byte boundary and never on a page boundary)
        if ((S & 1) && T > 0 && ((reinterpret_cast<uintptr_t>(p) + (size_t(T - 1) * size_t(ld) + size_t(S)) * sizeof(double)) & 4095u) == 0) return false;
    }
    return true;
}

}  // namespace
