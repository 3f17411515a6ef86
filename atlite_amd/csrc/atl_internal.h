// Internal declarations shared by the runtime (atl_runtime.cpp) and the kernels
// (atl_kernels.hip).  Not installed; the public surface is include/atlite_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "atlite_hip.h"

namespace atl {

// ---- geometry of the fused segment-reduce kernel -------------------------------------
constexpr int kLanes = 64;               // gfx950 wavefront
constexpr int kSegCells = 2 * kLanes;    // one wave covers 128 consecutive cells, 2 per lane
constexpr int kBatch = 8;                // output slots reduced per butterfly
#ifndef ATL_WAVES_PER_BLOCK
#define ATL_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = ATL_WAVES_PER_BLOCK;  // waves of a fused-kernel workgroup (independent of each other)
constexpr int kMaxKnots = 1023;          // wind power-curve table limit (LDS: 5 x 1024 doubles = 40 KiB)

void set_error(const char *fmt, ...);

#define ATL_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            atl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),         \
                           __FILE__, __LINE__);                                            \
            return e__ == hipErrorOutOfMemory ? ATL_E_NOMEM : ATL_E_HIP;                   \
        }                                                                                  \
    } while (0)

#define ATL_REQUIRE(cond, ...)                                                             \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            atl::set_error(__VA_ARGS__);                                                   \
            return ATL_E_INVALID;                                                          \
        }                                                                                  \
    } while (0)

// Device view of the aggregation plan (all pointers device-resident).
struct PlanDev {
    int64_t n_rows;        // N shapes
    int64_t n_cells;       // S
    // cell tiles: a segment is a w x h tile (w*h = 128) of the (Y, X) grid, lane l owns the two
    // adjacent cells (y0 + l / (w/2), x0 + 2*(l % (w/2)) + {0,1}).  Unknown grid: Y=1, X=S, 128x1.
    int64_t X, Y;
    int32_t ntx;           // tiles per grid row
    int32_t w2_log2;       // log2(w / 2): lanes per tile row
    int32_t n_segs;        // number of tiles
    int32_t n_prows;       // P partial rows = sum over segments of distinct shapes touching it
    const int32_t *seg_ptr;     // [n_segs+1] partial rows of segment s: [seg_ptr[s], seg_ptr[s+1])
    const double *prow_w;       // [P][kSegCells] weight of local cell, NaN = structurally absent
    const int32_t *shape_ptr;   // [N+1]
    const int32_t *shape_prow;  // [P] partial rows of each shape in ascending segment order
    const uint8_t *row_poison;  // [N] 1 = row has a NaN weight -> output row is NaN
    // Dense tiles (>= kMfmaRows partial rows): their rows are contracted on the matrix cores
    // (v_mfma_f64_16x16x4_f64: 16 rows x 16 slots x 4 cells per instruction).  seg_wm[s] = offset into prow_wm of
    // tile s's operand image, -1 for sparse tiles; per group of 16 rows and K-step k4 (4 cells) one 64-lane A
    // fragment: prow_wm[off + (g * 32 + k4) * 64 + lane] = weight(row 16 g + lane % 16, cell 4 k4 + lane / 16),
    // 0.0 where structurally absent or past the tile's last row.  nullptr when the plan has no dense tile.
    const int64_t *seg_wm;
    const double *prow_wm;
    // [n_segs] bit l = lane l of the tile owns a cell that carries a weight in some partial row.  The other lanes
    // (cells outside every shape: sea, neighbouring countries, tile padding) issue no loads at all - their values
    // could only ever meet structural zeros - so a 128-byte line without a covered cell is never fetched.
    const uint64_t *seg_mask;
    // Line-aligned plan for CONTIGUOUS (T, S) cubes whose slots do not start on 128-byte lines (S % 16 != 0), built by
    // atl_agg_create_aligned; shift_classes = 0 for every other plan.  Slot t starts o = (t * S) % 16 cells into a line,
    // and that offset repeats every p = 16 / gcd(S, 16) slots: class r = the slots r, r + p, ... .  The plan holds p
    // tilings of the same (Y, X) grid, class r's with its tile rows on the line grid of c + o_r (tile_row_lo): tiles
    // [r * shift_tiles, (r + 1) * shift_tiles), output rows [r * shift_rows, (r + 1) * shift_rows) - every tile row is
    // whole lines in every slot of its class.  n_rows is the stacked count, X / Y / ntx / n_cells the grid's own.
    int32_t shift_classes;  // p (0: an ordinary plan)
    int32_t shift_rows;     // N, the matrix's row count
    int32_t shift_tiles;    // tiles per class
    int32_t shift_pad_;
};

constexpr int kMfmaRows = 16;      // rows per MFMA group
constexpr int kMfmaMinRows = 12;   // rows from which a (last, partial) group beats the butterfly path (measured)
// LDS value rows of a wave (one row per slot of the batch, kSegCells doubles): cell c of row i sits at position
// c ^ (4 i), so that the B-fragment read of an MFMA K-step - 16 lanes = 8 rows at the same 4 cells - touches 32
// different banks instead of one, at no padding cost; a lane's own cell pair stays one aligned 16-byte slot.
// (SW = false: plain rows - kernels whose plan has no dense tile never read fragments and keep ONE address)
template <bool SW>
__host__ __device__ inline int vrow_pair(int i, int lane) { return i * kSegCells + 2 * (SW ? lane ^ (2 * i) : lane); }
// MFMA groups of a tile with n partial rows: full groups, plus one more if >= kMfmaMinRows rows remain
__host__ __device__ inline int mfma_groups(int n) {
    return n < kMfmaRows ? 0 : n / kMfmaRows + ((n % kMfmaRows) >= kMfmaMinRows ? 1 : 0);
}

// ---- cell tiles: ONE definition of which cells a lane of a tile owns ---------------------------------
// The stacked cell axis (flat index f = y*X + x) is cut into 128-byte lines of 16 cells.  The tile row
// of grid row y starts at lo(y) = the first cell of the line holding cell (y, 0) and owns the flat
// range [lo(y), lo(y+1)) (the last row: up to S): consecutive, disjoint, line-aligned ranges, so every
// tile row starts on a 128-byte line and a line that straddles grid rows is read by ONE tile row - the
// lowest grid row that starts in it.  (For X < 16 several grid rows start in the same line; all but
// the last of them own nothing.)  Lane l of tile column tx sits at position p = tx*w + 2*(l % (w/2))
// of its row's range.  The kernels call tile_lane_cells(); the plan builder uses the inverse
// (tile_of_cell, atl_runtime.cpp); atl_agg_selfcheck() proves on the host that the two agree and that
// every cell is owned exactly once.
struct TileLane {
    int64_t c0;   // flat index of the lane's first cell (the second is c0 + 1), always inside [0, S]
    bool v0, v1;  // the lane owns c0 / c0 + 1
};

// `o` (0 .. 15): the grid's first cell sits o cells into a 128-byte line - the alignment class of a line-aligned plan
// (PlanDev::shift_classes).  Tile rows then start on the line grid of the SHIFTED index c + o; all indices handed out
// and taken are the grid's own (c0 may be as low as -o: cells before the grid, never owned).
__host__ __device__ inline int64_t tile_row_lo(int64_t X, int64_t y, int64_t o = 0) { return ((y * X + o) & ~int64_t(15)) - o; }

__host__ __device__ inline TileLane tile_lane_cells(int64_t X, int64_t Y, int32_t ntx, int32_t w2_log2,
                                                    int32_t seg, int lane, int64_t o = 0) {
    const int32_t ty = seg / ntx, tx = seg - ty * ntx;
    const int64_t gy = int64_t(ty) * (kLanes >> w2_log2) + (lane >> w2_log2);
    const int64_t p = (int64_t(tx) << (w2_log2 + 1)) + ((lane & ((1 << w2_log2) - 1)) << 1);
    const int64_t hi = gy + 1 < Y ? tile_row_lo(X, gy + 1, o) : X * Y;
    TileLane t;
    t.c0 = tile_row_lo(X, gy, o) + p;
    t.v0 = gy < Y && t.c0 < hi && t.c0 >= 0;
    t.v1 = gy < Y && t.c0 + 1 < hi && t.c0 + 1 >= 0;
    return t;
}

// CSR-style offsets p[0..n] handed in by a caller: non-negative and non-decreasing, so that none of them leads past the
// p[n] entries the arrays they index are promised to hold
inline bool offsets_ok(const int64_t *p, int64_t n) {
    if (p[0] < 0) return false;
    for (int64_t i = 0; i < n; ++i)
        if (p[i + 1] < p[i]) return false;
    return true;
}

// every vertex coordinate of every ring of shape s is finite (offsets already validated).  NaN / inf vertices are
// invalid geometry for the reference's shapely too; both indicator-matrix algorithms give such a shape an EMPTY row -
// std::min / std::max drop a NaN operand, so a NaN in the MIDDLE of a ring would otherwise leave a valid-looking
// bounding box around a ring with two broken edges
inline bool shape_is_finite(int64_t s, const int64_t *shape_ring_ptr, const int64_t *ring_ptr, const double *xy) {
    for (int64_t r = shape_ring_ptr[s]; r < shape_ring_ptr[s + 1]; ++r)
        for (int64_t v = 2 * ring_ptr[r]; v < 2 * ring_ptr[r + 1]; ++v)
            if (!__builtin_isfinite(xy[v])) return false;
    return true;
}

// floor(v) as a grid index clamped to [lo, hi]: coordinates come from the caller's polygons, and converting a double
// beyond the 64-bit range (1e300, +-inf) to an integer is undefined; NaN gives lo (callers skip NaN boxes beforehand)
inline int64_t clamped_floor(double v, int64_t lo, int64_t hi) {
    if (!(v > double(lo))) return lo;
    if (v >= double(hi)) return hi;
    return int64_t(std::floor(v));
}

// ---- wind power-curve table (host) ---------------------------------------------------------------------
// Padded size of a table of n knots: the sizes with an unrolled search are 16, 32 and 128 (a power of two
// > n in any case, so that V[n..n_pad) = +inf terminates every probe sequence).
inline int wind_table_pad(int n) {
    int n_pad = n < 16 ? 16 : n < 32 ? 32 : 128;
    while (n_pad <= n) n_pad *= 2;
    return n_pad;
}

// tbl = V[n_pad] | K[n_pad][4] = {V[j], F[j], slope[j], 0}, the layout interp_padded() (atl_math.h) reads.
// Returns the number of knots in the table (-1 if V is not non-decreasing) and its padded size;
// *finite = every knot, value and slope is finite.  A REPEATED FIRST knot gets a guard knot one ulp
// below it carrying F[0]: np.interp answers F[0] left of the table but the upper duplicate's value AT the
// knot, and the clamped search could not tell the two apart otherwise.
inline int wind_table_build(const double *V, const double *F, int n, std::vector<double> &tbl, int *n_pad_out,
                            bool *finite) {
    std::vector<double> v(V, V + n), f(F, F + n);
    for (int i = 1; i < n; ++i)
        if (!(v[size_t(i)] >= v[size_t(i - 1)])) return -1;
    if (n >= 2 && v[1] == v[0] && __builtin_isfinite(v[0])) {
        v.insert(v.begin(), __builtin_nextafter(v[0], -__builtin_inf()));
        f.insert(f.begin(), f[0]);
        ++n;
    }
    const int n_pad = wind_table_pad(n);
    tbl.assign(size_t(5) * size_t(n_pad), 0.0);
    *finite = true;
    for (int i = 0; i < n_pad; ++i) tbl[size_t(i)] = __builtin_inf();
    for (int i = 0; i < n; ++i) {
        tbl[size_t(i)] = v[size_t(i)];
        double *k = &tbl[size_t(n_pad) + 4 * size_t(i)];
        k[0] = v[size_t(i)];
        k[1] = f[size_t(i)];
        // slope as numpy precomputes it; only the upper one of repeated knots is ever selected
        k[2] = (i + 1 < n && v[size_t(i + 1)] > v[size_t(i)]) ? (f[size_t(i + 1)] - f[size_t(i)]) / (v[size_t(i + 1)] - v[size_t(i)]) : 0.0;
        *finite = *finite && __builtin_isfinite(k[0]) && __builtin_isfinite(k[1]) && __builtin_isfinite(k[2]);
    }
    if (n_pad_out) *n_pad_out = n_pad;
    return n;
}

// Grid-aligned variant of the table (interp_grid, atl_math.h): returns the number of buckets, 0 if the knots
// are not all non-negative multiples of a w in {1, 1/2, 1/4, 1/8} (or the table is not finite / too long).
// v, f: the knots AFTER wind_table_build's guard-knot insertion (a guard knot is never grid-aligned).
constexpr int kMaxWindBuckets = 512;
inline int wind_grid_build(const double *tbl_search, int n, int n_pad, std::vector<double> &grid, double *inv_w_out,
                           int *b0_out) {
    const double *V = tbl_search;
    const double *K = tbl_search + n_pad;
    if (n < 1 || !(V[0] >= 0.0)) return 0;
    for (double inv_w : {1.0, 2.0, 4.0, 8.0}) {
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const double t = V[i] * inv_w;
            ok = __builtin_isfinite(t) && t == __builtin_floor(t) && t < 1e6;
        }
        if (!ok) continue;
        const int b0 = int(V[0] * inv_w), b1 = int(V[n - 1] * inv_w);
        const int n_b = b1 - b0 + 1;
        if (n_b > kMaxWindBuckets) return 0;
        grid.assign(size_t(4) * size_t(n_b), 0.0);
        int j = 0;
        for (int b = 0; b < n_b; ++b) {
            const double start = double(b + b0) / inv_w;  // exact
            while (j + 1 < n && V[j + 1] <= start) ++j;
            for (int q = 0; q < 3; ++q) grid[size_t(4 * b + q)] = K[4 * j + q];
        }
        *inv_w_out = inv_w;
        *b0_out = b0;
        return n_b;
    }
    return 0;
}

inline int64_t tile_columns(int64_t X, int64_t Y, int w2_log2, bool shifted = false) {
    const int w = 2 << w2_log2;
    const int64_t max_shift = ((Y > 1 && X % 16 != 0) || shifted) ? 15 : 0;  // (shifted: a line-aligned plan's classes)
    return (X - 1 + max_shift) / w + 1;
}

}  // namespace atl

struct atl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;  // created on first use (atl_upload_async)
    // scratch arena (grown on demand, stream-ordered reuse)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // small table buffer for per-call host tables (wind knots): device copy + pinned host staging;
    // ev_table marks the completion of the last H2D so the staging buffer is never overwritten early
    double *d_table = nullptr;
    double *h_table = nullptr;
    hipEvent_t ev_table = nullptr;
    bool table_pending = false;
    // timing
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // atl_timer_*
    // dominant-kernel brackets: a ring of event pairs (atl_set_profiling(ctx, n)), so a benchmark loop
    // can time every launch of a run without a host sync per step
    std::vector<hipEvent_t> ev_ring;  // 2 * ring size: {start, stop} per slot
    int64_t ring_count = 0;           // launches bracketed since profiling was enabled
    bool profiling = false;
    int n_cu = 256;
    // slots of the INPUT cubes of the next conversion calls are this many cells apart (the ld_cells argument of the *_ld entry points, set for the duration of a call: StrideScope); 0 = the
    // cubes are contiguous (T, S).  The library's own device copies of a cutout pad every slot to a 128-byte line.
    int64_t slot_stride = 0;
    // atl_capture_begin .. atl_capture_end: the calls in between are recorded into a hipGraph, not executed
    bool capturing = false;
    // file / narrow-dtype ingest (atl_ingest.hip): staging buffers, created on first use
    void *ingest = nullptr;
    void (*ingest_free)(void *) = nullptr;
    // two page-locked bounce buffers (created on first use): every transfer between device memory and host memory that is
    // NOT page-locked goes through them (atl::h2d / atl::d2h), so the runtime never has to pin a caller's heap or stack
    // pages on the fly
    uint8_t *bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    bool bounce_busy[2] = {false, false};
    int bounce_next = 0;
};

struct atl_event {
    hipEvent_t ev = nullptr;
    int device = 0;
};

struct atl_graph {
    hipGraphExec_t exec = nullptr;
    int device = 0;
};

struct atl_agg {
    atl_ctx *ctx = nullptr;
    atl::PlanDev dev{};
    std::vector<void *> allocs;
};

namespace atl {
// scratch: returns a device pointer valid until the next scratch_reserve on this ctx.
int scratch_reserve(atl_ctx *ctx, size_t bytes, void **out);
// the context's copy stream (created on first use)
int copy_stream_of(atl_ctx *ctx, hipStream_t *out);
// Every device allocation of the library goes through these two.  Normally hipMalloc / hipFree.  With
// $ATLITE_HIP_FENCE=1 (a debugging mode, tools/hang_hunt.sh) each block is its own virtual-memory mapping whose LAST
// byte is the last mapped byte before an unmapped guard range (8-byte granularity; $ATLITE_HIP_FENCE_SLACK bytes may
// stay mapped behind it), with another guard range in
// front, and a freed block's addresses are never handed out again: an access one element past either end, or after
// the free, is a page fault on the spot instead of a silent read of a neighbouring allocation.  dev_free does not
// order anything: callers synchronise first, as they had to for hipFree's sake.
// atl_ingest.hip: settle the context's device-inflate reads (no-op when there are none)
int ingest_finish(atl_ctx *ctx);
// atl_ingest.hip: order the copy stream `cs` behind the device-inflate reads in flight (device-side wait, no verdicts read)
int ingest_join(atl_ctx *ctx, hipStream_t cs);
// Transfers between device memory and ARBITRARY host memory (a caller's NumPy array, a std::vector, a stack variable).  The
// HIP runtime serves such a copy by pinning the host pages on the fly and letting a copy engine or a blit kernel touch them;
// three unexplained "Memory access fault by GPU ... on address <a host heap address>" aborts of the test suite (rounds 3-5, one
// in ~15 whole-suite runs, always in a process that had run for a minute and freed many small arrays) ended when the library
// stopped doing that: host memory that is not page-locked (hipHostMalloc / hipHostRegister) is copied through the context's
// own page-locked bounce buffers, 4 MiB at a time, double-buffered.  h2d returns once the source has been read (the device
// side completes in stream order); d2h returns with the data in place.
int h2d(atl_ctx *ctx, hipStream_t st, void *d_dst, const void *h_src, size_t bytes);
int d2h(atl_ctx *ctx, hipStream_t st, void *h_dst, const void *d_src, size_t bytes);
int h2d_2d(atl_ctx *ctx, hipStream_t st, void *d_dst, size_t dst_pitch, const void *h_src, size_t src_pitch, size_t width, size_t height);
int d2h_2d(atl_ctx *ctx, hipStream_t st, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch, size_t width, size_t height);
bool host_is_pinned(const void *p);
hipError_t dev_malloc(void **out, size_t bytes);
hipError_t dev_free(void *p);
bool fence_mode();
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// cells between the slots of a call's input cubes: the context's stride if one is set, else the cell count
inline int64_t slot_stride_of(const atl_ctx *ctx, int64_t S) { return ctx->slot_stride > 0 ? ctx->slot_stride : S; }
}  // namespace atl
