// Post-processing of the small (rows x time) result of runoff() on the device:
// rolling mean over time (min_periods, NaN-skipping), a quantile of all values (linear interpolation between order
// statistics), "below the threshold -> 0", and the scaling of every row to a reported total over a set of time steps.
// Reference: atlite/convert.py:1046-1082 (xarray rolling(...).mean() -> bottleneck.move_mean, pandas Series.quantile,
// DataArray.where, the yearly normalisation).  The series is time-contiguous per row (what the fused kernels write), a
// few MB: every kernel here is latency-, not bandwidth-bound, and is laid out for short dependent chains.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "atl_internal.h"

using namespace atl;

namespace {

__device__ __forceinline__ bool finite_d(double x) { return __builtin_fabs(x) < __builtin_inf(); }

// ---- rolling mean ---------------------------------------------------------------------------------------------
// out[r][t] = mean of the finite values among in[r][t - w + 1 .. t] if there are at least min_periods of them, else NaN.
// A thread owns kRollSeg consecutive outputs of one row: it sums the window of its first output directly and slides
// from there with compensated additions and removals (Kahan, one compensation term each way - the scheme of pandas'
// roll_mean; bottleneck's move_mean, what xarray calls, slides without compensation over the whole row).  Restarting
// every kRollSeg steps bounds the drift by the segment instead of the row and makes the rows' time axis parallel.
// Clamps as pandas applies them: a window whose values are all equal returns that value exactly; a mean of
// non-negative values is not negative (and vice versa).  +-inf counts as missing (pandas' rolling does the same;
// bottleneck would poison the rest of the row with inf - inf).
constexpr int kRollSeg = 256;

struct RollState {
    double sum = 0.0, c_add = 0.0, c_rem = 0.0, prev = 0.0;
    int64_t nobs = 0, neg = 0, same = 0;
    __device__ __forceinline__ void add(double v) {
        if (!finite_d(v)) return;
        ++nobs;
        const double y = v - c_add, t = sum + y;
        c_add = (t - sum) - y;
        sum = t;
        if (__builtin_signbit(v)) ++neg;
        same = (v == prev && same > 0) ? same + 1 : 1;
        prev = v;
    }
    __device__ __forceinline__ void remove(double v) {
        if (!finite_d(v)) return;
        --nobs;
        const double y = -v - c_rem, t = sum + y;
        c_rem = (t - sum) - y;
        sum = t;
        if (__builtin_signbit(v)) --neg;
    }
    __device__ __forceinline__ double mean(int64_t min_periods) const {
        if (nobs < min_periods || nobs <= 0) return __builtin_nan("");
        double r = sum / double(nobs);
        if (same >= nobs)
            r = prev;
        else if (neg == 0 && r < 0.0)
            r = 0.0;
        else if (neg == nobs && r > 0.0)
            r = 0.0;
        return r;
    }
};

__global__ __launch_bounds__(64) void k_rolling_mean(const double *__restrict__ in, int64_t rows, int64_t T, int64_t ld_in,
                                                     int64_t window, int64_t min_periods, double *__restrict__ out,
                                                     int64_t ld_out, int64_t n_seg) {
    const int64_t id = int64_t(blockIdx.x) * 64 + threadIdx.x;
    if (id >= rows * n_seg) return;
    // consecutive lanes take consecutive ROWS of the same segment: a wave's loads of one step touch 64 rows' lines, and
    // each lane walks its own lines through L1 / L2 for the whole segment
    const int64_t seg = id / rows, r = id - seg * rows;
    const double *x = in + r * ld_in;
    double *y = out + r * ld_out;
    const int64_t t0 = seg * kRollSeg, t1 = min(t0 + int64_t(kRollSeg), T);
    RollState s;
    for (int64_t t = max(int64_t(0), t0 - window + 1); t <= t0; ++t) s.add(x[t]);
    y[t0] = s.mean(min_periods);
    for (int64_t t = t0 + 1; t < t1; ++t) {
        s.add(x[t]);
        if (t - window >= 0) s.remove(x[t - window]);
        y[t] = s.mean(min_periods);
    }
}

// ---- order statistics: radix select on the order-preserving image of the doubles -----------------------------------------
__device__ __forceinline__ uint64_t key_of(double v) {
    const uint64_t u = uint64_t(__double_as_longlong(v));
    return (u >> 63) ? ~u : (u | (uint64_t(1) << 63));
}
__device__ __forceinline__ double value_of(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & ~(uint64_t(1) << 63)) : ~k;
    return __longlong_as_double((long long)u);
}

constexpr int kSelBits = 11, kSelBins = 1 << kSelBits, kSelPasses = 6;  // 6 x 11 = 66 >= 64 bits (the last pass has 9)

// state[0] = prefix of the selected key (bits above `shift` fixed), state[1] = rank still to find inside the prefix,
// state[2] = number of non-NaN values, state[3] = the selected key, state[4] = values <= selected, state[5] = smallest key above
struct SelState {
    unsigned long long prefix, rank, n, key, n_le, next;
};

__global__ __launch_bounds__(256) void k_sel_count(const double *__restrict__ in, int64_t rows, int64_t T, int64_t ld,
                                                   SelState *st) {
    // non-NaN count (block-wide reduction, one atomic per block)
    __shared__ unsigned long long part[256];
    unsigned long long c = 0;
    const int64_t n = rows * T;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const double v = in[(i / T) * ld + (i % T)];
        c += (v == v) ? 1 : 0;
    }
    part[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicAdd(&st->n, part[0]);
}

__global__ __launch_bounds__(256) void k_sel_hist(const double *__restrict__ in, int64_t rows, int64_t T, int64_t ld,
                                                  const SelState *st, int shift, int bits, unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[kSelBins];
    for (int i = threadIdx.x; i < kSelBins; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix;
    const int hi = shift + bits;  // bits [hi, 64) of a candidate equal the prefix's
    const int64_t n = rows * T;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const double v = in[(i / T) * ld + (i % T)];
        if (v != v) continue;
        const uint64_t k = key_of(v);
        if (hi < 64 && (k >> hi) != (prefix >> hi)) continue;
        atomicAdd(&h[(k >> shift) & ((1u << bits) - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one block: the bin that holds the wanted rank; clears the histogram for the next pass
__global__ __launch_bounds__(256) void k_sel_pick(SelState *st, int shift, int bits, unsigned int *hist) {
    __shared__ unsigned long long cum[256];
    const int per = kSelBins / 256;  // bins per thread, ascending
    unsigned long long mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[threadIdx.x * per + j];
    cum[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long rank = st->rank, before = 0;
        int t = 0;
        while (t < 255 && before + cum[t] <= rank) before += cum[t++];
        int b = t * per;
        while (b < kSelBins - 1 && before + hist[b] <= rank) before += hist[b++];
        st->rank = rank - before;
        st->prefix |= (unsigned long long)(b) << shift;
        if (shift == 0) st->key = st->prefix;
    }
    __syncthreads();
    for (int j = 0; j < per; ++j) hist[threadIdx.x * per + j] = 0;
    (void)bits;
}

// values <= the selected one, and the smallest key above it (the next order statistic when the selected value is not repeated)
__global__ __launch_bounds__(256) void k_sel_next(const double *__restrict__ in, int64_t rows, int64_t T, int64_t ld, SelState *st) {
    __shared__ unsigned long long s_le[256], s_nx[256];
    const unsigned long long key = st->key;
    unsigned long long le = 0, nx = ~0ull;
    const int64_t n = rows * T;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const double v = in[(i / T) * ld + (i % T)];
        if (v != v) continue;
        const unsigned long long k = key_of(v);
        if (k <= key)
            ++le;
        else if (k < nx)
            nx = k;
    }
    s_le[threadIdx.x] = le;
    s_nx[threadIdx.x] = nx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) {
            s_le[threadIdx.x] += s_le[threadIdx.x + w];
            s_nx[threadIdx.x] = min(s_nx[threadIdx.x], s_nx[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (s_le[0]) atomicAdd(&st->n_le, s_le[0]);
        atomicMin(&st->next, s_nx[0]);
    }
}

__global__ void k_sel_values(const SelState *st, double *out) {
    out[0] = value_of(st->key);
    out[1] = st->next == ~0ull ? __builtin_nan("") : value_of(st->next);
}

// ---- elementwise / per-row -------------------------------------------------------------------------------------------
// result.where(result >= threshold, 0.0): NaN compares false and becomes 0 as well (convert.py:1062)
__global__ __launch_bounds__(256) void k_zero_below(double *__restrict__ d, int64_t rows, int64_t T, int64_t ld, double thr) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= rows * T) return;
    double *p = d + (i / T) * ld + (i % T);
    const double v = *p;
    *p = (v >= thr) ? v : 0.0;
}

// row r: fac = ref[r] / nan-skipping sum of d[r][t] over the steps with mask[t] != 0; d[r][:] *= fac  (convert.py:1078-1081)
__global__ __launch_bounds__(256) void k_normalize_rows(double *__restrict__ d, int64_t T, int64_t ld, const uint8_t *__restrict__ mask,
                                                        const double *__restrict__ ref) {
    __shared__ double ss[256];
    double *row = d + int64_t(blockIdx.x) * ld;
    double s = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        const double v = row[t];
        if (mask[t] && v == v) s += v;
    }
    ss[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) ss[threadIdx.x] += ss[threadIdx.x + w];
        __syncthreads();
    }
    const double fac = ref[blockIdx.x] / ss[0];
    for (int64_t t = threadIdx.x; t < T; t += 256) row[t] *= fac;
}

int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

}  // namespace

extern "C" {

int atl_rolling_mean(atl_ctx *ctx, const double *d_in, int64_t rows, int64_t T, int64_t ld_in, int64_t window,
                     int64_t min_periods, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && rows >= 0 && T >= 0, "atl_rolling_mean: bad argument");
    ATL_REQUIRE(window >= 1, "atl_rolling_mean: window must be >= 1 (got %lld)", (long long)window);
    ATL_REQUIRE(min_periods >= 0 && min_periods <= window, "atl_rolling_mean: min_periods %lld must be in [0, window]", (long long)min_periods);
    ATL_REQUIRE(ld_in >= T && ld_out >= T, "atl_rolling_mean: a row stride is smaller than the row");
    if (rows == 0 || T == 0) return ATL_OK;
    ATL_REQUIRE(d_in && d_out && d_in != d_out, "atl_rolling_mean: d_in / d_out is NULL, or the same buffer (out of place only)");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t n_seg = (T + kRollSeg - 1) / kRollSeg;
    hipLaunchKernelGGL(k_rolling_mean, dim3(unsigned((rows * n_seg + 63) / 64)), dim3(64), 0, ctx->stream, d_in, rows, T, ld_in, window,
                       min_periods, d_out, ld_out, n_seg);
    return launched("atl_rolling_mean");
}

int atl_order_statistic(atl_ctx *ctx, const double *d_in, int64_t rows, int64_t T, int64_t ld, double q, int64_t *h_n,
                        double *h_pair, int64_t *h_rank, int64_t *h_n_le) {
    ATL_REQUIRE(ctx && rows >= 0 && T >= 0 && ld >= T && h_n && h_pair, "atl_order_statistic: bad argument");
    ATL_REQUIRE(q >= 0.0 && q <= 1.0, "percentiles should all be in the interval [0, 1]");
    *h_n = 0;
    h_pair[0] = h_pair[1] = __builtin_nan("");
    if (h_rank) *h_rank = 0;
    if (h_n_le) *h_n_le = 0;
    if (rows * T == 0) return ATL_OK;
    ATL_REQUIRE(d_in, "atl_order_statistic: d_in is NULL");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    // scratch: state | histogram | the two values
    void *scr = nullptr;
    const size_t bytes = 256 + kSelBins * sizeof(unsigned int) + 64;
    int rc = scratch_reserve(ctx, bytes, &scr);
    if (rc) return rc;
    SelState *st = static_cast<SelState *>(scr);
    unsigned int *hist = reinterpret_cast<unsigned int *>(static_cast<char *>(scr) + 256);
    double *vals = reinterpret_cast<double *>(static_cast<char *>(scr) + 256 + kSelBins * sizeof(unsigned int));
    ATL_HIP_TRY(hipMemsetAsync(scr, 0, bytes, ctx->stream));
    const unsigned grid = unsigned(std::min<int64_t>((rows * T + 255) / 256, int64_t(ctx->n_cu) * 8));
    hipLaunchKernelGGL(k_sel_count, dim3(grid), dim3(256), 0, ctx->stream, d_in, rows, T, ld, st);
    if ((rc = launched("atl_order_statistic"))) return rc;
    SelState h{};
    if ((rc = d2h(ctx, ctx->stream, &h, st, sizeof(h)))) return rc;
    *h_n = int64_t(h.n);
    if (h.n == 0) return ATL_OK;
    // numpy's "linear" method: virtual index q (n - 1); the order statistics floor(.) and floor(.) + 1 bracket it
    const double pos = q * double(h.n - 1);
    int64_t lo = int64_t(std::floor(pos));
    lo = std::min<int64_t>(std::max<int64_t>(lo, 0), int64_t(h.n) - 1);
    h.rank = (unsigned long long)lo;
    h.next = ~0ull;
    if ((rc = h2d(ctx, ctx->stream, st, &h, sizeof(h)))) return rc;
    for (int p = 0; p < kSelPasses; ++p) {
        const int hi = 64 - p * kSelBits;
        const int shift = std::max(hi - kSelBits, 0), bits = hi - shift;
        hipLaunchKernelGGL(k_sel_hist, dim3(grid), dim3(256), 0, ctx->stream, d_in, rows, T, ld, st, shift, bits, hist);
        hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, ctx->stream, st, shift, bits, hist);
    }
    hipLaunchKernelGGL(k_sel_next, dim3(grid), dim3(256), 0, ctx->stream, d_in, rows, T, ld, st);
    hipLaunchKernelGGL(k_sel_values, dim3(1), dim3(1), 0, ctx->stream, st, vals);
    if ((rc = launched("atl_order_statistic"))) return rc;
    double pair[2];
    if ((rc = d2h(ctx, ctx->stream, pair, vals, sizeof(pair)))) return rc;
    if ((rc = d2h(ctx, ctx->stream, &h, st, sizeof(h)))) return rc;
    h_pair[0] = pair[0];
    // the order statistic after x_(lo): x_(lo) itself when it is repeated beyond rank lo, else the smallest value above
    h_pair[1] = (int64_t(h.n_le) > lo + 1) ? pair[0] : pair[1];
    if (h_rank) *h_rank = lo;
    if (h_n_le) *h_n_le = int64_t(h.n_le);
    return ATL_OK;
}

int atl_zero_below(atl_ctx *ctx, double *d, int64_t rows, int64_t T, int64_t ld, double threshold) {
    ATL_REQUIRE(ctx && rows >= 0 && T >= 0 && ld >= T, "atl_zero_below: bad argument");
    if (rows * T == 0) return ATL_OK;
    ATL_REQUIRE(d, "atl_zero_below: d is NULL");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_zero_below, dim3(unsigned((rows * T + 255) / 256)), dim3(256), 0, ctx->stream, d, rows, T, ld, threshold);
    return launched("atl_zero_below");
}

int atl_normalize_rows(atl_ctx *ctx, double *d, int64_t rows, int64_t T, int64_t ld, const uint8_t *d_time_mask,
                       const double *d_ref) {
    ATL_REQUIRE(ctx && rows >= 0 && T >= 0 && ld >= T, "atl_normalize_rows: bad argument");
    if (rows == 0) return ATL_OK;
    ATL_REQUIRE(d && d_ref && (T == 0 || d_time_mask), "atl_normalize_rows: a pointer is NULL");
    ATL_REQUIRE(rows < (int64_t(1) << 31), "atl_normalize_rows: too many rows");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_normalize_rows, dim3(unsigned(rows)), dim3(256), 0, ctx->stream, d, T, ld, d_time_mask, d_ref);
    return launched("atl_normalize_rows");
}

}  // extern "C"
