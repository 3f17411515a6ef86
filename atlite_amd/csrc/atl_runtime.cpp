// Runtime half of libatlite_hip.so: contexts, device memory, timing, and the host-side
// construction of the aggregation plan (indicator matrix -> segment-local layout).
// Boundary: include/atlite_hip.h.  Reference semantics: atlite/convert.py:213-262 (matrix
// handling), atlite/aggregate.py:16-35 (the product that the plan implements).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include <execinfo.h>
#include <signal.h>
#include <fcntl.h>
#include <unistd.h>

#include "atl_internal.h"
#include "atl_math.h"

// The pv converter math (free functions of atl_conv_pv.h) is __host__ __device__; included here the
// way atl_kernels.hip includes it, so that atl_pv_probe_host() runs the kernels' own source on the CPU.
#ifndef ATL_PV_GROUP
#define ATL_PV_GROUP 1
#endif
namespace {
using namespace atl;
#include "atl_device_util.h"
#include "atl_conv_basic.h"
#include "atl_conv_wind.h"
#include "atl_conv_pv.h"
}  // namespace

namespace atl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- device memory: hipMalloc, or fenced virtual-memory mappings ($ATLITE_HIP_FENCE=1) ---------------------------
bool fence_mode() {
    static const bool on = [] {
        const char *e = getenv("ATLITE_HIP_FENCE");
        return e && *e && *e != '0';
    }();
    return on;
}

namespace {
struct FenceBlock {
    char *va;        // start of the reservation (guard | mapping | guard), never released: a stale pointer must keep faulting
    size_t mapped;   // bytes mapped at va + guard
    hipMemGenericAllocationHandle_t handle;
    int device;
};
std::mutex g_fence_m;
std::unordered_map<void *, FenceBlock> g_fence;
}  // namespace

hipError_t dev_malloc(void **out, size_t bytes) {
    if (!fence_mode()) return hipMalloc(out, bytes);
    *out = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    if (gran == 0) return hipErrorNotSupported;
    // $ATLITE_HIP_FENCE_SLACK bytes (default 0) stay mapped behind the block: 8 lets the vectorised kernels run on cubes with
    // an odd cell count (their documented 8-byte over-read of the last pair, vec_ok) while anything further still faults
    static const size_t slack = [] {
        const char *e = getenv("ATLITE_HIP_FENCE_SLACK");
        return e ? size_t(std::max(0, atoi(e))) / 8 * 8 : size_t(0);
    }();
    const size_t want = align_up(bytes ? bytes : 8, 8) + slack, mapped = align_up(want, gran);
    FenceBlock b{};
    b.mapped = mapped;
    b.device = dev;
    void *va = nullptr;
    if ((e = hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0)) != hipSuccess) return e;
    b.va = static_cast<char *>(va);
    if ((e = hipMemCreate(&b.handle, mapped, &prop, 0)) != hipSuccess) {
        (void)hipMemAddressFree(va, mapped + 2 * gran);
        return e;
    }
    if ((e = hipMemMap(b.va + gran, mapped, 0, b.handle, 0)) != hipSuccess) {
        (void)hipMemRelease(b.handle);
        (void)hipMemAddressFree(va, mapped + 2 * gran);
        return e;
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(b.va + gran, mapped, &acc, 1)) != hipSuccess) {
        (void)hipMemUnmap(b.va + gran, mapped);
        (void)hipMemRelease(b.handle);
        (void)hipMemAddressFree(va, mapped + 2 * gran);
        return e;
    }
    void *user = b.va + gran + (mapped - want);  // the block ends where the mapping ends
    {
        std::lock_guard<std::mutex> lk(g_fence_m);
        g_fence[user] = b;
    }
    *out = user;
    return hipSuccess;
}

hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    if (!fence_mode()) return hipFree(p);
    FenceBlock b;
    {
        std::lock_guard<std::mutex> lk(g_fence_m);
        auto it = g_fence.find(p);
        if (it == g_fence.end()) return hipErrorInvalidValue;  // not ours, or freed twice
        b = it->second;
        g_fence.erase(it);
    }
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != b.device) (void)hipSetDevice(b.device);  // the block's device, not whichever the calling thread last used
    hipError_t e = hipDeviceSynchronize();  // what hipFree does implicitly
    size_t gran = 0;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = b.device;
    (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    hipError_t e2 = hipMemUnmap(b.va + gran, b.mapped);
    hipError_t e3 = hipMemRelease(b.handle);
    if (cur >= 0 && cur != b.device) (void)hipSetDevice(cur);
    // the reservation stays: the addresses are never reused
    return e != hipSuccess ? e : e2 != hipSuccess ? e2 : e3;
}

int scratch_reserve(atl_ctx *ctx, size_t bytes, void **out) {
    bytes = align_up(bytes ? bytes : 256, 256);
    if (bytes > ctx->scratch_bytes) {
        if (ctx->capturing) {
            set_error("the scratch arena would have to grow (%zu -> %zu bytes) inside atl_capture_begin / atl_capture_end: "
                      "run the sequence once before capturing it", ctx->scratch_bytes, bytes);
            return ATL_E_INVALID;
        }
        // stream-ordered: earlier kernels may still read the old arena
        ATL_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) ATL_HIP_TRY(dev_free(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = dev_malloc(&ctx->scratch, want);
        if (e != hipSuccess) {
            want = bytes;
            ATL_HIP_TRY(dev_malloc(&ctx->scratch, want));
        }
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return ATL_OK;
}

}  // namespace atl

using namespace atl;

template <class T>
static int to_device(atl_agg *a, const std::vector<T> &v, const T **out) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    ATL_HIP_TRY(dev_malloc(&d, bytes));
    a->allocs.push_back(d);
    if (!v.empty()) {
        int rc = atl::h2d(a->ctx, a->ctx->stream, d, v.data(), v.size() * sizeof(T));
        if (rc) return rc;
        ATL_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    }
    *out = static_cast<const T *>(d);
    return ATL_OK;
}

namespace atl {
int copy_stream_of(atl_ctx *ctx, hipStream_t *out) {
    if (!ctx->copy_stream) {
        ATL_HIP_TRY(hipSetDevice(ctx->device));
        ATL_HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    }
    *out = ctx->copy_stream;
    return ATL_OK;
}
}  // namespace atl

namespace atl {
// ---- host <-> device through the context's page-locked bounce buffers (atl_internal.h) -----------------------------------
constexpr size_t kBounce = size_t(4) << 20;

bool host_is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // "not a registered pointer" is an answer, not an error
    return pinned;
}

// the whole range [p, p + bytes): a view that runs past a registered array (first byte page-locked, last byte not) must take
// the bounce buffers, or the runtime pins its tail on the fly - the very thing they exist to avoid
static bool host_range_is_pinned(const void *p, size_t bytes) {
    return host_is_pinned(p) && (bytes <= 1 || host_is_pinned(static_cast<const uint8_t *>(p) + bytes - 1));
}

static int bounce_slot(atl_ctx *ctx, int *slot) {
    const int s = ctx->bounce_next;
    ctx->bounce_next ^= 1;
    if (!ctx->bounce[s]) {
        ATL_HIP_TRY(hipSetDevice(ctx->device));
        ATL_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->bounce[s]), kBounce, hipHostMallocDefault));
        ATL_HIP_TRY(hipEventCreateWithFlags(&ctx->bounce_ev[s], hipEventDisableTiming));
    }
    if (ctx->bounce_busy[s]) {  // the transfer that last used this buffer has left it
        ATL_HIP_TRY(hipEventSynchronize(ctx->bounce_ev[s]));
        ctx->bounce_busy[s] = false;
    }
    *slot = s;
    return ATL_OK;
}

int h2d(atl_ctx *ctx, hipStream_t st, void *d_dst, const void *h_src, size_t bytes) {
    if (!bytes) return ATL_OK;
    if (host_range_is_pinned(h_src, bytes)) {
        ATL_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        return ATL_OK;
    }
    for (size_t off = 0; off < bytes; off += kBounce) {
        const size_t n = std::min(kBounce, bytes - off);
        int s;
        int rc = bounce_slot(ctx, &s);
        if (rc) return rc;
        memcpy(ctx->bounce[s], static_cast<const uint8_t *>(h_src) + off, n);
        ATL_HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + off, ctx->bounce[s], n, hipMemcpyHostToDevice, st));
        ATL_HIP_TRY(hipEventRecord(ctx->bounce_ev[s], st));
        ctx->bounce_busy[s] = true;
    }
    return ATL_OK;
}

int d2h(atl_ctx *ctx, hipStream_t st, void *h_dst, const void *d_src, size_t bytes) {
    if (!bytes) return ATL_OK;
    if (host_range_is_pinned(h_dst, bytes)) {
        ATL_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        ATL_HIP_TRY(hipStreamSynchronize(st));
        return ATL_OK;
    }
    // slice i + 1 travels while slice i is copied out of its buffer
    int cur = -1;
    size_t cur_off = 0, cur_n = 0;
    for (size_t off = 0; off < bytes || cur >= 0; off += kBounce) {
        int nxt = -1;
        size_t nxt_n = 0;
        if (off < bytes) {
            nxt_n = std::min(kBounce, bytes - off);
            int rc = bounce_slot(ctx, &nxt);
            if (rc) return rc;
            ATL_HIP_TRY(hipMemcpyAsync(ctx->bounce[nxt], static_cast<const uint8_t *>(d_src) + off, nxt_n, hipMemcpyDeviceToHost, st));
            ATL_HIP_TRY(hipEventRecord(ctx->bounce_ev[nxt], st));
            ctx->bounce_busy[nxt] = true;
        }
        if (cur >= 0) {
            ATL_HIP_TRY(hipEventSynchronize(ctx->bounce_ev[cur]));
            ctx->bounce_busy[cur] = false;
            memcpy(static_cast<uint8_t *>(h_dst) + cur_off, ctx->bounce[cur], cur_n);
        }
        cur = nxt;
        cur_off = off;
        cur_n = nxt_n;
    }
    return ATL_OK;
}

int h2d_2d(atl_ctx *ctx, hipStream_t st, void *d_dst, size_t dst_pitch, const void *h_src, size_t src_pitch, size_t width, size_t height) {
    if (!width || !height) return ATL_OK;
    if (host_range_is_pinned(h_src, (height - 1) * src_pitch + width)) {
        ATL_HIP_TRY(hipMemcpy2DAsync(d_dst, dst_pitch, h_src, src_pitch, width, height, hipMemcpyHostToDevice, st));
        return ATL_OK;
    }
    if (width > kBounce) {  // rows longer than a buffer: row by row
        for (size_t r = 0; r < height; ++r) {
            int rc = h2d(ctx, st, static_cast<uint8_t *>(d_dst) + r * dst_pitch, static_cast<const uint8_t *>(h_src) + r * src_pitch, width);
            if (rc) return rc;
        }
        return ATL_OK;
    }
    const size_t rows_per = std::max<size_t>(1, kBounce / width);
    for (size_t r0 = 0; r0 < height; r0 += rows_per) {
        const size_t nr = std::min(rows_per, height - r0);
        int s;
        int rc = bounce_slot(ctx, &s);
        if (rc) return rc;
        for (size_t r = 0; r < nr; ++r) memcpy(ctx->bounce[s] + r * width, static_cast<const uint8_t *>(h_src) + (r0 + r) * src_pitch, width);
        ATL_HIP_TRY(hipMemcpy2DAsync(static_cast<uint8_t *>(d_dst) + r0 * dst_pitch, dst_pitch, ctx->bounce[s], width, width, nr,
                                     hipMemcpyHostToDevice, st));
        ATL_HIP_TRY(hipEventRecord(ctx->bounce_ev[s], st));
        ctx->bounce_busy[s] = true;
    }
    return ATL_OK;
}

int d2h_2d(atl_ctx *ctx, hipStream_t st, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch, size_t width, size_t height) {
    if (!width || !height) return ATL_OK;
    if (host_range_is_pinned(h_dst, (height - 1) * dst_pitch + width)) {
        ATL_HIP_TRY(hipMemcpy2DAsync(h_dst, dst_pitch, d_src, src_pitch, width, height, hipMemcpyDeviceToHost, st));
        ATL_HIP_TRY(hipStreamSynchronize(st));
        return ATL_OK;
    }
    if (width > kBounce) {
        for (size_t r = 0; r < height; ++r) {
            int rc = d2h(ctx, st, static_cast<uint8_t *>(h_dst) + r * dst_pitch, static_cast<const uint8_t *>(d_src) + r * src_pitch, width);
            if (rc) return rc;
        }
        return ATL_OK;
    }
    const size_t rows_per = std::max<size_t>(1, kBounce / width);
    for (size_t r0 = 0; r0 < height; r0 += rows_per) {
        const size_t nr = std::min(rows_per, height - r0);
        int s;
        int rc = bounce_slot(ctx, &s);
        if (rc) return rc;
        ATL_HIP_TRY(hipMemcpy2DAsync(ctx->bounce[s], width, static_cast<const uint8_t *>(d_src) + r0 * src_pitch, src_pitch, width, nr,
                                     hipMemcpyDeviceToHost, st));
        ATL_HIP_TRY(hipStreamSynchronize(st));
        for (size_t r = 0; r < nr; ++r) memcpy(static_cast<uint8_t *>(h_dst) + (r0 + r) * dst_pitch, ctx->bounce[s] + r * width, width);
    }
    return ATL_OK;
}

// grid layout of a plan: (Y, X) cells in tiles of (2 << w2_log2) x (64 >> w2_log2) cells
struct Layout {
    int64_t X, Y;
    int w2_log2;  // log2 of the lanes per tile row
    // line-aligned plans (PlanDev::shift_classes): `classes` tilings of the grid; the builder sees the block-diagonal stack
    // of the matrix, class r's copy in columns [r * S, (r + 1) * S) with S = X * Y
    int classes = 0;
};
inline int64_t class_origin(const Layout &L, int64_t r) { return (r * L.X * L.Y) & 15; }
inline int64_t layout_columns(const Layout &L) { return tile_columns(L.X, L.Y, L.w2_log2, L.classes > 0); }
inline int64_t layout_tiles_per_class(const Layout &L) {
    const int h = kLanes >> L.w2_log2;
    return layout_columns(L) * ((L.Y + h - 1) / h);
}

// inverse of tile_lane_cells (atl_internal.h): the tile that owns `cell` and the cell's slot
// (2 * lane + {0, 1}) inside it
int64_t tile_of_cell(const Layout &L, int64_t cell, int32_t *local) {
    const int w = 2 << L.w2_log2, h = kLanes >> L.w2_log2;
    int64_t r = 0, o = 0;
    if (L.classes > 0) {  // stacked column -> (class, the grid's own cell)
        r = cell / (L.X * L.Y);
        cell -= r * L.X * L.Y;
        o = class_origin(L, r);
    }
    // the grid row whose flat range [lo(y), lo(y+1)) holds the cell: its own row, or - when the line
    // it lies in is shared with the start of later rows - the last row that starts in that line
    int64_t y = cell / L.X;
    while (y + 1 < L.Y && cell >= tile_row_lo(L.X, y + 1, o)) ++y;
    const int64_t p = cell - tile_row_lo(L.X, y, o);
    const int64_t tx = p / w, ty = y / h;
    const int lane = int((y % h) << L.w2_log2) + int((p % w) >> 1);
    *local = lane * 2 + int(p & 1);
    return r * layout_tiles_per_class(L) + ty * layout_columns(L) + tx;
}
}  // namespace atl

// pv_cell<TAIL, TRACK> / pvx_cell<TRACK, TRIGON> for run-time option codes
template <class F>
static int pv_probe_switch(int tracking, F &&f) {
    switch (tracking) {
        case ATL_TRACK_NONE: return f(std::integral_constant<int, ATL_TRACK_NONE>());
        case ATL_TRACK_HORIZONTAL: return f(std::integral_constant<int, ATL_TRACK_HORIZONTAL>());
        // the instantiation the kernels run for these three: the tracker is a run-time switch (kTrackAny) over the same
        // closed forms
        case ATL_TRACK_TILTED_HORIZONTAL:
        case ATL_TRACK_VERTICAL:
        case ATL_TRACK_DUAL: return f(std::integral_constant<int, kTrackAny>());
        default: set_error("atl_pv_probe_host: bad tracking code %d", tracking); return ATL_E_INVALID;
    }
}

extern "C" {

int atl_version(void) { return ATL_VERSION; }

const char *atl_last_error(void) { return g_err; }

int atl_device_count(int *count) {
    ATL_REQUIRE(count, "atl_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return ATL_OK;
}

// $ATLITE_HIP_BACKTRACE=1 (stderr) or =<file> (appended; survives a test runner that captures stderr): a native backtrace
// when the process dies of SIGABRT / SIGSEGV / SIGBUS (a runtime's abort(), glibc's heap checks, a stray pointer) -
// library offsets, resolvable with llvm-symbolizer against the same .so
static int g_backtrace_fd = 2;
static void fatal_signal_backtrace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n[atlite-hip] fatal signal, native backtrace:\n";
    (void)!write(g_backtrace_fd, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, g_backtrace_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}
static void install_backtrace_once() {
    static const bool once = [] {
        if (const char *e = getenv("ATLITE_HIP_BACKTRACE")) {
            if (*e && *e != '0') {
                if (strcmp(e, "1") != 0) {
                    const int fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
                    if (fd >= 0) g_backtrace_fd = fd;
                }
                for (int sig : {SIGABRT, SIGSEGV, SIGBUS}) signal(sig, fatal_signal_backtrace);
            }
        }
        return true;
    }();
    (void)once;
}

int atl_create(int device, void *stream, atl_ctx **out) {
    install_backtrace_once();
    ATL_REQUIRE(out, "atl_create: out is NULL");
    *out = nullptr;
    int n = 0;
    ATL_HIP_TRY(hipGetDeviceCount(&n));
    ATL_REQUIRE(device >= 0 && device < n, "atl_create: device %d out of range (have %d)",
                device, n);
    ATL_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    ATL_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("atl_create: device %d is %s; this library is built for gfx950 only", device,
                  prop.gcnArchName);
        return ATL_E_UNSUPPORTED;
    }
    atl_ctx *c = new atl_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return ATL_E_HIP;
        }
        c->own_stream = true;
    }
    (void)hipEventCreate(&c->ev_t0);
    (void)hipEventCreate(&c->ev_t1);
    (void)hipEventCreateWithFlags(&c->ev_table, hipEventDisableTiming);
    if (dev_malloc(reinterpret_cast<void **>(&c->d_table), 5 * 2 * kMaxKnots * sizeof(double)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_table), 5 * 2 * kMaxKnots * sizeof(double), hipHostMallocDefault) !=
            hipSuccess) {
        set_error("atl_create: table allocation failed");
        delete c;
        return ATL_E_NOMEM;
    }
    *out = c;
    return ATL_OK;
}

int atl_destroy(atl_ctx *ctx) {
    if (!ctx) return ATL_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)dev_free(ctx->scratch);
    if (ctx->d_table) (void)dev_free(ctx->d_table);
    if (ctx->h_table) (void)hipHostFree(ctx->h_table);
    if (ctx->ev_table) (void)hipEventDestroy(ctx->ev_table);
    (void)hipEventDestroy(ctx->ev_t0);
    (void)hipEventDestroy(ctx->ev_t1);
    for (hipEvent_t e : ctx->ev_ring) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->ingest && ctx->ingest_free) ctx->ingest_free(ctx->ingest);
    for (int b = 0; b < 2; ++b) {
        if (ctx->bounce_ev[b]) {
            if (ctx->bounce_busy[b]) (void)hipEventSynchronize(ctx->bounce_ev[b]);
            (void)hipEventDestroy(ctx->bounce_ev[b]);
        }
        if (ctx->bounce[b]) (void)hipHostFree(ctx->bounce[b]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ATL_OK;
}

int atl_sync(atl_ctx *ctx) {
    ATL_REQUIRE(ctx, "atl_sync: ctx is NULL");
    ATL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ATL_OK;
}

int atl_device_name(atl_ctx *ctx, char *buf, size_t buflen) {
    ATL_REQUIRE(ctx && buf && buflen, "atl_device_name: bad argument");
    hipDeviceProp_t prop;
    ATL_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
             prop.multiProcessorCount);
    return ATL_OK;
}

int atl_alloc(atl_ctx *ctx, size_t bytes, void **d_ptr) {
    ATL_REQUIRE(ctx && d_ptr, "atl_alloc: bad argument");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    *d_ptr = nullptr;
    ATL_HIP_TRY(dev_malloc(d_ptr, bytes ? bytes : 256));
    return ATL_OK;
}

int atl_free(atl_ctx *ctx, void *d_ptr) {
    ATL_REQUIRE(ctx, "atl_free: ctx is NULL");
    if (!d_ptr) return ATL_OK;
    ATL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ATL_HIP_TRY(dev_free(d_ptr));
    return ATL_OK;
}

int atl_upload(atl_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    ATL_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "atl_upload: bad argument");
    if (!bytes) return ATL_OK;
    int rc = h2d(ctx, ctx->stream, d_dst, h_src, bytes);
    if (rc) return rc;
    ATL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ATL_OK;
}

int atl_download(atl_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    ATL_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "atl_download: bad argument");
    if (!bytes) return ATL_OK;
    return d2h(ctx, ctx->stream, h_dst, d_src, bytes);
}


int atl_copy_2d(atl_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width, size_t height,
                int kind, int async_on_copy_stream) {
    ATL_REQUIRE(ctx && (width == 0 || height == 0 || (dst && src)), "atl_copy_2d: bad argument");
    ATL_REQUIRE(dst_pitch >= width && src_pitch >= width, "atl_copy_2d: a pitch is smaller than the row width");
    ATL_REQUIRE(kind >= 0 && kind <= 2, "atl_copy_2d: kind must be 0 (host to device), 1 (device to host) or 2 (device to device)");
    if (!width || !height) return ATL_OK;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    hipStream_t st = ctx->stream;
    if (async_on_copy_stream) {
        int rc = copy_stream_of(ctx, &st);
        if (rc) return rc;
        rc = ingest_join(ctx, st);
        if (rc) return rc;
    }
    if (kind == 0) {  // (asynchronous or not: the host rows have been read when this returns)
        int rc = h2d_2d(ctx, st, dst, dst_pitch, src, src_pitch, width, height);
        if (rc) return rc;
    } else if (kind == 1) {
        return d2h_2d(ctx, st, dst, dst_pitch, src, src_pitch, width, height);  // the data is in place on return
    } else {
        ATL_HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, height, k, st));
    }
    if (!async_on_copy_stream) ATL_HIP_TRY(hipStreamSynchronize(st));
    return ATL_OK;
}

int atl_memset(atl_ctx *ctx, void *d_dst, int byte_value, size_t bytes) {
    ATL_REQUIRE(ctx && (bytes == 0 || d_dst), "atl_memset: bad argument");
    if (!bytes) return ATL_OK;
    ATL_HIP_TRY(hipMemsetAsync(d_dst, byte_value, bytes, ctx->stream));
    return ATL_OK;
}

// ---- host-resident cutouts: copy stream + events ------------------------------------------

int atl_pinned_alloc(size_t bytes, void **h_ptr) {
    ATL_REQUIRE(h_ptr, "atl_pinned_alloc: h_ptr is NULL");
    *h_ptr = nullptr;
    hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        *h_ptr = nullptr;
        return e == hipErrorOutOfMemory ? ATL_E_NOMEM : ATL_E_HIP;
    }
    return ATL_OK;
}

int atl_pinned_free(void *h_ptr) {
    if (h_ptr) ATL_HIP_TRY(hipHostFree(h_ptr));
    return ATL_OK;
}

int atl_host_register(void *h_ptr, size_t bytes) {
    ATL_REQUIRE(h_ptr && bytes, "atl_host_register: bad argument");
    ATL_HIP_TRY(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return ATL_OK;
}

int atl_host_unregister(void *h_ptr) {
    ATL_REQUIRE(h_ptr, "atl_host_unregister: bad argument");
    ATL_HIP_TRY(hipHostUnregister(h_ptr));
    return ATL_OK;
}

int atl_upload_async(atl_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    ATL_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "atl_upload_async: bad argument");
    if (!bytes) return ATL_OK;
    hipStream_t cs;
    int rc = copy_stream_of(ctx, &cs);
    if (rc) return rc;
    rc = ingest_join(ctx, cs);
    if (rc) return rc;
    return h2d(ctx, cs, d_dst, h_src, bytes);  // page-locked sources (atl_host_register, Dataset.pin) go as they lie
}

int atl_event_create(atl_ctx *ctx, atl_event **out) {
    ATL_REQUIRE(ctx && out, "atl_event_create: bad argument");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    atl_event *e = new atl_event();
    e->device = ctx->device;
    hipError_t err = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (err != hipSuccess) {
        delete e;
        set_error("hipEventCreate failed: %s", hipGetErrorString(err));
        return ATL_E_HIP;
    }
    *out = e;
    return ATL_OK;
}

int atl_event_destroy(atl_event *ev) {
    if (!ev) return ATL_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return ATL_OK;
}

int atl_event_record(atl_ctx *ctx, atl_event *ev, int which_stream) {
    ATL_REQUIRE(ctx && ev, "atl_event_record: bad argument");
    ATL_REQUIRE(which_stream >= 0 && which_stream <= 2, "atl_event_record: which_stream must be 0, 1 or 2");
    hipStream_t st = ctx->stream;
    if (which_stream == 1) {
        int rc = ingest_finish(ctx);  // reads whose chunks the device inflated: their verdicts first
        if (rc) return rc;
        rc = copy_stream_of(ctx, &st);
        if (rc) return rc;
    } else if (which_stream == 2) {  // the copy stream as a fence only: behind the reads in flight, their verdicts left for an observer
        int rc = copy_stream_of(ctx, &st);
        if (rc) return rc;
        rc = ingest_join(ctx, st);
        if (rc) return rc;
    }
    ATL_HIP_TRY(hipEventRecord(ev->ev, st));
    return ATL_OK;
}

int atl_stream_wait_event(atl_ctx *ctx, int which_stream, atl_event *ev) {
    ATL_REQUIRE(ctx && ev, "atl_stream_wait_event: bad argument");
    hipStream_t st = ctx->stream;
    if (which_stream == 1) {
        int rc = copy_stream_of(ctx, &st);
        if (rc) return rc;
    }
    ATL_HIP_TRY(hipStreamWaitEvent(st, ev->ev, 0));
    return ATL_OK;
}

int atl_event_synchronize(atl_event *ev) {
    ATL_REQUIRE(ev, "atl_event_synchronize: ev is NULL");
    ATL_HIP_TRY(hipEventSynchronize(ev->ev));
    return ATL_OK;
}

int atl_capture_begin(atl_ctx *ctx) {
    ATL_REQUIRE(ctx, "atl_capture_begin: ctx is NULL");
    ATL_REQUIRE(!ctx->capturing, "atl_capture_begin: a capture is already open on this context");
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    ATL_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return ATL_OK;
}

int atl_capture_end(atl_ctx *ctx, atl_graph **out) {
    ATL_REQUIRE(ctx && out, "atl_capture_end: bad argument");
    *out = nullptr;
    ATL_REQUIRE(ctx->capturing, "atl_capture_end: no capture is open on this context");
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    ATL_HIP_TRY(hipStreamEndCapture(ctx->stream, &g));
    ATL_REQUIRE(g, "atl_capture_end: the capture was invalidated (a call between begin and end synchronised or allocated)");
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return ATL_E_HIP;
    }
    atl_graph *a = new atl_graph();
    a->exec = ex;
    a->device = ctx->device;
    *out = a;
    return ATL_OK;
}

int atl_graph_launch(atl_ctx *ctx, atl_graph *graph) {
    ATL_REQUIRE(ctx && graph && graph->exec, "atl_graph_launch: bad argument");
    ATL_REQUIRE(graph->device == ctx->device, "atl_graph_launch: the graph was captured on device %d", graph->device);
    ATL_HIP_TRY(hipGraphLaunch(graph->exec, ctx->stream));
    return ATL_OK;
}

int atl_graph_destroy(atl_graph *graph) {
    if (!graph) return ATL_OK;
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    delete graph;
    return ATL_OK;
}

int atl_timer_start(atl_ctx *ctx) {
    ATL_REQUIRE(ctx, "atl_timer_start: ctx is NULL");
    ATL_HIP_TRY(hipEventRecord(ctx->ev_t0, ctx->stream));
    return ATL_OK;
}

int atl_timer_stop(atl_ctx *ctx, float *ms) {
    ATL_REQUIRE(ctx && ms, "atl_timer_stop: bad argument");
    ATL_HIP_TRY(hipEventRecord(ctx->ev_t1, ctx->stream));
    ATL_HIP_TRY(hipEventSynchronize(ctx->ev_t1));
    ATL_HIP_TRY(hipEventElapsedTime(ms, ctx->ev_t0, ctx->ev_t1));
    return ATL_OK;
}

int atl_set_profiling(atl_ctx *ctx, int enabled) {
    ATL_REQUIRE(ctx, "atl_set_profiling: ctx is NULL");
    ATL_REQUIRE(enabled >= 0 && enabled <= (1 << 20), "atl_set_profiling: ring size %d out of range", enabled);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    ATL_HIP_TRY(hipStreamSynchronize(ctx->stream));  // no bracket of an earlier launch is in flight
    const size_t want = 2 * size_t(enabled);
    while (ctx->ev_ring.size() > want) {
        (void)hipEventDestroy(ctx->ev_ring.back());
        ctx->ev_ring.pop_back();
    }
    while (ctx->ev_ring.size() < want) {
        hipEvent_t e = nullptr;
        ATL_HIP_TRY(hipEventCreate(&e));
        ctx->ev_ring.push_back(e);
    }
    ctx->profiling = enabled != 0;
    ctx->ring_count = 0;
    return ATL_OK;
}

int atl_last_kernel_ms(atl_ctx *ctx, float *ms) {
    ATL_REQUIRE(ctx && ms, "atl_last_kernel_ms: bad argument");
    ATL_REQUIRE(ctx->profiling && ctx->ring_count > 0,
                "atl_last_kernel_ms: no profiled kernel (call atl_set_profiling(ctx,1) first)");
    const size_t slot = size_t((ctx->ring_count - 1) % int64_t(ctx->ev_ring.size() / 2));
    ATL_HIP_TRY(hipEventSynchronize(ctx->ev_ring[2 * slot + 1]));
    ATL_HIP_TRY(hipEventElapsedTime(ms, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1]));
    return ATL_OK;
}

int atl_kernel_times(atl_ctx *ctx, float *ms, int64_t cap, int64_t *n_out) {
    ATL_REQUIRE(ctx && n_out && (ms || cap == 0) && cap >= 0, "atl_kernel_times: bad argument");
    *n_out = 0;
    if (!ctx->profiling || ctx->ring_count == 0) return ATL_OK;
    const int64_t ring = int64_t(ctx->ev_ring.size() / 2);
    const int64_t have = std::min(ctx->ring_count, ring), n = std::min(have, cap);
    for (int64_t i = 0; i < n; ++i) {  // oldest of the n most recent first
        const size_t slot = size_t((ctx->ring_count - n + i) % ring);
        ATL_HIP_TRY(hipEventSynchronize(ctx->ev_ring[2 * slot + 1]));
        ATL_HIP_TRY(hipEventElapsedTime(ms + i, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1]));
    }
    *n_out = n;
    return ATL_OK;
}

// ---- aggregation plan ------------------------------------------------------------------

// The plan as host arrays: everything atl_agg_create uploads (PlanDev's members by the same names).  Pure host code:
// atl_agg_check_host() builds and verifies it without a device (CPU tests, the sanitizer build).
struct PlanHost {
    int64_t X = 0, Y = 0, ntx = 0, n_segs = 0, P = 0;
    int w2_log2 = 0;
    std::vector<int32_t> seg_ptr, shape_ptr, shape_prow;
    std::vector<double> prow_w, prow_wm;
    std::vector<uint8_t> poison;
    std::vector<uint64_t> seg_mask;
    std::vector<int64_t> seg_wm;
};

// classes > 0: a line-aligned plan - the CSR is the block-diagonal stack of `classes` copies of the matrix (n_rows and
// n_cells are the stacked counts), tiled class by class (Layout::classes)
static int build_plan(int64_t n_rows, int64_t n_cells, int64_t row_len, int64_t slot_stride, const int64_t *h_indptr, const int32_t *h_indices,
                      const double *h_data, PlanHost *plan, int classes = 0) {
    ATL_REQUIRE(n_rows >= 0 && n_cells >= 0, "atl_agg_create: negative shape (%lld, %lld)",
                (long long)n_rows, (long long)n_cells);
    ATL_REQUIRE(n_rows < 65536, "atl_agg_create: at most 65535 rows (shapes) are supported");
    ATL_REQUIRE(n_cells < (int64_t(1) << 31), "atl_agg_create: matrix shape too large");
    const int64_t grid_cells = classes > 0 ? n_cells / classes : n_cells;  // the grid's own cell count
    ATL_REQUIRE(row_len >= 0 && (row_len == 0 || grid_cells % row_len == 0),
                "atl_agg_create: row_len %lld does not divide the %lld cells", (long long)row_len,
                (long long)grid_cells);
    ATL_REQUIRE(h_indptr, "atl_agg_create: indptr is NULL");
    ATL_REQUIRE(h_indptr[0] == 0, "atl_agg_create: indptr[0] must be 0");
    const int64_t nnz = h_indptr[n_rows];
    ATL_REQUIRE(nnz >= 0 && (nnz == 0 || (h_indices && h_data)),
                "atl_agg_create: indices/data missing");
    // the row pointers are checked as a whole before any of them is used as an offset: indices / data hold
    // indptr[n_rows] entries, and a non-monotone interior pointer must not be followed past them
    for (int64_t r = 0; r < n_rows; ++r)
        ATL_REQUIRE(h_indptr[r + 1] >= h_indptr[r] && h_indptr[r + 1] <= nnz, "atl_agg_create: indptr not monotone at row %lld",
                    (long long)r);

    // ---- validate, collect (row, cell, weight) -------------------------------------------
    struct Raw {
        int32_t row, cell;
        double w;
    };
    std::vector<Raw> raw;
    raw.reserve(static_cast<size_t>(nnz));
    std::vector<uint8_t> poison(static_cast<size_t>(n_rows), 0);
    for (int64_t r = 0; r < n_rows; ++r) {
        ATL_REQUIRE(h_indptr[r + 1] >= h_indptr[r], "atl_agg_create: indptr not monotone at row %lld",
                    (long long)r);
        for (int64_t k = h_indptr[r]; k < h_indptr[r + 1]; ++k) {
            const int64_t j = h_indices[k];
            ATL_REQUIRE(j >= 0 && j < n_cells,
                        "atl_agg_create: column index %lld out of range [0,%lld)", (long long)j,
                        (long long)n_cells);
            const double w = h_data[k];
            if (std::isnan(w)) {
                poison[r] = 1;
                continue;
            }
            raw.push_back({int32_t(r), int32_t(j), w});
        }
    }

    // ---- choose the tile shape --------------------------------------------------------------
    auto ntx_of = [](const Layout &L) { return layout_columns(L); };
    auto tile_of = [](const Layout &L, int64_t cell, int32_t *local) { return tile_of_cell(L, cell, local); };
    std::vector<Layout> cands;
    cands.push_back({grid_cells > 0 ? grid_cells : 1, 1, 6, classes});  // flat 128 x 1 over the stacked axis
    if (row_len > 0 && grid_cells / row_len > 1) {
        const int64_t Yg = grid_cells / row_len;
        for (int l2 : {5, 4, 3}) cands.push_back({row_len, Yg, l2, classes});  // 64x2, 32x4, 16x8
    }
    if (const char *env = getenv("ATLITE_HIP_TILE")) {  // experiments: "16x8", "32x4", "64x2", "128x1", "flat"
        int w = 0, h = 0;
        if (row_len > 0 && sscanf(env, "%dx%d", &w, &h) == 2 && w * h == kSegCells && w >= 16) {
            int l2 = 0;
            while ((2 << l2) < w) ++l2;
            cands.assign(1, Layout{row_len, grid_cells / row_len, l2, classes});
        } else if (strcmp(env, "flat") == 0) {
            cands.resize(1);
        }
    }
    // Entries arrive grouped by row (CSR).  For a layout, the distinct tiles a row touches are found
    // with a per-row marker array in O(nnz): no sorting of the entries.
    const size_t n_raw = raw.size();
    std::vector<int64_t> row_begin(static_cast<size_t>(n_rows) + 1, 0);
    for (const Raw &e : raw) row_begin[size_t(e.row) + 1]++;
    for (int64_t r = 0; r < n_rows; ++r) row_begin[r + 1] += row_begin[r];

    auto n_tiles_of = [&](const Layout &L) { return n_cells > 0 ? layout_tiles_per_class(L) * std::max(1, L.classes) : int64_t(0); };
    size_t best = 0;
    if (cands.size() > 1) {
        double best_cost = 0;
        std::vector<int32_t> mark;
        std::vector<uint8_t> used;
        for (size_t c = 0; c < cands.size(); ++c) {
            const int64_t nt = n_tiles_of(cands[c]);
            mark.assign(size_t(nt), -1);
            used.assign(size_t(nt), 0);
            int64_t P = 0, tiles = 0;
            int32_t loc;
            for (int64_t r = 0; r < n_rows; ++r) {
                for (int64_t k = row_begin[r]; k < row_begin[r + 1]; ++k) {
                    const int64_t t = tile_of(cands[c], raw[size_t(k)].cell, &loc);
                    if (mark[size_t(t)] != int32_t(r)) {
                        mark[size_t(t)] = int32_t(r);
                        ++P;
                        if (!used[size_t(t)]) {
                            used[size_t(t)] = 1;
                            ++tiles;
                        }
                    }
                }
            }
            // A tile costs its memory stream (and conversion work) even for masked lanes, a partial row one wave
            // reduction: about 40 : 1 (C2, 325 tiles: 16x8 3.09 ms with 1131 partial rows, 32x4 3.11 / 1462, 64x2
            // 3.20 / 2245, flat 3.26 / 3590 - profiles/r03_tiles_unaligned.txt).  With 128-byte-aligned (sheared) tile
            // rows and nontemporal loads the row width itself hardly matters, so the partial-row count decides; 64x2
            // keeps a small measured penalty.
            // UNLESS the slots' rows are not 128-byte aligned (S % 16 != 0: every real-world grid with odd dimensions):
            // the line grid then moves with the slot - in a fraction f of the slots (all but those whose phase
            // (t S) % 16 is 0) a tile row of w cells straddles one line more than its w / 16, shared with its
            // neighbour: +16 / w of the traffic.  Measured (201 x 200, f = 1/2): 16x8 4.65 ms, 32x4 4.02, 64x2 3.72,
            // flat 3.63; (189 x 157, odd S): 16x8 4.43, flat 3.01.  Wide tiles win there.
            static const double row_eff[7] = {0, 0, 0, 1.0, 1.0, 1.05, 1.0};
            const int64_t stride = slot_stride > 0 ? slot_stride : n_cells;  // cells between slots (ld_cells)
            int64_t g = stride % 16;  // phases are the multiples of gcd(stride % 16, 16)
            for (int64_t b = 16; b != 0;) {
                const int64_t r = g % b;
                g = b;
                b = r;
            }
            const double f_mis = (stride % 16 == 0 || classes > 0) ? 0.0 : 1.0 - double(g) / 16.0;  // (line-aligned plans: no tile row off the line grid)
            const int w = 2 << cands[c].w2_log2;
            const double cost = (40.0 * double(tiles) * (1.0 + f_mis * 16.0 / w) + double(P)) * row_eff[cands[c].w2_log2];
            if (c == 0 || cost < best_cost) {
                best_cost = cost;
                best = c;
            }
        }
    }
    const Layout L = cands[best];
    const int64_t ntx = ntx_of(L);
    const int64_t n_segs = n_tiles_of(L);

    // ---- partial rows: one per (tile, row) pair, ordered by tile then row ------------------------
    // pass 1: per row, the tiles it touches (first-touch order), counted per tile
    std::vector<int32_t> seg_ptr(static_cast<size_t>(n_segs) + 1, 0);
    std::vector<int64_t> pair_tile;  // (row-major) tile of every (row, tile) pair
    std::vector<int64_t> pair_ptr(static_cast<size_t>(n_rows) + 1, 0);
    {
        std::vector<int32_t> mark(static_cast<size_t>(n_segs), -1);
        int32_t loc;
        for (int64_t r = 0; r < n_rows; ++r) {
            for (int64_t k = row_begin[r]; k < row_begin[r + 1]; ++k) {
                const int64_t t = tile_of(L, raw[size_t(k)].cell, &loc);
                if (mark[size_t(t)] != int32_t(r)) {
                    mark[size_t(t)] = int32_t(r);
                    pair_tile.push_back(t);
                    seg_ptr[size_t(t) + 1]++;
                }
            }
            pair_ptr[r + 1] = int64_t(pair_tile.size());
        }
        for (int64_t s2 = 0; s2 < n_segs; ++s2) seg_ptr[s2 + 1] += seg_ptr[s2];
    }
    const int64_t P = int64_t(pair_tile.size());
    ATL_REQUIRE(P < (int64_t(1) << 31), "atl_agg_create: too many partial rows");
    // pass 2: rows are visited in ascending order, so filling each tile's slots in visiting order
    // sorts the partial rows of a tile by row
    std::vector<int32_t> pair_prow(static_cast<size_t>(P));
    {
        std::vector<int32_t> fill(seg_ptr.begin(), seg_ptr.end() - 1);
        for (int64_t q = 0; q < P; ++q) pair_prow[size_t(q)] = fill[size_t(pair_tile[size_t(q)])]++;
    }
    // pass 3: weights
    std::vector<double> prow_w(static_cast<size_t>(P) * kSegCells,
                               std::numeric_limits<double>::quiet_NaN());
    std::vector<int32_t> shape_ptr(static_cast<size_t>(n_rows) + 1, 0);
    std::vector<int32_t> shape_prow(static_cast<size_t>(P), 0);
    {
        std::vector<int32_t> tile_prow(static_cast<size_t>(n_segs), -1);
        int32_t loc;
        for (int64_t r = 0; r < n_rows; ++r) {
            for (int64_t q = pair_ptr[r]; q < pair_ptr[r + 1]; ++q) tile_prow[size_t(pair_tile[size_t(q)])] = pair_prow[size_t(q)];
            for (int64_t k = row_begin[r]; k < row_begin[r + 1]; ++k) {
                const Raw &e = raw[size_t(k)];
                const int64_t t = tile_of(L, e.cell, &loc);
                double &slot = prow_w[size_t(tile_prow[size_t(t)]) * kSegCells + size_t(loc)];
                slot = std::isnan(slot) ? e.w : slot + e.w;  // duplicates are summed (scipy CSR)
            }
            // the shape's partial rows in ascending tile (= ascending partial row) order
            shape_ptr[r + 1] = shape_ptr[r] + int32_t(pair_ptr[r + 1] - pair_ptr[r]);
            std::vector<int32_t> mine(pair_prow.begin() + pair_ptr[r], pair_prow.begin() + pair_ptr[r + 1]);
            std::sort(mine.begin(), mine.end());
            std::copy(mine.begin(), mine.end(), shape_prow.begin() + shape_ptr[r]);
        }
    }
    (void)n_raw;
    // lanes of each tile that own a weighted cell (PlanDev::seg_mask)
    std::vector<uint64_t> seg_mask(static_cast<size_t>(n_segs), 0);
    for (int64_t t = 0; t < n_segs; ++t)
        for (int32_t q = seg_ptr[size_t(t)]; q < seg_ptr[size_t(t) + 1]; ++q)
            for (int c = 0; c < kSegCells; ++c)
                if (!std::isnan(prow_w[size_t(q) * kSegCells + size_t(c)])) seg_mask[size_t(t)] |= uint64_t(1) << (c >> 1);
    // pass 4: the MFMA operand image of dense tiles (PlanDev::prow_wm)
    std::vector<int64_t> seg_wm;
    std::vector<double> prow_wm;
    if (!getenv("ATLITE_HIP_NO_MFMA")) {
        // Only for plans DOMINATED by dense tiles (3/4 of the partial rows in MFMA groups): the kernel instantiation that carries the MFMA path costs every tile
        // ~20 VGPRs and 8 KiB of LDS per wave (measured: a stack of 8 tessellations, 12.8 rows per tile on average with
        // a third of them in dense tiles, ran 11 % slower with it; 16 / 32 dense rows per tile run 12 / 19 % faster).
        int64_t total = 0, dense_rows = 0;
        for (int64_t t = 0; t < n_segs; ++t) {
            const int n = seg_ptr[size_t(t) + 1] - seg_ptr[size_t(t)], G = mfma_groups(n);
            total += int64_t(G) * 32 * 64;
            dense_rows += std::min(n, G * kMfmaRows);
        }
        if (total > 0 && (4 * dense_rows >= 3 * P || getenv("ATLITE_HIP_FORCE_MFMA"))) {
            seg_wm.assign(static_cast<size_t>(n_segs), -1);
            prow_wm.assign(static_cast<size_t>(total), 0.0);
            int64_t off = 0;
            for (int64_t t = 0; t < n_segs; ++t) {
                const int32_t q0 = seg_ptr[size_t(t)], n = seg_ptr[size_t(t) + 1] - q0;
                const int G = mfma_groups(n);
                if (G == 0) continue;
                seg_wm[size_t(t)] = off;
                for (int g = 0; g < G; ++g)
                    for (int k4 = 0; k4 < 32; ++k4)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int row = kMfmaRows * g + lane % 16, cell = 4 * k4 + lane / 16;
                            if (row >= n) continue;
                            const double w = prow_w[size_t(q0 + row) * kSegCells + size_t(cell)];
                            if (!std::isnan(w)) prow_wm[size_t(off) + (size_t(g) * 32 + size_t(k4)) * 64 + size_t(lane)] = w;
                        }
                off += int64_t(G) * 32 * 64;
            }
        }
    }
    plan->X = L.X;
    plan->Y = L.Y;
    plan->ntx = ntx;
    plan->n_segs = n_segs;
    plan->P = P;
    plan->w2_log2 = L.w2_log2;
    plan->seg_ptr = std::move(seg_ptr);
    plan->shape_ptr = std::move(shape_ptr);
    plan->shape_prow = std::move(shape_prow);
    plan->prow_w = std::move(prow_w);
    plan->prow_wm = std::move(prow_wm);
    plan->poison = std::move(poison);
    plan->seg_mask = std::move(seg_mask);
    plan->seg_wm = std::move(seg_wm);
    return ATL_OK;
}

// the device copy of a host plan
static int plan_to_device(atl_ctx *ctx, const PlanHost &ph, int64_t n_rows, int64_t n_cells, atl_agg **out) {
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    atl_agg *a = new atl_agg();
    a->ctx = ctx;
    a->dev.n_rows = n_rows;
    a->dev.n_cells = n_cells;
    a->dev.X = ph.X;
    a->dev.Y = ph.Y;
    a->dev.ntx = int32_t(ph.ntx);
    a->dev.w2_log2 = ph.w2_log2;
    a->dev.n_segs = int32_t(ph.n_segs);
    a->dev.n_prows = int32_t(ph.P);
    int rc = ATL_OK;
    if ((rc = to_device(a, ph.seg_ptr, &a->dev.seg_ptr)) ||
        (rc = to_device(a, ph.prow_w, &a->dev.prow_w)) ||
        (rc = to_device(a, ph.shape_ptr, &a->dev.shape_ptr)) ||
        (rc = to_device(a, ph.shape_prow, &a->dev.shape_prow)) ||
        (rc = to_device(a, ph.poison, &a->dev.row_poison)) ||
        (rc = to_device(a, ph.seg_mask, &a->dev.seg_mask)) ||
        (!ph.prow_wm.empty() && ((rc = to_device(a, ph.seg_wm, &a->dev.seg_wm)) || (rc = to_device(a, ph.prow_wm, &a->dev.prow_wm))))) {
        atl_agg_destroy(a);
        return rc;
    }
    *out = a;
    return ATL_OK;
}

// alignment classes of contiguous cubes with n_cells cells per slot: 16 / gcd(n_cells, 16)
static int64_t alignment_classes(int64_t n_cells) {
    int64_t g = n_cells % 16, b = 16;
    while (g) {
        const int64_t t = b % g;
        b = g;
        g = t;
    }
    return 16 / b;
}

// block-diagonal stack of p copies of a CSR matrix (copy r in columns [r S, (r + 1) S)): what build_plan tiles class by class
static int stack_classes(int64_t p, int64_t n_rows, int64_t n_cells, const int64_t *h_indptr, const int32_t *h_indices, const double *h_data,
                         std::vector<int64_t> *indptr, std::vector<int32_t> *indices, std::vector<double> *data) {
    ATL_REQUIRE(h_indptr && h_indptr[0] == 0, "atl_agg_create_aligned: indptr is NULL or does not start at 0");
    for (int64_t r = 0; r < n_rows; ++r)
        ATL_REQUIRE(h_indptr[r + 1] >= h_indptr[r], "atl_agg_create_aligned: indptr not monotone at row %lld", (long long)r);
    const int64_t nnz = h_indptr[n_rows];
    ATL_REQUIRE(nnz == 0 || (h_indices && h_data), "atl_agg_create_aligned: indices/data missing");
    ATL_REQUIRE(p * n_rows < 65536, "atl_agg_create_aligned: %lld rows x %lld alignment classes exceed 65535 plan rows", (long long)n_rows,
                (long long)p);
    ATL_REQUIRE(p * n_cells < (int64_t(1) << 31) && p * nnz < (int64_t(1) << 40), "atl_agg_create_aligned: matrix too large");
    indptr->assign(size_t(p * n_rows) + 1, 0);
    indices->resize(size_t(p * nnz));
    data->resize(size_t(p * nnz));
    for (int64_t r = 0; r < p; ++r)
        for (int64_t n = 0; n < n_rows; ++n) {
            (*indptr)[size_t(r * n_rows + n) + 1] = r * nnz + h_indptr[n + 1];
            for (int64_t k = h_indptr[n]; k < h_indptr[n + 1]; ++k) {
                ATL_REQUIRE(h_indices[k] >= 0 && h_indices[k] < n_cells, "atl_agg_create_aligned: column index %lld out of range [0,%lld)",
                            (long long)h_indices[k], (long long)n_cells);
                (*indices)[size_t(r * nnz + k)] = int32_t(r * n_cells + h_indices[k]);
                (*data)[size_t(r * nnz + k)] = h_data[k];
            }
        }
    return ATL_OK;
}

int atl_agg_create(atl_ctx *ctx, int64_t n_rows, int64_t n_cells, int64_t row_len,
                   const int64_t *h_indptr, const int32_t *h_indices, const double *h_data,
                   atl_agg **out) {
    ATL_REQUIRE(ctx && out, "atl_agg_create: bad argument");
    *out = nullptr;
    PlanHost ph;
    {
        const int brc = build_plan(n_rows, n_cells, row_len, ctx->slot_stride, h_indptr, h_indices, h_data, &ph);
        if (brc) return brc;
    }
    return plan_to_device(ctx, ph, n_rows, n_cells, out);
}

int atl_agg_create_aligned(atl_ctx *ctx, int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr, const int32_t *h_indices,
                           const double *h_data, atl_agg **out) {
    ATL_REQUIRE(ctx && out, "atl_agg_create_aligned: bad argument");
    *out = nullptr;
    ATL_REQUIRE(n_rows >= 0 && n_cells >= 16, "atl_agg_create_aligned: needs at least 16 cells (got %lld)", (long long)n_cells);
    ATL_REQUIRE(n_cells % 16 != 0, "atl_agg_create_aligned: the slots of %lld cells start on 128-byte lines already (use atl_agg_create)",
                (long long)n_cells);
    const int64_t p = alignment_classes(n_cells);
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> data;
    int rc = stack_classes(p, n_rows, n_cells, h_indptr, h_indices, h_data, &indptr, &indices, &data);
    if (rc) return rc;
    PlanHost ph;
    if ((rc = build_plan(p * n_rows, p * n_cells, row_len, 0, indptr.data(), indices.data(), data.data(), &ph, int(p)))) return rc;
    if ((rc = plan_to_device(ctx, ph, p * n_rows, n_cells, out))) return rc;
    (*out)->dev.shift_classes = int32_t(p);
    (*out)->dev.shift_rows = int32_t(n_rows);
    (*out)->dev.shift_tiles = int32_t(ph.n_segs / p);
    return ATL_OK;
}

// classes > 0: the CSR is the stacked matrix of a line-aligned plan (n_rows, n_cells the stacked counts)
static int verify_plan(int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr, const int32_t *h_indices,
                       const double *h_data, int classes, int64_t *n_partial_rows, int64_t *n_dense_tiles, int64_t *n_errors) {
    PlanHost ph;
    const int rc = build_plan(n_rows, n_cells, row_len, 0, h_indptr, h_indices, h_data, &ph, classes);
    if (rc) return rc;
    int64_t err = 0, dense = 0;
    const Layout L{ph.X, ph.Y, ph.w2_log2, classes};
    const int64_t tpc = std::max<int64_t>(1, layout_tiles_per_class(L)), grid_cells = ph.X * ph.Y;
    if (ph.n_segs != tpc * std::max(1, classes) && n_cells > 0) ++err;
    // (1) every CSR entry (duplicates summed, NaN weights poison their row) sits at ONE place of the partial rows of
    //     its shape, and nothing else does: rebuild the dense (row x cell) matrix from the plan and from the CSR
    std::vector<double> want(size_t(n_rows) * size_t(n_cells), std::numeric_limits<double>::quiet_NaN()), got = want;
    std::vector<uint8_t> poison(size_t(n_rows), 0);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = h_indptr[r]; k < h_indptr[r + 1]; ++k) {
            if (std::isnan(h_data[k])) {
                poison[size_t(r)] = 1;
                continue;
            }
            double &w = want[size_t(r) * size_t(n_cells) + size_t(h_indices[k])];
            w = std::isnan(w) ? h_data[k] : w + h_data[k];
        }
    // which tile a partial row belongs to
    std::vector<int32_t> prow_tile(size_t(ph.P), -1);
    for (int64_t t = 0; t < ph.n_segs; ++t) {
        if (ph.seg_ptr[size_t(t) + 1] < ph.seg_ptr[size_t(t)]) ++err;
        for (int32_t q = ph.seg_ptr[size_t(t)]; q < ph.seg_ptr[size_t(t) + 1]; ++q) prow_tile[size_t(q)] = int32_t(t);
    }
    for (int64_t r = 0; r < n_rows; ++r) {
        if (poison[size_t(r)] != ph.poison[size_t(r)]) ++err;
        int32_t last = -1;
        for (int32_t i = ph.shape_ptr[size_t(r)]; i < ph.shape_ptr[size_t(r) + 1]; ++i) {
            const int32_t q = ph.shape_prow[size_t(i)];
            if (q <= last || q < 0 || q >= ph.P) {  // ascending partial rows = ascending tiles: k_combine's fixed order
                ++err;
                continue;
            }
            last = q;
            const int32_t t = prow_tile[size_t(q)];
            // the tile's class: its tiling's origin and where its cells sit in the stacked matrix
            const int64_t cls = classes > 0 ? t / tpc : 0, col0 = cls * grid_cells;
            if (classes > 0 && cls != r / (n_rows / classes)) ++err;  // a class's rows live in that class's tiles
            for (int lane = 0; lane < kLanes; ++lane) {
                const TileLane tl = tile_lane_cells(ph.X, ph.Y, int32_t(ph.ntx), ph.w2_log2, int32_t(t - cls * tpc), lane,
                                                    classes > 0 ? class_origin(L, cls) : 0);
                for (int j = 0; j < 2; ++j) {
                    const double w = ph.prow_w[size_t(q) * kSegCells + size_t(2 * lane + j)];
                    if (std::isnan(w)) continue;
                    if (!(j == 0 ? tl.v0 : tl.v1)) {  // a weight on a lane that owns no cell
                        ++err;
                        continue;
                    }
                    double &g = got[size_t(r) * size_t(n_cells) + size_t(col0 + tl.c0 + j)];
                    if (!std::isnan(g)) ++err;  // the same (row, cell) twice
                    g = w;
                }
            }
        }
    }
    for (size_t i = 0; i < want.size(); ++i) {
        const bool a = std::isnan(want[i]), b = std::isnan(got[i]);
        if (a != b || (!a && want[i] != got[i])) ++err;
    }
    // (2) the coverage mask = lanes with a weight in some partial row of the tile; (3) the MFMA operand image
    for (int64_t t = 0; t < ph.n_segs; ++t) {
        uint64_t m = 0;
        const int32_t q0 = ph.seg_ptr[size_t(t)], n = ph.seg_ptr[size_t(t) + 1] - q0;
        for (int32_t q = q0; q < q0 + n; ++q)
            for (int c = 0; c < kSegCells; ++c)
                if (!std::isnan(ph.prow_w[size_t(q) * kSegCells + size_t(c)])) m |= uint64_t(1) << (c >> 1);
        if (m != ph.seg_mask[size_t(t)]) ++err;
        if (ph.prow_wm.empty()) continue;
        const int G = mfma_groups(n);
        if ((G == 0) != (ph.seg_wm[size_t(t)] < 0)) {
            ++err;
            continue;
        }
        if (G == 0) continue;
        ++dense;
        for (int g = 0; g < G; ++g)
            for (int k4 = 0; k4 < 32; ++k4)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = kMfmaRows * g + lane % 16, cell = 4 * k4 + lane / 16;
                    const double img = ph.prow_wm[size_t(ph.seg_wm[size_t(t)]) + (size_t(g) * 32 + size_t(k4)) * 64 + size_t(lane)];
                    const double w = row < n ? ph.prow_w[size_t(q0 + row) * kSegCells + size_t(cell)] : std::numeric_limits<double>::quiet_NaN();
                    if (img != (std::isnan(w) ? 0.0 : w)) ++err;
                }
    }
    if (n_partial_rows) *n_partial_rows = ph.P;
    if (n_dense_tiles) *n_dense_tiles = dense;
    *n_errors = err;
    return ATL_OK;
}

int atl_agg_check_host(int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr, const int32_t *h_indices,
                       const double *h_data, int64_t *n_partial_rows, int64_t *n_dense_tiles, int64_t *n_errors) {
    ATL_REQUIRE(n_errors, "atl_agg_check_host: n_errors is NULL");
    *n_errors = -1;
    return verify_plan(n_rows, n_cells, row_len, h_indptr, h_indices, h_data, 0, n_partial_rows, n_dense_tiles, n_errors);
}

int atl_agg_check_host_aligned(int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr, const int32_t *h_indices,
                               const double *h_data, int64_t *n_partial_rows, int64_t *n_dense_tiles, int64_t *n_errors) {
    ATL_REQUIRE(n_errors, "atl_agg_check_host_aligned: n_errors is NULL");
    *n_errors = -1;
    ATL_REQUIRE(n_rows >= 0 && n_cells >= 16 && n_cells % 16 != 0, "atl_agg_check_host_aligned: needs a cell count >= 16 that is not a multiple of 16");
    const int64_t p = alignment_classes(n_cells);
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> data;
    const int rc = stack_classes(p, n_rows, n_cells, h_indptr, h_indices, h_data, &indptr, &indices, &data);
    if (rc) return rc;
    return verify_plan(p * n_rows, p * n_cells, row_len, indptr.data(), indices.data(), data.data(), int(p), n_partial_rows, n_dense_tiles, n_errors);
}

int atl_math_probe_host(int fn, const double *h_in, int64_t n, double *h_out) {
    ATL_REQUIRE(fn >= 0 && fn <= 8 && n >= 0 && (n == 0 || (h_in && h_out)), "atl_math_probe_host: bad argument");
    double ltab[2 * kLogTabN];
    for (int i = 0; i < kLogTabN; ++i) log_table_entry(ltab, i);
    for (int64_t i = 0; i < n; ++i) {
        const double x = h_in[i];
        double r;
        switch (fn) {
            case 0: r = lean_sin(x); break;
            case 1: r = lean_cos(x); break;
            case 2: r = lean_log(x); break;
            case 3: {
                double sn, cs;
                lean_sincos(x, &sn, &cs);
                r = sn;
                h_out[n + i] = cs;
                break;
            }
            case 5: r = log_core_tab(x, ltab); break;
            case 7: r = lean_sqrt(x); break;
            case 8: r = lean_sqrt_rsqrt(x, &h_out[n + i]); break;
            case 6: r = guarded_div(x, h_in[n + i]); break;
            default: r = fast_div(x, h_in[n + i]); break;
        }
        h_out[i] = r;
    }
    return ATL_OK;
}

int atl_wind_interp_host(const double *h_V, const double *h_F, int n_knots, const double *h_x, int64_t m,
                         double *h_out) {
    ATL_REQUIRE(h_V && h_F && n_knots >= 1 && n_knots <= kMaxKnots && m >= 0 && (m == 0 || (h_x && h_out)),
                "atl_wind_interp_host: bad argument");
    std::vector<double> tbl;
    bool finite = true;
    int n_pad = 0;
    n_knots = wind_table_build(h_V, h_F, n_knots, tbl, &n_pad, &finite);
    ATL_REQUIRE(n_knots > 0, "wind speed 'V' in the turbine config is expected to be increasing");
    // grid-aligned knots: the bucket lookup make_wind() selects (ATLITE_HIP_WIND_NO_GRID forces the search)
    std::vector<double> grid;
    double inv_w = 0.0;
    int b0 = 0;
    const int n_grid = (finite && !getenv("ATLITE_HIP_WIND_NO_GRID")) ? wind_grid_build(tbl.data(), n_knots, n_pad, grid, &inv_w, &b0) : 0;
    for (int64_t i = 0; i < m; ++i) {
        if (n_grid > 0) {
            h_out[i] = interp_grid(grid.data(), tbl[0], tbl[size_t(n_knots - 1)], inv_w, b0, h_x[i]);
            continue;
        }
        if (!finite) {  // non-finite knots / values: the literal transcription (wind_dispatch's generic converter)
            h_out[i] = interp_literal(tbl.data(), n_knots, n_pad, h_x[i]);
            continue;
        }
        switch (n_pad) {  // the instantiations wind_dispatch() selects
            case 16: h_out[i] = interp_padded<4>(tbl.data(), n_knots, n_pad, h_x[i]); break;
            case 32: h_out[i] = interp_padded<5>(tbl.data(), n_knots, n_pad, h_x[i]); break;
            case 128: h_out[i] = interp_padded<7>(tbl.data(), n_knots, n_pad, h_x[i]); break;
            default: h_out[i] = interp_padded<0>(tbl.data(), n_knots, n_pad, h_x[i]); break;
        }
    }
    return ATL_OK;
}

int atl_pv_probe_host(const atl_pv_params *p, int family, int64_t n, const double *const *h_in, double *h_out) {
    ATL_REQUIRE(p && n >= 0 && h_in && (n == 0 || h_out), "atl_pv_probe_host: bad argument");
    for (int k : {3, 8, 9, 10, 11}) ATL_REQUIRE(h_in[k], "atl_pv_probe_host: toa, altitude, azimuth and the panel angles are needed");
    const double *dir = h_in[0], *dif = h_in[1], *infl = h_in[2], *toa = h_in[3], *alb = h_in[4], *outf = h_in[5],
                 *tmp = h_in[6], *hum = h_in[7], *alt = h_in[8], *az = h_in[9], *slope = h_in[10], *pazim = h_in[11];
    const PvConst k = pv_const_of(p);
    auto at = [](const double *a, int64_t i) { return a ? a[i] : 0.0; };
    if (family == 1) {  // the general kernel's per-cell routine
        ATL_REQUIRE((infl != nullptr) != (dir != nullptr && dif != nullptr) || infl, "atl_pv_probe_host: need influx or direct + diffuse");
        const PvxOpt o = pvx_opt_of(p, infl != nullptr, alb != nullptr);
        const bool other = p->trigon_model == ATL_TRIGON_OTHER;
        return pv_probe_switch(p->tracking, [&](auto tr) {
            constexpr int TR = decltype(tr)::value;
            for (int64_t i = 0; i < n; ++i) {
                h_out[i] = other ? pvx_cell<TR, ATL_TRIGON_OTHER>(at(dir, i), at(dif, i), at(infl, i), toa[i], at(alb, i), at(outf, i),
                                                                  at(tmp, i), at(hum, i), alt[i], az[i], slope[i], pazim[i], k, o)
                                 : pvx_cell<TR, ATL_TRIGON_SIMPLE>(at(dir, i), at(dif, i), at(infl, i), toa[i], at(alb, i), at(outf, i),
                                                                   at(tmp, i), at(hum, i), alt[i], az[i], slope[i], pazim[i], k, o);
            }
            return int(ATL_OK);
        });
    }
    // the fast family's influx head (pv_influx_fast, atl_kernels_pvi.hip): Reindl split (either clearsky model) +
    // albedo from outflux or the albedo variable, then the Huld panel after either trigon model
    if (family == 0 && infl && (outf || alb) && !dir && !dif) {
        PvConst kk = k;
        const double *second = outf;
        if (alb) {  // the dataset's albedo variable wins (irradiation.py:129-131)
            kk.alb_cube = 1;
            second = alb;
        }
        ATL_REQUIRE(tmp && p->tracking == ATL_TRACK_NONE && (p->trigon_model == ATL_TRIGON_SIMPLE || p->trigon_model == ATL_TRIGON_OTHER) &&
                        (p->clearsky_model == ATL_CLEARSKY_SIMPLE || (p->clearsky_model == ATL_CLEARSKY_ENHANCED && hum)) &&
                        p->panel_model == ATL_PANEL_HULD && p->irradiation == ATL_IRR_TOTAL,
                    "atl_pv_probe_host: the influx / outflux head serves the Huld panel on a fixed mount (either trigon / clearsky model)");
        const bool hd = p->trigon_model == ATL_TRIGON_OTHER, enh = p->clearsky_model == ATL_CLEARSKY_ENHANCED;
        for (int64_t i = 0; i < n; ++i) {
            const double rh = at(hum, i);
            if (hd) {
                const PvOri o = PvConvT<false, true, false, kTailHuldHayDavies>::make_ori(slope[i], pazim[i]);
                h_out[i] = enh ? pv_cell_influx_auto<kTailHuldHayDavies, true>(infl[i], second[i], toa[i], tmp[i], rh, alt[i], az[i], o, kk)
                               : pv_cell_influx_auto<kTailHuldHayDavies, false>(infl[i], second[i], toa[i], tmp[i], rh, alt[i], az[i], o, kk);
            } else {
                const PvOri o = PvConvT<false, true, false, kTailHuld>::make_ori(slope[i], pazim[i]);
                h_out[i] = enh ? pv_cell_influx_auto<kTailHuld, true>(infl[i], second[i], toa[i], tmp[i], rh, alt[i], az[i], o, kk)
                               : pv_cell_influx_auto<kTailHuld, false>(infl[i], second[i], toa[i], tmp[i], rh, alt[i], az[i], o, kk);
            }
        }
        return ATL_OK;
    }
    // the fast family: stored angles, direct / diffuse / albedo / temperature; tail and tracker from the options
    ATL_REQUIRE(family == 0 && dir && dif && alb && tmp, "atl_pv_probe_host: the fast family needs direct, diffuse, albedo, temperature");
    const bool hd = p->trigon_model == ATL_TRIGON_OTHER;
    const int tail = p->panel_model == ATL_PANEL_SOLAR_THERMAL ? (hd ? kTailThermalHayDavies : kTailThermal)
                     : p->panel_model == ATL_PANEL_NONE        ? (hd ? kTailIrradiationHayDavies : kTailIrradiation)
                     : p->panel_model == ATL_PANEL_BOFINGER    ? (hd ? kTailBofingerHayDavies : kTailBofinger)
                                                               : (hd ? kTailHuldHayDavies : kTailHuld);
    ATL_REQUIRE(p->tracking == ATL_TRACK_NONE || p->panel_model != ATL_PANEL_SOLAR_THERMAL,
                "atl_pv_probe_host: the fast family has no tracker for the solar thermal collector");
    return pv_probe_switch(p->tracking, [&](auto tr) {
        constexpr int TR = decltype(tr)::value;
        auto run = [&](auto tl) {
            constexpr int TL = decltype(tl)::value;
            for (int64_t i = 0; i < n; ++i) {
                const PvOri o = PvConvT<false, true, false, TL>::make_ori(slope[i], pazim[i]);
                h_out[i] = pv_cell_auto<TL, TR>(dir[i], dif[i], toa[i], alb[i], tmp[i], alt[i], az[i], o, k);
            }
            return int(ATL_OK);
        };
        switch (tail) {
            case kTailHuldHayDavies: return run(std::integral_constant<int, kTailHuldHayDavies>());
            case kTailThermal: return run(std::integral_constant<int, kTailThermal>());
            case kTailIrradiation: return run(std::integral_constant<int, kTailIrradiation>());
            case kTailBofinger: return run(std::integral_constant<int, kTailBofinger>());
            case kTailThermalHayDavies: return run(std::integral_constant<int, kTailThermalHayDavies>());
            case kTailIrradiationHayDavies: return run(std::integral_constant<int, kTailIrradiationHayDavies>());
            case kTailBofingerHayDavies: return run(std::integral_constant<int, kTailBofingerHayDavies>());
            default: return run(std::integral_constant<int, kTailHuld>());
        }
    });
}

int atl_wind_probe_host(const atl_wind_params *p, int64_t n, const double *h_wnd, const double *h_aux, double *h_out) {
    ATL_REQUIRE(p && n >= 0 && (n == 0 || (h_wnd && h_out)), "atl_wind_probe_host: bad argument");
    ATL_REQUIRE(p->method == ATL_WIND_NONE || p->method == ATL_WIND_LOG || p->method == ATL_WIND_POWER,
                "Interpolation method must be 'logarithmic' or 'power' (got code %d)", p->method);
    ATL_REQUIRE(p->method == ATL_WIND_NONE || h_aux, "atl_wind_probe_host: method needs roughness / wnd_shear_exp");
    if (p->n_knots == 0) {  // no power curve: the extrapolated wind speed (kWindIdentity, as make_wind / wind_dispatch)
        ATL_REQUIRE(p->method == ATL_WIND_NONE || (p->to_height > 0 && p->from_height > 0 && std::isfinite(p->to_height) &&
                                                    std::isfinite(p->from_height)),
                    "atl_wind: heights must be positive and finite");
        std::vector<double> lds0(2 * size_t(kLogTabN));
        for (int i = 0; i < kLogTabN; ++i) log_table_entry(lds0.data(), i);
        const int method0 = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;
        auto run0 = [&](auto conv) {
            conv.wnd = h_wnd;
            conv.aux = h_aux;
            conv.S = n;
            conv.aux_static = 0;
            conv.method = method0;
            conv.to_height = p->to_height;
            conv.from_height = p->from_height;
            conv.log_ratio = log(p->to_height / p->from_height);
            conv.table = nullptr;
            conv.n_knots = conv.n_pad = conv.tab_doubles = conv.b0 = 0;
            conv.vmin = conv.vmax = conv.inv_w = 0.0;
            const auto cell = conv.cell_setup(0, true, true, lds0.data());
            for (int64_t i = 0; i < n; ++i) {
                typename decltype(conv)::Raw q;
                q.v.x = q.v.y = h_wnd[i];
                q.z.x = q.z.y = h_aux ? h_aux[i] : 0.0;
                h_out[i] = conv.compute(q, true, false, cell, lds0.data()).x;
            }
            return int(ATL_OK);
        };
        if (method0 == ATL_WIND_LOG) return run0(WindConvT<ATL_WIND_LOG, kWindIdentity>());
        if (method0 == ATL_WIND_POWER) return run0(WindConvT<ATL_WIND_POWER, kWindIdentity>());
        return run0(WindConvT<ATL_WIND_NONE, kWindIdentity>());
    }
    ATL_REQUIRE(p->n_knots >= 1 && p->n_knots <= kMaxKnots && p->h_V && p->h_POWn, "atl_wind_probe_host: bad power curve");
    std::vector<double> tbl;
    bool finite = true;
    int n_pad = 0;
    const int nk = wind_table_build(p->h_V, p->h_POWn, p->n_knots, tbl, &n_pad, &finite);
    ATL_REQUIRE(nk > 0, "wind speed 'V' in the turbine config is expected to be increasing");
    std::vector<double> grid;
    double inv_w = 0.0;
    int b0 = 0;
    const int n_grid = (finite && !getenv("ATLITE_HIP_WIND_NO_GRID")) ? wind_grid_build(tbl.data(), nk, n_pad, grid, &inv_w, &b0) : 0;
    const double vmin = tbl[0], vmax = tbl[size_t(nk - 1)];
    if (n_grid > 0) tbl = grid;
    // the "LDS" image of a block: power-curve table, then the log table
    std::vector<double> lds(tbl.size() + 2 * size_t(kLogTabN));
    memcpy(lds.data(), tbl.data(), tbl.size() * sizeof(double));
    for (int i = 0; i < kLogTabN; ++i) log_table_entry(lds.data() + tbl.size(), i);
    const int method = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;  // make_wind
    auto fill = [&](auto &c) {
        c.wnd = h_wnd;
        c.aux = h_aux;
        c.S = n;
        c.aux_static = 0;
        c.method = method;
        c.to_height = p->to_height;
        c.from_height = p->from_height;
        c.log_ratio = log(p->to_height / p->from_height);
        c.table = nullptr;
        c.n_knots = nk;
        c.n_pad = n_pad;
        c.tab_doubles = int(tbl.size());
        c.vmin = vmin;
        c.vmax = vmax;
        c.inv_w = n_grid > 0 ? inv_w : 0.0;
        c.b0 = b0;
    };
    auto run = [&](auto conv) {
        fill(conv);
        const auto cell = conv.cell_setup(0, true, true, lds.data());
        for (int64_t i = 0; i < n; ++i) {
            typename decltype(conv)::Raw q;
            q.v.x = q.v.y = h_wnd[i];
            q.z.x = q.z.y = h_aux ? h_aux[i] : 0.0;
            h_out[i] = conv.compute(q, true, false, cell, lds.data()).x;
        }
        return int(ATL_OK);
    };
    const bool heights_ok = p->to_height > 0 && p->from_height > 0 && std::isfinite(p->to_height) && std::isfinite(p->from_height);
    if (!finite || (method == ATL_WIND_LOG && !heights_ok)) return run(WindConvT<-1>());  // as wind_dispatch()
    if (n_grid > 0) {
        if (method == ATL_WIND_LOG) return run(WindConvT<ATL_WIND_LOG, kWindGrid>());
        if (method == ATL_WIND_POWER) return run(WindConvT<ATL_WIND_POWER, kWindGrid>());
        return run(WindConvT<ATL_WIND_NONE, kWindGrid>());
    }
    if (method == ATL_WIND_LOG) {
        if (n_pad == 16) return run(WindConvT<ATL_WIND_LOG, 4>());
        if (n_pad == 32) return run(WindConvT<ATL_WIND_LOG, 5>());
        if (n_pad == 128) return run(WindConvT<ATL_WIND_LOG, 7>());
        return run(WindConvT<ATL_WIND_LOG>());
    }
    if (method == ATL_WIND_POWER) return n_pad == 16 ? run(WindConvT<ATL_WIND_POWER, 4>()) : run(WindConvT<ATL_WIND_POWER>());
    return n_pad == 32 ? run(WindConvT<ATL_WIND_NONE, 5>()) : run(WindConvT<ATL_WIND_NONE>());
}

int atl_agg_selfcheck(int64_t n_cells, int64_t row_len, int tile_w, int64_t *n_tiles, int64_t *n_owned,
                      int64_t *n_errors) {
    ATL_REQUIRE(n_cells >= 0 && n_cells < (int64_t(1) << 31) && n_tiles && n_owned && n_errors,
                "atl_agg_selfcheck: bad argument");
    ATL_REQUIRE(tile_w == 16 || tile_w == 32 || tile_w == 64 || tile_w == 128, "atl_agg_selfcheck: tile_w must be 16, 32, 64 or 128");
    ATL_REQUIRE(row_len >= 0 && (row_len == 0 || n_cells % row_len == 0), "atl_agg_selfcheck: row_len does not divide n_cells");
    int l2 = 0;
    while ((2 << l2) < tile_w) ++l2;
    Layout L{row_len > 0 ? row_len : std::max<int64_t>(n_cells, 1), row_len > 0 ? n_cells / row_len : 1, l2};
    const int h = kLanes >> L.w2_log2;
    const int64_t ntx = tile_columns(L.X, L.Y, L.w2_log2), nty = (L.Y + h - 1) / h;
    *n_tiles = n_cells > 0 ? ntx * nty : 0;
    *n_owned = 0;
    *n_errors = 0;
    std::vector<uint8_t> seen(static_cast<size_t>(n_cells), 0);
    for (int64_t seg = 0; seg < *n_tiles; ++seg) {
        for (int lane = 0; lane < kLanes; ++lane) {
            const TileLane t = tile_lane_cells(L.X, L.Y, int32_t(ntx), L.w2_log2, int32_t(seg), lane);
            for (int k = 0; k < 2; ++k) {
                if (!(k ? t.v1 : t.v0)) continue;
                const int64_t c = t.c0 + k;
                if (c < 0 || c >= n_cells) {
                    ++*n_errors;  // the kernel would read outside the cube
                    continue;
                }
                ++*n_owned;
                if (seen[size_t(c)]++) ++*n_errors;  // owned twice
                int32_t loc = -1;
                const int64_t back = tile_of_cell(L, c, &loc);
                if (back != seg || loc != 2 * lane + k) ++*n_errors;  // plan builder and kernel disagree
            }
        }
    }
    for (int64_t c = 0; c < n_cells; ++c)
        if (!seen[size_t(c)]) ++*n_errors;  // never owned
    return ATL_OK;
}

int atl_agg_selfcheck_aligned(int64_t n_cells, int64_t row_len, int tile_w, int64_t *n_classes, int64_t *n_tiles, int64_t *n_owned,
                              int64_t *n_errors) {
    ATL_REQUIRE(n_cells >= 16 && n_cells % 16 != 0 && n_cells < (int64_t(1) << 27) && n_classes && n_tiles && n_owned && n_errors,
                "atl_agg_selfcheck_aligned: bad argument");
    ATL_REQUIRE(tile_w == 16 || tile_w == 32 || tile_w == 64 || tile_w == 128, "atl_agg_selfcheck_aligned: tile_w must be 16, 32, 64 or 128");
    ATL_REQUIRE(row_len >= 0 && (row_len == 0 || n_cells % row_len == 0), "atl_agg_selfcheck_aligned: row_len does not divide n_cells");
    int l2 = 0;
    while ((2 << l2) < tile_w) ++l2;
    const int64_t p = alignment_classes(n_cells);
    Layout L{row_len > 0 ? row_len : n_cells, row_len > 0 ? n_cells / row_len : 1, l2, int(p)};
    const int64_t ntx = layout_columns(L), tpc = layout_tiles_per_class(L);
    *n_classes = p;
    *n_tiles = p * tpc;
    *n_owned = 0;
    *n_errors = 0;
    std::vector<uint8_t> seen;
    for (int64_t r = 0; r < p; ++r) {
        const int64_t o = class_origin(L, r);
        if (o != ((r * n_cells) & 15)) ++*n_errors;
        seen.assign(static_cast<size_t>(n_cells), 0);
        for (int64_t seg = 0; seg < tpc; ++seg) {
            for (int lane = 0; lane < kLanes; ++lane) {
                const TileLane t = tile_lane_cells(L.X, L.Y, int32_t(ntx), L.w2_log2, int32_t(seg), lane, o);
                // the lane's 16-byte load sits at cell r * S + c0 of a cube whose slots are p * S cells apart: aligned in every
                // slot of the class; the first lane of a tile row starts a 128-byte line
                if ((t.v0 || t.v1) && ((r * n_cells + t.c0) & 1)) ++*n_errors;
                if ((t.v0 || t.v1) && (lane & ((1 << L.w2_log2) - 1)) == 0 && ((r * n_cells + t.c0) & 15)) ++*n_errors;
                for (int k = 0; k < 2; ++k) {
                    if (!(k ? t.v1 : t.v0)) continue;
                    const int64_t c = t.c0 + k;
                    if (c < 0 || c >= n_cells) {
                        ++*n_errors;  // the kernel would read outside the slot
                        continue;
                    }
                    ++*n_owned;
                    if (seen[size_t(c)]++) ++*n_errors;  // owned twice within the class
                    int32_t loc = -1;
                    const int64_t back = tile_of_cell(L, r * n_cells + c, &loc);  // the stacked column build_plan sees
                    if (back != r * tpc + seg || loc != 2 * lane + k) ++*n_errors;  // plan builder and kernel disagree
                }
            }
        }
        for (int64_t c = 0; c < n_cells; ++c)
            if (!seen[size_t(c)]) ++*n_errors;  // never owned
    }
    return ATL_OK;
}

int atl_agg_destroy(atl_agg *agg) {
    if (!agg) return ATL_OK;
    if (agg->ctx) {
        (void)hipSetDevice(agg->ctx->device);
        (void)hipStreamSynchronize(agg->ctx->stream);
    }
    for (void *p : agg->allocs) (void)dev_free(p);
    delete agg;
    return ATL_OK;
}

int atl_agg_info(const atl_agg *agg, int64_t *n_rows, int64_t *n_cells, int64_t *n_segments,
                 int64_t *n_partial_rows, int32_t *tile_w, int32_t *tile_h) {
    ATL_REQUIRE(agg, "atl_agg_info: agg is NULL");
    const bool aligned = agg->dev.shift_classes > 0;  // atl_agg_create_aligned: the matrix's own shape, not the stacked one
    if (n_rows) *n_rows = aligned ? agg->dev.shift_rows : agg->dev.n_rows;
    if (n_cells) *n_cells = agg->dev.n_cells;
    if (n_segments) *n_segments = agg->dev.n_segs;
    if (n_partial_rows) *n_partial_rows = agg->dev.n_prows;
    if (tile_w) *tile_w = 2 << agg->dev.w2_log2;
    if (tile_h) *tile_h = atl::kLanes >> agg->dev.w2_log2;
    return ATL_OK;
}

}  // extern "C"


// ---- the conversion calls with the slot stride as an ARGUMENT (round 5) ------------------------------------------------
// Rounds 3-5 exported a setter that made the stride context state a conversion call read: two calls whose coupling a forgotten
// reset broke silently (removed in round 6).  These entry points take it with the call (ld_cells: cells between the slots of
// the call's (T, S) input cubes, 0 = contiguous); inside the library it is still carried by the context for the call's duration.
namespace {
struct StrideScope {
    atl_ctx *c;
    int64_t old;
    StrideScope(atl_ctx *ctx, int64_t ld) : c(ctx), old(ctx->slot_stride) { c->slot_stride = ld; }
    ~StrideScope() { c->slot_stride = old; }
};
}  // namespace
#define ATL_LD_GUARD(name)                                                                      \
    ATL_REQUIRE(ctx && ld_cells >= 0, name ": ctx is NULL or the slot stride is negative");    \
    StrideScope stride_scope(ctx, ld_cells)

extern "C" {

int atl_spmm_csr_ld(atl_ctx *ctx, int64_t ld_cells, const atl_agg *agg, const double *d_dense, int64_t T, int64_t S, int time_agg,
                    double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_spmm_csr_ld");
    return atl_spmm_csr(ctx, agg, d_dense, T, S, time_agg, d_out, ld_out);
}
int atl_pv_convert_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                      int time_agg, double *d_out) {
    ATL_LD_GUARD("atl_pv_convert_ld");
    return atl_pv_convert(ctx, in, p, T, S, time_agg, d_out);
}
int atl_pv_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                                int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_pv_convert_aggregate_ld");
    return atl_pv_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
}
int atl_pv_day_map_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                      const atl_agg *agg, uint8_t *d_map, int64_t ld) {
    ATL_LD_GUARD("atl_pv_day_map_ld");
    return atl_pv_day_map(ctx, in, p, T, S, agg, d_map, ld);
}
int atl_wind_convert_ld(atl_ctx *ctx, int64_t ld_cells, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T, int64_t S,
                        int time_agg, double *d_out) {
    ATL_LD_GUARD("atl_wind_convert_ld");
    return atl_wind_convert(ctx, in, p, T, S, time_agg, d_out);
}
int atl_wind_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T,
                                  int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_wind_convert_aggregate_ld");
    return atl_wind_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
}
int atl_heat_demand_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_temperature, const atl_heat_params *p, int64_t T,
                               int64_t S, int time_agg, double *d_out) {
    ATL_LD_GUARD("atl_heat_demand_convert_ld");
    return atl_heat_demand_convert(ctx, d_temperature, p, T, S, time_agg, d_out);
}
int atl_heat_demand_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_temperature, const atl_heat_params *p,
                                         int64_t T, int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_heat_demand_convert_aggregate_ld");
    return atl_heat_demand_convert_aggregate(ctx, d_temperature, p, T, S, agg, time_agg, d_out, ld_out);
}
int atl_thermo_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_var, const atl_thermo_params *p, int64_t T, int64_t S,
                          int time_agg, double *d_out) {
    ATL_LD_GUARD("atl_thermo_convert_ld");
    return atl_thermo_convert(ctx, d_var, p, T, S, time_agg, d_out);
}
int atl_thermo_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_var, const atl_thermo_params *p, int64_t T,
                                    int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_thermo_convert_aggregate_ld");
    return atl_thermo_convert_aggregate(ctx, d_var, p, T, S, agg, time_agg, d_out, ld_out);
}
int atl_runoff_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_runoff, const double *d_height, int64_t T, int64_t S,
                          int time_agg, double *d_out) {
    ATL_LD_GUARD("atl_runoff_convert_ld");
    return atl_runoff_convert(ctx, d_runoff, d_height, T, S, time_agg, d_out);
}
int atl_runoff_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_runoff, const double *d_height, int64_t T,
                                    int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_LD_GUARD("atl_runoff_convert_aggregate_ld");
    return atl_runoff_convert_aggregate(ctx, d_runoff, d_height, T, S, agg, time_agg, d_out, ld_out);
}
int atl_agg_create_ld(atl_ctx *ctx, int64_t ld_cells, int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr,
                      const int32_t *h_indices, const double *h_data, atl_agg **out) {
    ATL_LD_GUARD("atl_agg_create_ld");
    return atl_agg_create(ctx, n_rows, n_cells, row_len, h_indptr, h_indices, h_data, out);
}
int atl_nc_read_slabs_ld(atl_ctx *ctx, int64_t ld_cells, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0,
                         double *const *d_outs, int n_threads) {
    ATL_LD_GUARD("atl_nc_read_slabs_ld");
    return atl_nc_read_slabs(ctx, f, n_vars, names, start0, count0, d_outs, n_threads);
}
int atl_nc_read_slab_ld(atl_ctx *ctx, int64_t ld_cells, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out,
                        int n_threads) {
    ATL_LD_GUARD("atl_nc_read_slab_ld");
    return atl_nc_read_slab(ctx, f, name, start0, count0, d_out, n_threads);
}

}  // extern "C"
