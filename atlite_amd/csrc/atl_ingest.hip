// Cutout-file ingest: NetCDF-4 (HDF5) chunks -> fp64 (time, cell) blocks in HBM.
// Replaces xr.open_dataset(path, chunks=...) + dask chunk reads for the inputs of the hot path
// (atlite/cutout.py:143,151-153; atlite/data.py:139,246-248).  Part of libatlite_hip.so (gfx950).
//
// Split of the work:
//   host threads : locate chunks (atl_h5.cpp), strip fletcher32, zlib inflate - the only inherently
//                  serial step - straight from the file mapping into pinned staging
//   copy stream  : one DMA per call, then k_unpack: byte un-shuffle, endian swap, widen to fp64,
//                  _FillValue / missing_value -> NaN, scale_factor / add_offset, scatter of the chunk
//                  grid into the (rows, cells) block the conversion kernels read
// so the CPU never touches the inflated bytes again and PCIe carries the narrow on-disk dtype.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "atl_h5.h"
#include "atl_internal.h"

using namespace atl;
using atl::h5::Attribute;
using atl::h5::Dataset;
using atl::h5::TypeClass;

struct atl_nc {
    h5::File file;
};

namespace {

// ---- dtype helpers ---------------------------------------------------------------------------------
int dtype_code(const h5::Datatype &t) {
    if (t.cls == TypeClass::Float) return t.size == 4 ? ATL_NC_F32 : t.size == 8 ? ATL_NC_F64 : ATL_NC_OTHER;
    if (t.cls == TypeClass::Fixed) {
        switch (t.size) {
            case 1: return t.is_signed ? ATL_NC_I8 : ATL_NC_U8;
            case 2: return t.is_signed ? ATL_NC_I16 : ATL_NC_U16;
            case 4: return t.is_signed ? ATL_NC_I32 : ATL_NC_U32;
            case 8: return t.is_signed ? ATL_NC_I64 : ATL_NC_U64;
        }
    }
    return ATL_NC_OTHER;
}

int dtype_size(int code) {
    switch (code) {
        case ATL_NC_I8: case ATL_NC_U8: return 1;
        case ATL_NC_I16: case ATL_NC_U16: return 2;
        case ATL_NC_F32: case ATL_NC_I32: case ATL_NC_U32: return 4;
        case ATL_NC_F64: case ATL_NC_I64: case ATL_NC_U64: return 8;
    }
    return 0;
}

// raw little-endian bits -> double, by dtype code (shared by host and device)
__host__ __device__ inline double bits_to_double(uint64_t b, int code) {
    switch (code) {
        case ATL_NC_F32: {
            uint32_t u = uint32_t(b);
            float f;
            memcpy(&f, &u, 4);
            return double(f);
        }
        case ATL_NC_F64: {
            double d;
            memcpy(&d, &b, 8);
            return d;
        }
        case ATL_NC_I8: return double(int8_t(b));
        case ATL_NC_I16: return double(int16_t(b));
        case ATL_NC_I32: return double(int32_t(b));
        case ATL_NC_I64: return double(int64_t(b));
        case ATL_NC_U8: return double(uint8_t(b));
        case ATL_NC_U16: return double(uint16_t(b));
        case ATL_NC_U32: return double(uint32_t(b));
        default: return double(b);
    }
}

struct Decode {  // CF conventions as xarray applies them: mask first, then scale
    int dtype = 0, esize = 0, big_endian = 0;
    int has_fill = 0, has_missing = 0, has_scale = 0;
    double fill = 0, missing = 0, scale = 1, offset = 0;
};

__host__ __device__ inline double cf_decode(double v, const Decode &d) {
#pragma clang fp contract(off)  // xarray: data *= scale_factor; data += add_offset - two roundings, no FMA
    if ((d.has_fill && v == d.fill) || (d.has_missing && v == d.missing)) return __builtin_nan("");
    if (!d.has_scale) return v;
    const double scaled = v * d.scale;
    return scaled + d.offset;
}

bool attr_double(const h5::File &f, const Attribute *a, double *out, int64_t max_n, int64_t *n) {
    *n = 0;
    if (!a) return false;
    const int code = dtype_code(a->type);
    if (code == ATL_NC_OTHER) return false;
    const int64_t cnt = int64_t(a->nbytes / a->type.size);
    for (int64_t i = 0; i < cnt && i < max_n; ++i) {
        uint64_t b = 0;
        for (uint32_t k = 0; k < a->type.size; ++k) {
            const uint32_t src = a->type.big_endian ? a->type.size - 1 - k : k;
            b |= uint64_t(a->data[i * a->type.size + src]) << (8 * k);
        }
        out[i] = bits_to_double(b, code);
    }
    *n = cnt;
    return cnt > 0;
}

Decode decode_of(const h5::File &f, const Dataset &d) {
    Decode dc;
    dc.dtype = dtype_code(d.type);
    dc.esize = int(d.type.size);
    dc.big_endian = d.type.big_endian;
    double v;
    int64_t n;
    if (attr_double(f, d.attr("_FillValue"), &v, 1, &n)) dc.has_fill = 1, dc.fill = v;
    if (attr_double(f, d.attr("missing_value"), &v, 1, &n)) dc.has_missing = 1, dc.missing = v;
    if (attr_double(f, d.attr("scale_factor"), &v, 1, &n)) dc.has_scale = 1, dc.scale = v;
    if (attr_double(f, d.attr("add_offset"), &v, 1, &n)) dc.has_scale = 1, dc.offset = v;
    return dc;
}

// ---- variable geometry, normalised to 3 dims (rows, d1, d2) ----------------------------------------------
struct Geometry {
    int64_t shape[3] = {1, 1, 1};
    int64_t chunk[3] = {1, 1, 1};
    int64_t grid[3] = {1, 1, 1};
    int64_t chunk_elems = 1, row_elems = 1;
};

int geometry_of(const Dataset &d, Geometry *g, const char *who) {
    const int rank = int(d.shape.size());
    ATL_REQUIRE(rank >= 1 && rank <= 3, "%s: variable '%s' has %d dimensions; 1 to 3 are supported", who,
                d.name.c_str(), rank);
    if (dtype_code(d.type) == ATL_NC_OTHER) {
        set_error("%s: variable '%s' is not numeric", who, d.name.c_str());
        return ATL_E_UNSUPPORTED;
    }
    if (d.layout < 0 || d.layout > 2) {
        set_error("%s: variable '%s' uses a storage layout / chunk index this reader does not support "
                  "(extensible-array or v2-B-tree chunk index, or a pre-1.6 layout); rewrite the file with "
                  "libver bounds <= v18 (NetCDF-4 default)", who, d.name.c_str());
        return ATL_E_UNSUPPORTED;
    }
    for (int i = 0; i < rank; ++i) {
        g->shape[i] = int64_t(d.shape[i]);
        g->chunk[i] = int64_t(d.chunk[i]);
        g->grid[i] = d.layout == 2 ? int64_t(d.grid[i]) : 1;
    }
    g->chunk_elems = g->chunk[0] * g->chunk[1] * g->chunk[2];
    g->row_elems = g->shape[1] * g->shape[2];
    return ATL_OK;
}

// ---- device side ------------------------------------------------------------------------------------
struct UnpackDesc {
    int64_t src_off;   // byte offset of the chunk payload in the raw buffer
    int64_t org[3];    // element origin of the chunk in the variable
    int64_t dim[3];    // chunk extent (elements, as stored)
    int32_t shuffled;  // payload is byte-shuffled (HDF5 shuffle filter)
    int32_t missing;   // chunk was never written -> _FillValue / NaN
};

struct UnpackParams {
    int64_t shape1, shape2;   // trailing dims of the variable
    int64_t ld;               // elements between the rows of the output block (>= shape1 * shape2)
    int64_t r0, r1;           // rows wanted
    Decode dec;
};

__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ raw, const UnpackDesc *__restrict__ desc,
                                                UnpackParams p, double *__restrict__ out) {
    const UnpackDesc d = desc[blockIdx.y];
    const int64_t n = d.dim[0] * d.dim[1] * d.dim[2];
    const int64_t plane = d.dim[1] * d.dim[2];
    const int es = p.dec.esize;
    const uint8_t *src = raw + d.src_off;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t ct = e / plane, rem = e - ct * plane;
        const int64_t cy = rem / d.dim[2], cx = rem - cy * d.dim[2];
        const int64_t t = d.org[0] + ct, y = d.org[1] + cy, x = d.org[2] + cx;
        if (t < p.r0 || t >= p.r1 || y >= p.shape1 || x >= p.shape2) continue;
        double v;
        if (d.missing) {
            v = p.dec.has_fill ? cf_decode(p.dec.fill, p.dec) : __builtin_nan("");
        } else {
            uint64_t b = 0;
            if (d.shuffled) {
                // byte k of element e sits at k*n + e: es coalesced byte streams
                for (int k = 0; k < es; ++k) {
                    const int dstk = p.dec.big_endian ? es - 1 - k : k;
                    b |= uint64_t(src[int64_t(k) * n + e]) << (8 * dstk);
                }
            } else if (es == 4 && !p.dec.big_endian) {
                b = reinterpret_cast<const uint32_t *>(src)[e];
            } else if (es == 8 && !p.dec.big_endian) {
                b = reinterpret_cast<const uint64_t *>(src)[e];
            } else {
                for (int k = 0; k < es; ++k) {
                    const int dstk = p.dec.big_endian ? es - 1 - k : k;
                    b |= uint64_t(src[e * es + k]) << (8 * dstk);
                }
            }
            v = cf_decode(bits_to_double(b, p.dec.dtype), p.dec);
        }
        out[(t - p.r0) * p.ld + y * p.shape2 + x] = v;
    }
}

// ---- per-context staging: two slots, each {pinned host, device raw, descriptor buffers, event} -------------
struct Slot {
    uint8_t *h = nullptr;
    uint8_t *d = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct IngestState {
    Slot slot[2];
    unsigned calls = 0;
};

void ingest_free(void *p) {
    IngestState *s = static_cast<IngestState *>(p);
    for (Slot &sl : s->slot) {
        if (sl.ev) {
            if (sl.pending) (void)hipEventSynchronize(sl.ev);
            (void)hipEventDestroy(sl.ev);
        }
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d) (void)dev_free(sl.d);
    }
    delete s;
}

int slot_acquire(atl_ctx *ctx, size_t bytes, Slot **out) {
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->ingest) {
        ctx->ingest = new IngestState();
        ctx->ingest_free = ingest_free;
    }
    IngestState *st = static_cast<IngestState *>(ctx->ingest);
    Slot &sl = st->slot[st->calls++ & 1];
    if (!sl.ev) ATL_HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.pending) {
        ATL_HIP_TRY(hipEventSynchronize(sl.ev));  // the previous user of this slot has left the copy stream
        sl.pending = false;
    }
    if (sl.bytes < bytes) {
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d) (void)dev_free(sl.d);
        sl.h = nullptr;
        sl.d = nullptr;
        sl.bytes = 0;
        const size_t want = align_up(bytes + bytes / 4, size_t(1) << 20);
        ATL_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sl.h), want, hipHostMallocDefault));
        ATL_HIP_TRY(dev_malloc(reinterpret_cast<void **>(&sl.d), want));
        sl.bytes = want;
    }
    *out = &sl;
    return ATL_OK;
}

// stage `payload` bytes (in sl->h, or in the caller's pinned buffer h_payload) + descriptors, copy, decode
int submit(atl_ctx *ctx, Slot *sl, size_t payload, const std::vector<UnpackDesc> &descs, const UnpackParams &p,
           int64_t max_chunk_elems, double *d_out, const void *h_payload = nullptr) {
    hipStream_t cs;
    int rc = copy_stream_of(ctx, &cs);
    if (rc) return rc;
    const size_t desc_off = align_up(payload, 256);
    memcpy(sl->h + desc_off, descs.data(), descs.size() * sizeof(UnpackDesc));
    if (h_payload) {
        ATL_HIP_TRY(hipMemcpyAsync(sl->d, h_payload, payload, hipMemcpyHostToDevice, cs));
        ATL_HIP_TRY(hipMemcpyAsync(sl->d + desc_off, sl->h + desc_off, descs.size() * sizeof(UnpackDesc),
                                   hipMemcpyHostToDevice, cs));
    } else {
        ATL_HIP_TRY(hipMemcpyAsync(sl->d, sl->h, desc_off + descs.size() * sizeof(UnpackDesc), hipMemcpyHostToDevice, cs));
    }
    const UnpackDesc *d_desc = reinterpret_cast<const UnpackDesc *>(sl->d + desc_off);
    const unsigned bx = unsigned(std::min<int64_t>((max_chunk_elems + 255) / 256, 2048));
    for (size_t c0 = 0; c0 < descs.size(); c0 += 32768) {
        const unsigned by = unsigned(std::min<size_t>(descs.size() - c0, 32768));
        hipLaunchKernelGGL(k_unpack, dim3(std::max(bx, 1u), by), dim3(256), 0, cs, sl->d, d_desc + c0, p, d_out);
    }
    ATL_HIP_TRY(hipGetLastError());
    ATL_HIP_TRY(hipEventRecord(sl->ev, cs));
    sl->pending = true;
    return ATL_OK;
}

// CPUs this process may actually use: the cgroup CPU quota (containers routinely expose all host
// threads but grant a fraction of them), else the hardware thread count
int usable_cpus() {
    static const int cached = [] {
        int hw = int(std::max(1u, std::thread::hardware_concurrency()));
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1) period = 0;
                fclose(g);
            }
        }
        if (quota > 0 && period > 0) hw = int(std::min<long long>(hw, (quota + period - 1) / period));
        return std::max(1, hw);
    }();
    return cached;
}

int pick_threads(int requested, size_t n_items) {
    int n = requested;
    if (n <= 0) {
        // a few more threads than CPUs smooths over stragglers; measured best on a 16-CPU quota: 32
        n = std::min(2 * usable_cpus(), 128);
        if (const char *e = getenv("ATLITE_HIP_IO_THREADS")) n = std::max(1, atoi(e));
    }
    return int(std::max<size_t>(1, std::min<size_t>(size_t(n), n_items)));
}

// Persistent worker pool: a read call fans its chunks out to up to 128 threads several times per
// slab, so the threads are created once per process and parked on a condition variable in between.
class Pool {
   public:
    static Pool &get() {
        static Pool p;
        return p;
    }
    // run body() on the caller + up to (n_threads - 1) workers; returns when all have finished
    void run(int n_threads, const std::function<void()> &body) {
        std::lock_guard<std::mutex> serial(run_m_);
        const int helpers = std::max(0, n_threads - 1);
        {
            std::unique_lock<std::mutex> lk(m_);
            while (int(workers_.size()) < helpers) workers_.emplace_back([this] { loop(); });
            body_ = &body;
            want_ = helpers;
            started_ = finished_ = 0;
            ++gen_;
        }
        cv_.notify_all();
        body();
        std::unique_lock<std::mutex> lk(m_);
        want_ = started_;  // late wakers find nothing to do
        done_.wait(lk, [this] { return finished_ == started_; });
        body_ = nullptr;
    }

   private:
    Pool() = default;
    ~Pool() {
        {
            std::unique_lock<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (!body_ || started_ >= want_) continue;
            ++started_;
            const std::function<void()> *b = body_;
            lk.unlock();
            (*b)();
            lk.lock();
            ++finished_;
            if (finished_ == started_) done_.notify_all();
        }
    }
    std::mutex run_m_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void()> *body_ = nullptr;
    int want_ = 0, started_ = 0, finished_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// run fn(i) for i in [0, n) on up to n_threads threads; first failure wins (message carried over,
// since atl_last_error is thread-local)
template <class F>
int parallel_for(size_t n, int n_threads, F fn) {
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    std::string msg;
    std::atomic<bool> have_msg{false};
    std::function<void()> body = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n || err.load()) return;
            int rc;
            try {  // an exception must not leave a worker thread (std::terminate): a chunk buffer that cannot be allocated
                rc = fn(i);
            } catch (const std::bad_alloc &) {
                set_error("out of host memory while reading chunk %zu", i);
                rc = ATL_E_NOMEM;
            } catch (const std::exception &e) {
                set_error("%s", e.what());
                rc = ATL_E_INVALID;
            }
            if (rc) {
                int expect = 0;
                if (err.compare_exchange_strong(expect, rc)) {
                    msg = atl_last_error();
                    have_msg = true;
                }
            }
        }
    };
    if (n_threads <= 1 || n <= 1) {
        body();
    } else {
        Pool::get().run(n_threads, body);
    }
    if (err.load() && have_msg) set_error("%s", msg.c_str());
    return err.load();
}

struct Selection {  // chunks of a variable overlapping rows [r0, r1)
    std::vector<size_t> lin;          // linear chunk index
    std::vector<UnpackDesc> desc;
};

void select_chunks(const Geometry &g, int64_t r0, int64_t r1, int64_t chunk_bytes, Selection *s) {
    const int64_t g0a = r0 / g.chunk[0], g0b = (r1 - 1) / g.chunk[0];
    int64_t off = 0;
    for (int64_t a = g0a; a <= g0b; ++a)
        for (int64_t b = 0; b < g.grid[1]; ++b)
            for (int64_t c = 0; c < g.grid[2]; ++c) {
                UnpackDesc d{};
                d.src_off = off;
                d.org[0] = a * g.chunk[0];
                d.org[1] = b * g.chunk[1];
                d.org[2] = c * g.chunk[2];
                d.dim[0] = g.chunk[0];
                d.dim[1] = g.chunk[1];
                d.dim[2] = g.chunk[2];
                s->lin.push_back(size_t((a * g.grid[1] + b) * g.grid[2] + c));
                s->desc.push_back(d);
                off += align_up(size_t(chunk_bytes), 16);
            }
}

int lookup(atl_nc *f, const char *name, const Dataset **out, const char *who) {
    ATL_REQUIRE(f && name, "%s: bad argument", who);
    const Dataset *d = f->file.find(name);
    ATL_REQUIRE(d, "%s: no variable '%s' in the file", who, name);
    *out = d;
    return ATL_OK;
}

int copy_text(const std::string &s, char *buf, int64_t buflen, int64_t *needed) {
    if (needed) *needed = int64_t(s.size()) + 1;
    if (buf && buflen > 0) {
        const size_t n = std::min<size_t>(s.size(), size_t(buflen - 1));
        memcpy(buf, s.data(), n);
        buf[n] = '\0';
    }
    return ATL_OK;
}

}  // namespace

extern "C" {

int atl_nc_open(const char *path, atl_nc **out) {
    ATL_REQUIRE(path && out, "atl_nc_open: bad argument");
    *out = nullptr;
    atl_nc *f = new atl_nc();
    const int rc = f->file.open(path);
    if (rc) {
        delete f;
        return rc;
    }
    *out = f;
    return ATL_OK;
}

int atl_nc_close(atl_nc *f) {
    delete f;
    return ATL_OK;
}

int atl_nc_list(atl_nc *f, char *buf, int64_t buflen, int64_t *needed) {
    ATL_REQUIRE(f, "atl_nc_list: bad argument");
    std::string s;
    for (auto &d : f->file.datasets()) {
        if (!s.empty()) s += '\n';
        s += d.name;
    }
    return copy_text(s, buf, buflen, needed);
}

int atl_nc_inquire(atl_nc *f, const char *name, atl_nc_var *info) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_inquire");
    if (rc) return rc;
    ATL_REQUIRE(info, "atl_nc_inquire: info is NULL");
    memset(info, 0, sizeof *info);
    info->ndim = int32_t(std::min<size_t>(d->shape.size(), 4));
    const Decode dc = decode_of(f->file, *d);
    info->dtype = dc.dtype;
    info->elem_size = int32_t(d->type.size);
    info->big_endian = d->type.big_endian;
    for (int i = 0; i < info->ndim; ++i) {
        info->shape[i] = int64_t(d->shape[i]);
        info->chunk[i] = i < int(d->chunk.size()) ? int64_t(d->chunk[i]) : int64_t(d->shape[i]);
    }
    info->layout = d->layout;
    for (auto &fl : d->filters) {
        if (fl.id == 2) info->shuffle = 1;
        if (fl.id == 1) info->deflate = 1 + (fl.params.empty() ? 0 : int(fl.params[0]));
        if (fl.id == 3) info->fletcher32 = 1;
    }
    info->has_scale = dc.has_scale;
    info->has_fill = dc.has_fill;
    info->has_missing = dc.has_missing;
    info->scale_factor = dc.scale;
    info->add_offset = dc.offset;
    info->fill_value = dc.fill;
    info->missing_value = dc.missing;
    info->n_chunks = d->layout == 2 ? int64_t(d->chunks.size()) : 1;
    if (d->layout == 2) {
        for (auto &c : d->chunks) info->stored_bytes += int64_t(c.size);
    } else {
        info->stored_bytes = int64_t(d->contiguous_size);
    }
    return ATL_OK;
}

int atl_nc_dims(atl_nc *f, const char *name, char *buf, int64_t buflen, int64_t *needed) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_dims");
    if (rc) return rc;
    std::string s;
    for (size_t i = 0; i < d->dims.size(); ++i) {
        if (i) s += '\n';
        s += d->dims[i];
    }
    return copy_text(s, buf, buflen, needed);
}

static int find_attr(atl_nc *f, const char *var, const char *att, const Attribute **out, const char *who) {
    ATL_REQUIRE(f && att, "%s: bad argument", who);
    *out = nullptr;
    if (!var || !*var) {
        for (auto &a : f->file.global_attrs())
            if (a.name == att) *out = &a;
        return ATL_OK;
    }
    const Dataset *d;
    int rc = lookup(f, var, &d, who);
    if (rc) return rc;
    *out = d->attr(att);
    return ATL_OK;
}

int atl_nc_att_text(atl_nc *f, const char *var, const char *att, char *buf, int64_t buflen, int64_t *needed) {
    const Attribute *a;
    int rc = find_attr(f, var, att, &a, "atl_nc_att_text");
    if (rc) return rc;
    if (needed) *needed = 0;
    if (buf && buflen > 0) buf[0] = '\0';
    if (!a) return ATL_OK;
    std::string s;
    if (a->type.cls == TypeClass::String) {
        s.assign(reinterpret_cast<const char *>(a->data), size_t(a->nbytes));
        const size_t z = s.find('\0');
        if (z != std::string::npos) s.resize(z);
    } else if (a->type.cls == TypeClass::VlenStr && a->nbytes >= uint64_t(8 + f->file.off_size())) {
        const uint8_t *p = nullptr;
        uint64_t n = 0;
        uint32_t cnt = 0;
        if (f->file.vlen_payload(a->data, &p, &n, &cnt)) s.assign(reinterpret_cast<const char *>(p), size_t(std::min<uint64_t>(n, cnt)));
    } else {
        ATL_REQUIRE(false, "atl_nc_att_text: attribute '%s' is not a string", att);
    }
    return copy_text(s, buf, buflen, needed);
}

int atl_nc_att_double(atl_nc *f, const char *var, const char *att, double *out, int64_t max_n, int64_t *n) {
    const Attribute *a;
    int rc = find_attr(f, var, att, &a, "atl_nc_att_double");
    if (rc) return rc;
    ATL_REQUIRE(n && (out || max_n == 0), "atl_nc_att_double: bad argument");
    *n = 0;
    if (!a) return ATL_OK;
    ATL_REQUIRE(attr_double(f->file, a, out, max_n, n), "atl_nc_att_double: attribute '%s' is not numeric", att);
    return ATL_OK;
}

int atl_nc_read_host(atl_nc *f, const char *name, int64_t start0, int64_t count0, double *out) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_read_host");
    if (rc) return rc;
    Geometry g;
    rc = geometry_of(*d, &g, "atl_nc_read_host");
    if (rc) return rc;
    ATL_REQUIRE(start0 >= 0 && count0 >= 0 && start0 + count0 <= g.shape[0],
                "atl_nc_read_host: rows [%lld, %lld) outside '%s' (%lld rows)", (long long)start0,
                (long long)(start0 + count0), name, (long long)g.shape[0]);
    if (count0 == 0 || g.row_elems == 0) return ATL_OK;
    ATL_REQUIRE(out, "atl_nc_read_host: out is NULL");
    const Decode dc = decode_of(f->file, *d);
    const int es = dc.esize;
    const int64_t r0 = start0, r1 = start0 + count0;
    auto element = [&](const uint8_t *src, int64_t n, int64_t e, bool shuffled) {
        uint64_t b = 0;
        for (int k = 0; k < es; ++k) {
            const int dstk = dc.big_endian ? es - 1 - k : k;
            const uint8_t byte = shuffled ? src[int64_t(k) * n + e] : src[e * es + k];
            b |= uint64_t(byte) << (8 * dstk);
        }
        return cf_decode(bits_to_double(b, dc.dtype), dc);
    };
    if (d->layout != 2) {
        const uint8_t *src = d->layout == 0 ? d->compact : f->file.base() + d->contiguous_addr;
        const uint64_t need = uint64_t(g.shape[0]) * g.row_elems * es;
        if (d->layout == 1 && (d->contiguous_addr == 0 || d->contiguous_addr > f->file.size() ||
                               need > f->file.size() - d->contiguous_addr || d->contiguous_size < need)) {
            // never written (address undefined) -> fill value
            const double v = dc.has_fill ? cf_decode(dc.fill, dc) : __builtin_nan("");
            std::fill(out, out + count0 * g.row_elems, v);
            return ATL_OK;
        }
        ATL_REQUIRE(d->layout == 1 || d->contiguous_size >= need, "atl_nc_read_host: compact data of '%s' is short", name);
        const int64_t n = g.shape[0] * g.row_elems;
        for (int64_t i = 0; i < count0 * g.row_elems; ++i) out[i] = element(src, n, r0 * g.row_elems + i, false);
        return ATL_OK;
    }
    Selection sel;
    const int64_t chunk_bytes = g.chunk_elems * es;
    select_chunks(g, r0, r1, chunk_bytes, &sel);
    const int nt = pick_threads(0, sel.lin.size());
    return parallel_for(sel.lin.size(), nt, [&](size_t i) -> int {
        const h5::Chunk &c = d->chunks[sel.lin[i]];
        const UnpackDesc &ds = sel.desc[i];
        std::vector<uint8_t> tmp;
        bool shuffled = false;
        const bool missing = c.size == 0;
        if (!missing) {
            tmp.resize(size_t(chunk_bytes));
            const int e = h5::chunk_inflate(*d, c, f->file.base(), -1, tmp.data(), uint64_t(chunk_bytes), &shuffled);
            if (e) return e;
        }
        const double fillv = dc.has_fill ? cf_decode(dc.fill, dc) : __builtin_nan("");
        for (int64_t ct = 0; ct < ds.dim[0]; ++ct) {
            const int64_t t = ds.org[0] + ct;
            if (t < r0 || t >= r1) continue;
            for (int64_t cy = 0; cy < ds.dim[1]; ++cy) {
                const int64_t y = ds.org[1] + cy;
                if (y >= g.shape[1]) break;
                for (int64_t cx = 0; cx < ds.dim[2]; ++cx) {
                    const int64_t x = ds.org[2] + cx;
                    if (x >= g.shape[2]) break;
                    const int64_t e = (ct * ds.dim[1] + cy) * ds.dim[2] + cx;
                    out[(t - r0) * g.row_elems + y * g.shape[2] + x] =
                        missing ? fillv : element(tmp.data(), g.chunk_elems, e, shuffled);
                }
            }
        }
        return ATL_OK;
    });
}

int atl_nc_read_slab(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out,
                     int n_threads) {
    ATL_REQUIRE(ctx, "atl_nc_read_slab: ctx is NULL");
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_read_slab");
    if (rc) return rc;
    Geometry g;
    rc = geometry_of(*d, &g, "atl_nc_read_slab");
    if (rc) return rc;
    ATL_REQUIRE(start0 >= 0 && count0 >= 0 && start0 + count0 <= g.shape[0],
                "atl_nc_read_slab: rows [%lld, %lld) outside '%s' (%lld rows)", (long long)start0,
                (long long)(start0 + count0), name, (long long)g.shape[0]);
    if (count0 == 0 || g.row_elems == 0) return ATL_OK;
    ATL_REQUIRE(d_out, "atl_nc_read_slab: d_out is NULL");
    const Decode dc = decode_of(f->file, *d);
    const int es = dc.esize;
    const int64_t r0 = start0, r1 = start0 + count0;
    UnpackParams p{};
    p.shape1 = g.shape[1];
    p.shape2 = g.shape[2];
    // rows of the output block: contiguous unless the context asks for padded slots (atl_set_slot_stride) and the
    // variable has rows to pad (a (time, y, x) cube)
    p.ld = g.shape[1] * g.shape[2];
    if (ctx->slot_stride > 0 && d->shape.size() == 3) {
        ATL_REQUIRE(ctx->slot_stride >= p.ld, "atl_nc_read_slab: slot stride %lld is smaller than a row of %lld cells",
                    (long long)ctx->slot_stride, (long long)p.ld);
        p.ld = ctx->slot_stride;
    }
    p.r0 = r0;
    p.r1 = r1;
    p.dec = dc;

    Selection sel;
    int64_t max_elems;
    size_t payload;
    Slot *sl = nullptr;
    if (d->layout != 2) {
        // compact / contiguous: the wanted rows are one pseudo-chunk, copied in parallel slices
        const uint64_t row_bytes = uint64_t(g.row_elems) * es;
        const uint64_t need = uint64_t(g.shape[0]) * row_bytes;
        UnpackDesc ds{};
        ds.org[0] = r0;
        ds.dim[0] = count0;
        ds.dim[1] = g.shape[1];
        ds.dim[2] = g.shape[2];
        const bool written = d->layout == 0 ? d->contiguous_size >= need
                                            : (d->contiguous_addr != 0 && d->contiguous_addr <= f->file.size() &&
                                               need <= f->file.size() - d->contiguous_addr && d->contiguous_size >= need);
        ds.missing = !written;
        sel.desc.push_back(ds);
        payload = written ? size_t(count0 * row_bytes) : 16;
        max_elems = count0 * g.row_elems;
        rc = slot_acquire(ctx, align_up(payload, 256) + sizeof(UnpackDesc), &sl);
        if (rc) return rc;
        if (written) {
            const uint8_t *src = (d->layout == 0 ? d->compact : f->file.base() + d->contiguous_addr) + r0 * row_bytes;
            const size_t slice = size_t(4) << 20;
            const size_t ns = (payload + slice - 1) / slice;
            rc = parallel_for(ns, pick_threads(n_threads, ns), [&](size_t i) -> int {
                const size_t a = i * slice, b = std::min(payload, a + slice);
                memcpy(sl->h + a, src + a, b - a);
                return ATL_OK;
            });
            if (rc) return rc;
        }
    } else {
        const int64_t chunk_bytes = g.chunk_elems * es;
        select_chunks(g, r0, r1, chunk_bytes, &sel);
        payload = sel.desc.size() * align_up(size_t(chunk_bytes), 16);
        max_elems = g.chunk_elems;
        rc = slot_acquire(ctx, align_up(payload, 256) + sel.desc.size() * sizeof(UnpackDesc), &sl);
        if (rc) return rc;
        rc = parallel_for(sel.lin.size(), pick_threads(n_threads, sel.lin.size()), [&](size_t i) -> int {
            const h5::Chunk &c = d->chunks[sel.lin[i]];
            UnpackDesc &ds = sel.desc[i];
            if (c.size == 0) {
                ds.missing = 1;
                return ATL_OK;
            }
            bool shuffled = false;
            const int e = h5::chunk_inflate(*d, c, f->file.base(), f->file.fd(), sl->h + ds.src_off, uint64_t(chunk_bytes), &shuffled);
            ds.shuffled = shuffled;
            return e;
        });
        if (rc) return rc;
    }
    return submit(ctx, sl, payload, sel.desc, p, max_elems, d_out);
}

int atl_inflate_probe(const void *h_src, size_t src_n, void *h_dst, size_t dst_n, int which, int64_t *ns) {
    ATL_REQUIRE(h_src && (h_dst || dst_n == 0) && which >= 0 && which <= 2, "atl_inflate_probe: bad argument");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ATL_OK;
    const uint8_t *src = static_cast<const uint8_t *>(h_src);
    uint8_t *dst = static_cast<uint8_t *>(h_dst);
    bool done = false;
    if (which != 1) {
        done = h5::fast_inflate_zlib(src, src_n, dst, dst_n) == 0;
        if (!done && which == 0) {
            set_error("atl_inflate_probe: the fast decoder declined the stream");
            rc = ATL_E_UNSUPPORTED;
        }
    }
    if (!done && which != 0) {
        uLongf out_n = uLongf(dst_n);
        const int z = uncompress(dst, &out_n, src, uLong(src_n));
        if (z != Z_OK || out_n != dst_n) {
            set_error("atl_inflate_probe: zlib rc %d, %llu of %llu bytes", z, (unsigned long long)out_n,
                      (unsigned long long)dst_n);
            rc = ATL_E_INVALID;
        }
    }
    if (ns) *ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int atl_upload_convert_async(atl_ctx *ctx, double *d_dst, const void *h_src, int dtype, int64_t n) {
    return atl_upload_convert_2d_async(ctx, d_dst, 1, h_src, dtype, n, 1);
}

int atl_upload_convert_2d_async(atl_ctx *ctx, double *d_dst, int64_t ld_cells, const void *h_src, int dtype, int64_t rows,
                                int64_t cols) {
    ATL_REQUIRE(ctx && rows >= 0 && cols >= 0 && ld_cells >= cols, "atl_upload_convert_async: bad argument");
    const int64_t n = rows * cols;
    if (n == 0) return ATL_OK;
    ATL_REQUIRE(d_dst && h_src, "atl_upload_convert_async: NULL buffer");
    const int es = dtype_size(dtype);
    ATL_REQUIRE(es > 0, "atl_upload_convert_async: unknown dtype code %d", dtype);
    const size_t payload = size_t(n) * es;
    Slot *sl = nullptr;
    int rc = slot_acquire(ctx, align_up(payload, 256) + sizeof(UnpackDesc), &sl);
    if (rc) return rc;
    // page-locked source (atl_host_register / Dataset.pin): DMA straight from it; pageable: gather
    // through the pinned staging on host threads (a pageable hipMemcpyAsync would serialise the stream)
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, h_src) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned) {
        const size_t slice = size_t(4) << 20;
        const size_t ns = (payload + slice - 1) / slice;
        const uint8_t *src = static_cast<const uint8_t *>(h_src);
        rc = parallel_for(ns, pick_threads(0, ns), [&](size_t i) -> int {
            const size_t a = i * slice, b = std::min(payload, a + slice);
            memcpy(sl->h + a, src + a, b - a);
            return ATL_OK;
        });
        if (rc) return rc;
    }
    UnpackParams p{};  // a (rows, 1, cols) variable in one "chunk": element (r, 0, c) goes to d_dst[r * ld + c]
    p.shape1 = 1;
    p.shape2 = cols;
    p.ld = ld_cells;
    p.r0 = 0;
    p.r1 = rows;
    p.dec.dtype = dtype;
    p.dec.esize = es;
    UnpackDesc ds{};
    ds.dim[0] = rows;
    ds.dim[1] = 1;
    ds.dim[2] = cols;
    std::vector<UnpackDesc> descs{ds};
    return submit(ctx, sl, payload, descs, p, n, d_dst, pinned ? h_src : nullptr);
}

}  // extern "C"
