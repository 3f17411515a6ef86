// Multi-GPU collectives of the hot path, for hosts that do not go through torch.distributed.  The time axis
// is sharded, so the only exchanges are an all-gather of the small (shapes x T_r) result
// blocks and an all-reduce of time sums.  Two transports behind one atl_comm:
//   * RCCL over xGMI (atl_comm_init): one rank per process or per thread; librccl.so.1 is opened lazily, the
//     library loads (and every single-GPU entry point works) without it;
//   * in-process peer copies (atl_comm_init_local): the ranks are host threads of ONE process that share an
//     atl_comm_group; every rank PULLS its peers' blocks with hipMemcpyPeerAsync on its own stream - on a node
//     that is seven point-to-point xGMI links read at once, no ring - ordered by events, with a host
//     rendezvous that times out instead of hanging.  Ranks may share a device (how a one-GPU box runs the
//     N-rank code, tests/test_gpu_multidevice.py).
// What is packed, gathered and placed is the same code for both.  The reference has no distributed path to
// mirror.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <rccl/rccl.h>

#include "atl_internal.h"

using namespace atl;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (diagnostics, single-thread initialisation of several devices): absent symbols leave these NULL
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return ATL_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("RCCL is not available: %s", dlerror());
        return ATL_E_UNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
    r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
        set_error("librccl is missing a required symbol");
        return ATL_E_UNSUPPORTED;
    }
    g_rccl = r;
    return ATL_OK;
}

#define ATL_NCCL_TRY(expr)                                                                      \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r__));                      \
            return ATL_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

// ---- placement: ONE definition of where an element of the gathered buffer goes ----------------------
// gathered [rank][n][Tmax] (rank r uses its first len_r = off[r+1] - off[r] columns) -> out[n][off[r] + t].
// The kernel and the host instantiation (atl_gather_place_v_host: what the CPU tests run) share it.
constexpr int kMaxRanks = 64;
struct RankOffsets {
    int64_t off[kMaxRanks + 1];
};
struct Placement {
    int64_t src, dst;
    bool live;
};
__host__ __device__ inline Placement gather_placement(const RankOffsets &ro, int r, int64_t n, int64_t t, int64_t N,
                                                      int64_t Tmax, int64_t ld_out) {
    Placement p;
    p.live = t < ro.off[r + 1] - ro.off[r];
    p.src = (int64_t(r) * N + n) * Tmax + t;
    p.dst = n * ld_out + ro.off[r] + t;
    return p;
}

__global__ __launch_bounds__(256) void k_gather_place_v(const double *__restrict__ g, RankOffsets ro, int64_t N,
                                                        int64_t Tmax, double *__restrict__ out, int64_t ld_out) {
    const Placement p = gather_placement(ro, int(blockIdx.z), int64_t(blockIdx.y),
                                         int64_t(blockIdx.x) * 256 + threadIdx.x, N, Tmax, ld_out);
    if (p.live) out[p.dst] = g[p.src];
}

// out[i] = sum over ranks (ascending) of g[r][i]: the same bits on every rank
__global__ __launch_bounds__(256) void k_sum_ranks(const double *__restrict__ g, int n_ranks, int64_t n,
                                                   double *__restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    double s = g[i];
    for (int r = 1; r < n_ranks; ++r) s += g[int64_t(r) * n + i];
    out[i] = s;
}

int fill_offsets(const char *what, int n_ranks, const int64_t *h_lens, RankOffsets *ro, int64_t *Tmax) {
    ATL_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxRanks, "%s: 1..%d ranks", what, kMaxRanks);
    ro->off[0] = 0;
    *Tmax = 0;
    for (int r = 0; r < n_ranks; ++r) {
        ATL_REQUIRE(h_lens[r] >= 0, "%s: negative shard length", what);
        ro->off[r + 1] = ro->off[r] + h_lens[r];
        *Tmax = std::max(*Tmax, h_lens[r]);
    }
    return ATL_OK;
}

double comm_timeout_s() {
    if (const char *e = getenv("ATLITE_HIP_COMM_TIMEOUT_S")) {
        const double v = atof(e);
        if (v > 0) return v;
    }
    return 120.0;
}

}  // namespace

// In-process rendezvous of the local transport: n_ranks host threads, one atl_comm each.
struct atl_comm_group {
    int n = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    int attached = 0;
    std::vector<const void *> send;   // rank r's block of the collective in flight
    std::vector<int> device;          // rank r's device
    std::vector<hipEvent_t> ready;    // recorded on rank r's stream: its block is complete
    std::vector<hipEvent_t> done;     // recorded on rank r's stream: it has read every peer's block

    // all ranks arrive or nobody leaves with ATL_OK: a missing peer (it failed before the collective) is a
    // time-out and an error on every waiting rank, never a hang
    int barrier(const char *what) {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) {
            set_error("%s: the communicator group was aborted", what);
            return ATL_E_HIP;
        }
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return ATL_OK;
        }
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(comm_timeout_s());
        while (generation == gen && !aborted) {
            if (cv.wait_until(lk, deadline) == std::cv_status::timeout && generation == gen) {
                aborted = true;
                cv.notify_all();
                set_error("%s: a rank did not reach the collective within %.0f s (ATLITE_HIP_COMM_TIMEOUT_S)", what,
                          comm_timeout_s());
                return ATL_E_HIP;
            }
        }
        if (generation == gen) {
            set_error("%s: the communicator group was aborted", what);
            return ATL_E_HIP;
        }
        return ATL_OK;
    }
};

constexpr int kTickets = 8;  // completion events of the asynchronous collectives, reused round robin

struct atl_comm {
    atl_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;        // RCCL transport
    atl_comm_group *group = nullptr;  // local transport
    int n_ranks = 1, rank = 0;
    // asynchronous collectives (atl_allgather_time_v_async): a stream of the communicator's own, so that a step's
    // all-gather and placement run behind the NEXT step's kernel on the context's stream
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr;          // "the context's stream has come this far"
    hipEvent_t ring[kTickets] = {};      // ticket t is complete: ring[t % kTickets]
    int64_t tickets = 0;
    void *buf = nullptr;                 // send / receive staging of the asynchronous collectives
    size_t buf_bytes = 0;
    bool aborted = false;                // atl_comm_abort ran: only atl_comm_destroy is left
};

namespace {

// an error inside a collective of the local transport: wake the peers now instead of letting them wait for the
// rendezvous to time out
int local_fail(atl_comm_group *g, int rc) {
    std::lock_guard<std::mutex> lk(g->m);
    g->aborted = true;
    g->cv.notify_all();
    return rc;
}

#define ATL_LOCAL_TRY(g, expr)                                                             \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return local_fail(g, ATL_E_HIP);                                               \
        }                                                                                  \
    } while (0)

// recv[q] = rank q's `count` doubles, for every q, on stream `s` of this rank (local transport)
int local_all_gather(atl_comm *c, const double *send, double *recv, size_t count, hipStream_t s, const char *what) {
    atl_comm_group *g = c->group;
    atl_ctx *ctx = c->ctx;
    const int r = c->rank;
    g->send[size_t(r)] = send;
    ATL_LOCAL_TRY(g, hipEventRecord(g->ready[size_t(r)], s));
    int rc = g->barrier(what);  // every block is published and its event recorded
    if (rc) return rc;
    for (int k = 0; k < g->n; ++k) {
        const int q = (r + k) % g->n;  // start with the own block, then the peers in a rotated order: no two ranks
                                       // read the same peer first
        if (q != r) ATL_LOCAL_TRY(g, hipStreamWaitEvent(s, g->ready[size_t(q)], 0));
        if (g->device[size_t(q)] == ctx->device)
            ATL_LOCAL_TRY(g, hipMemcpyAsync(recv + size_t(q) * count, g->send[size_t(q)], count * sizeof(double),
                                            hipMemcpyDeviceToDevice, s));
        else
            ATL_LOCAL_TRY(g, hipMemcpyPeerAsync(recv + size_t(q) * count, ctx->device, g->send[size_t(q)],
                                                g->device[size_t(q)], count * sizeof(double), s));
    }
    ATL_LOCAL_TRY(g, hipEventRecord(g->done[size_t(r)], s));
    rc = g->barrier(what);  // every rank's reads are enqueued and marked
    if (rc) return rc;
    // whatever this stream does next (e.g. overwrite the block it sent) waits until every peer has read it
    for (int q = 0; q < g->n; ++q)
        if (q != r) ATL_LOCAL_TRY(g, hipStreamWaitEvent(s, g->done[size_t(q)], 0));
    return ATL_OK;
}

int all_gather(atl_comm *c, const double *send, double *recv, size_t count, hipStream_t s, const char *what) {
    if (c->group) return local_all_gather(c, send, recv, count, s, what);
    ATL_NCCL_TRY(g_rccl.AllGather(send, recv, count, ncclDouble, c->comm, s));
    return ATL_OK;
}

int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

// pack (ragged shards) -> all-gather -> placement on stream `s`; `scr` holds (n_ranks + 1) * N * Tmax doubles
int gather_place(atl_comm *comm, const double *d_local, int64_t N, const RankOffsets &ro, int64_t Tmax, double *scr,
                 double *d_out, int64_t ld_out, hipStream_t s, const char *what) {
    const int64_t T_r = ro.off[comm->rank + 1] - ro.off[comm->rank];
    // every rank sends a full-width block (a rank whose shard has the full width sends its block as it is)
    const bool pack = T_r < Tmax;
    double *send = scr, *recv = scr + N * Tmax;
    if (pack) {
        ATL_HIP_TRY(hipMemsetAsync(send, 0, size_t(N * Tmax) * sizeof(double), s));
        if (N * T_r > 0)
            ATL_HIP_TRY(hipMemcpy2DAsync(send, size_t(Tmax) * sizeof(double), d_local, size_t(T_r) * sizeof(double),
                                         size_t(T_r) * sizeof(double), size_t(N), hipMemcpyDeviceToDevice, s));
    }
    int rc = all_gather(comm, pack ? send : d_local, recv, size_t(N * Tmax), s, what);
    if (rc) return rc;
    const dim3 grid(unsigned((Tmax + 255) / 256), unsigned(N), unsigned(comm->n_ranks));
    hipLaunchKernelGGL(k_gather_place_v, grid, dim3(256), 0, s, recv, ro, N, Tmax, d_out, ld_out);
    return launched(what);
}

int check_gather_args(const char *what, atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens,
                      double *d_out, int64_t ld_out, RankOffsets *ro, int64_t *Tmax) {
    ATL_REQUIRE(comm && h_lens && d_out, "%s: bad argument", what);
    ATL_REQUIRE(!comm->aborted, "%s: the communicator was aborted", what);
    ATL_REQUIRE(N >= 0 && N < 65536, "%s: bad shape", what);
    int rc = fill_offsets(what, comm->n_ranks, h_lens, ro, Tmax);
    if (rc) return rc;
    ATL_REQUIRE(ld_out >= ro->off[comm->n_ranks], "%s: ld_out too small", what);
    ATL_REQUIRE(d_local || N * h_lens[comm->rank] == 0, "%s: d_local is NULL", what);
    return ATL_OK;
}

// the communicator's own stream and events, created on first use
int async_setup(atl_comm *c) {
    if (c->stream) return ATL_OK;
    ATL_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ATL_HIP_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (int i = 0; i < kTickets; ++i) ATL_HIP_TRY(hipEventCreateWithFlags(&c->ring[i], hipEventDisableTiming));
    return ATL_OK;
}

// ncclCommInitRank is a rendezvous: a peer that never arrives would block the caller for good.  It runs on a helper
// thread; the caller gives up after $ATLITE_HIP_COMM_TIMEOUT_S (120 s) and reports an error (the helper thread is
// left behind - the process is expected to end after such a failure)
int init_rank_with_timeout(int device, int n_ranks, int rank, const ncclUniqueId &id, ncclComm_t *out) {
    struct Job {
        std::promise<ncclResult_t> done;
        ncclComm_t comm = nullptr;
    };
    auto job = std::make_shared<Job>();
    std::future<ncclResult_t> fut = job->done.get_future();
    std::thread([job, device, n_ranks, rank, id]() {
        ncclResult_t r = ncclSuccess;
        if (hipSetDevice(device) != hipSuccess) r = ncclUnhandledCudaError;
        if (r == ncclSuccess) r = g_rccl.CommInitRank(&job->comm, n_ranks, id, rank);
        job->done.set_value(r);
    }).detach();
    if (fut.wait_for(std::chrono::duration<double>(comm_timeout_s())) != std::future_status::ready) {
        set_error("atl_comm_init: rank %d of %d: the other ranks did not join the RCCL communicator within %.0f s "
                  "(ATLITE_HIP_COMM_TIMEOUT_S)", rank, n_ranks, comm_timeout_s());
        return ATL_E_HIP;
    }
    const ncclResult_t r = fut.get();
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank (rank %d of %d, device %d) failed: %s", rank, n_ranks, device, g_rccl.GetErrorString(r));
        return ATL_E_HIP;
    }
    *out = job->comm;
    return ATL_OK;
}

}  // namespace

extern "C" {

int atl_comm_unique_id(void *h_id128) {
    ATL_REQUIRE(h_id128, "atl_comm_unique_id: id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == ATL_COMM_ID_BYTES, "unique id size");
    ATL_NCCL_TRY(g_rccl.GetUniqueId(static_cast<ncclUniqueId *>(h_id128)));
    return ATL_OK;
}

int atl_comm_init(atl_ctx *ctx, int n_ranks, int rank, const void *h_id128, atl_comm **out) {
    ATL_REQUIRE(ctx && h_id128 && out, "atl_comm_init: bad argument");
    *out = nullptr;
    ATL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "atl_comm_init: rank %d of %d", rank, n_ranks);
    int rc = load_rccl();
    if (rc) return rc;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ncclComm_t c = nullptr;
    rc = init_rank_with_timeout(ctx->device, n_ranks, rank, id, &c);
    if (rc) return rc;
    atl_comm *a = new atl_comm();
    a->ctx = ctx;
    a->comm = c;
    a->n_ranks = n_ranks;
    a->rank = rank;
    *out = a;
    return ATL_OK;
}

int atl_comm_init_all(atl_ctx *const *ctxs, int n_ranks, atl_comm **out) {
    ATL_REQUIRE(ctxs && out, "atl_comm_init_all: bad argument");
    ATL_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxRanks, "atl_comm_init_all: 1..%d ranks", kMaxRanks);
    for (int r = 0; r < n_ranks; ++r) {
        out[r] = nullptr;
        ATL_REQUIRE(ctxs[r], "atl_comm_init_all: context %d is NULL", r);
        for (int q = 0; q < r; ++q)
            ATL_REQUIRE(ctxs[q]->device != ctxs[r]->device, "atl_comm_init_all: device %d appears twice (RCCL cannot put one "
                        "GPU into a communicator twice; use atl_comm_init_local)", ctxs[r]->device);
    }
    int rc = load_rccl();
    if (rc) return rc;
    ATL_REQUIRE(g_rccl.GroupStart && g_rccl.GroupEnd, "atl_comm_init_all: librccl lacks ncclGroupStart / ncclGroupEnd");
    ncclUniqueId id;
    ATL_NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::vector<ncclComm_t> comms(size_t(n_ranks), nullptr);
    // one thread, one group: the N rendezvous calls cannot wait for each other (N threads calling ncclCommInitRank
    // each on its own is the pattern that hangs)
    ATL_NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int r = 0; r < n_ranks && bad == ncclSuccess; ++r) {
        if (hipSetDevice(ctxs[r]->device) != hipSuccess) bad = ncclUnhandledCudaError;
        if (bad == ncclSuccess) bad = g_rccl.CommInitRank(&comms[size_t(r)], n_ranks, id, r);
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
        set_error("atl_comm_init_all: RCCL communicator of %d devices failed: %s", n_ranks,
                  g_rccl.GetErrorString(bad != ncclSuccess ? bad : end));
        for (ncclComm_t c : comms)
            if (c) (void)g_rccl.CommDestroy(c);
        return ATL_E_HIP;
    }
    for (int r = 0; r < n_ranks; ++r) {
        atl_comm *a = new atl_comm();
        a->ctx = ctxs[r];
        a->comm = comms[size_t(r)];
        a->n_ranks = n_ranks;
        a->rank = r;
        out[r] = a;
    }
    return ATL_OK;
}

int atl_comm_info(atl_comm *comm, int *n_ranks, int *rank, int *device, int *transport) {
    ATL_REQUIRE(comm, "atl_comm_info: comm is NULL");
    int n = comm->n_ranks, dev = comm->ctx->device;
    if (comm->comm) {  // what RCCL itself says
        if (g_rccl.CommCount) ATL_NCCL_TRY(g_rccl.CommCount(comm->comm, &n));
        if (g_rccl.CommCuDevice) ATL_NCCL_TRY(g_rccl.CommCuDevice(comm->comm, &dev));
    }
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = comm->rank;
    if (device) *device = dev;
    if (transport) *transport = comm->comm ? ATL_COMM_RCCL : ATL_COMM_LOCAL;
    return ATL_OK;
}

int atl_comm_group_create(int n_ranks, atl_comm_group **out) {
    ATL_REQUIRE(out, "atl_comm_group_create: out is NULL");
    *out = nullptr;
    ATL_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxRanks, "atl_comm_group_create: 1..%d ranks", kMaxRanks);
    atl_comm_group *g = new atl_comm_group();
    g->n = n_ranks;
    g->send.assign(size_t(n_ranks), nullptr);
    g->device.assign(size_t(n_ranks), -1);
    g->ready.assign(size_t(n_ranks), nullptr);
    g->done.assign(size_t(n_ranks), nullptr);
    *out = g;
    return ATL_OK;
}

int atl_comm_group_destroy(atl_comm_group *g) {
    if (!g) return ATL_OK;
    ATL_REQUIRE(g->attached == 0, "atl_comm_group_destroy: %d communicator(s) still attached", g->attached);
    delete g;
    return ATL_OK;
}

int atl_comm_init_local(atl_ctx *ctx, atl_comm_group *g, int rank, atl_comm **out) {
    ATL_REQUIRE(ctx && g && out, "atl_comm_init_local: bad argument");
    *out = nullptr;
    ATL_REQUIRE(rank >= 0 && rank < g->n, "atl_comm_init_local: rank %d of %d", rank, g->n);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t ready = nullptr, done = nullptr;
    ATL_HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    ATL_HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->ready[size_t(rank)]) {
            (void)hipEventDestroy(ready);
            (void)hipEventDestroy(done);
            set_error("atl_comm_init_local: rank %d is already attached", rank);
            return ATL_E_INVALID;
        }
        g->ready[size_t(rank)] = ready;
        g->done[size_t(rank)] = done;
        g->device[size_t(rank)] = ctx->device;
        ++g->attached;
    }
    atl_comm *a = new atl_comm();
    a->ctx = ctx;
    a->group = g;
    a->n_ranks = g->n;
    a->rank = rank;
    // every rank attached: peer access towards the other devices (direct xGMI reads; without it the runtime stages
    // the copy, still correct)
    int rc = g->barrier("atl_comm_init_local");
    if (rc) {
        (void)atl_comm_destroy(a);
        return rc;
    }
    for (int q = 0; q < g->n; ++q) {
        const int d = g->device[size_t(q)];
        int can = 0;
        if (d != ctx->device && hipDeviceCanAccessPeer(&can, ctx->device, d) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(d, 0);
            if (e != hipSuccess) (void)hipGetLastError();  // already enabled: fine
        }
    }
    *out = a;
    return ATL_OK;
}

int atl_comm_abort(atl_comm *comm) {
    if (!comm) return ATL_OK;
    if (comm->group) {
        std::lock_guard<std::mutex> lk(comm->group->m);
        comm->group->aborted = true;
        comm->group->cv.notify_all();
    }
    // RCCL: a rank that never enqueues its collective leaves the others blocked on the device; ncclCommAbort ends this
    // rank's outstanding operations and frees its side of the communicator (atl_comm_destroy then finds nothing to destroy)
    comm->aborted = true;
    if (comm->comm && g_rccl.CommAbort) {
        (void)hipSetDevice(comm->ctx->device);
        (void)g_rccl.CommAbort(comm->comm);
        comm->comm = nullptr;
    }
    return ATL_OK;
}

int atl_comm_destroy(atl_comm *comm) {
    if (!comm) return ATL_OK;
    (void)hipSetDevice(comm->ctx->device);
    if (comm->stream) (void)hipStreamSynchronize(comm->stream);
    if (comm->comm) {
        (void)hipStreamSynchronize(comm->ctx->stream);
        (void)g_rccl.CommDestroy(comm->comm);
    }
    if (comm->group) {
        (void)hipStreamSynchronize(comm->ctx->stream);
        atl_comm_group *g = comm->group;
        std::lock_guard<std::mutex> lk(g->m);
        if (g->ready[size_t(comm->rank)]) (void)hipEventDestroy(g->ready[size_t(comm->rank)]);
        if (g->done[size_t(comm->rank)]) (void)hipEventDestroy(g->done[size_t(comm->rank)]);
        g->ready[size_t(comm->rank)] = g->done[size_t(comm->rank)] = nullptr;
        --g->attached;
    }
    if (comm->stream) {
        (void)hipEventDestroy(comm->ev_in);
        for (int i = 0; i < kTickets; ++i) (void)hipEventDestroy(comm->ring[i]);
        (void)hipStreamDestroy(comm->stream);
    }
    if (comm->buf) (void)dev_free(comm->buf);
    delete comm;
    return ATL_OK;
}

int atl_allgather_time(atl_comm *comm, const double *d_local, int64_t N, int64_t T_r, double *d_out,
                       int64_t ld_out) {
    ATL_REQUIRE(comm, "atl_allgather_time: comm is NULL");
    ATL_REQUIRE(T_r >= 0 && comm->n_ranks <= kMaxRanks, "atl_allgather_time: bad shape");
    int64_t lens[kMaxRanks];
    for (int r = 0; r < comm->n_ranks; ++r) lens[r] = T_r;
    return atl_allgather_time_v(comm, d_local, N, lens, d_out, ld_out);  // equal shards: nothing to pad
}

int atl_allgather_time_v(atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens, double *d_out,
                         int64_t ld_out) {
    RankOffsets ro;
    int64_t Tmax = 0;
    int rc = check_gather_args("atl_allgather_time_v", comm, d_local, N, h_lens, d_out, ld_out, &ro, &Tmax);
    if (rc) return rc;
    atl_ctx *ctx = comm->ctx;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (N * Tmax == 0) return ATL_OK;
    // scratch = [send: N x Tmax][recv: n_ranks x N x Tmax]
    void *scr = nullptr;
    rc = scratch_reserve(ctx, size_t(comm->n_ranks + 1) * size_t(N * Tmax) * sizeof(double), &scr);
    if (rc) return comm->group ? local_fail(comm->group, rc) : rc;
    return gather_place(comm, d_local, N, ro, Tmax, static_cast<double *>(scr), d_out, ld_out, ctx->stream,
                        "atl_allgather_time_v");
}

int atl_allgather_time_v_async(atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens, double *d_out,
                               int64_t ld_out, int64_t *ticket) {
    RankOffsets ro;
    int64_t Tmax = 0;
    int rc = check_gather_args("atl_allgather_time_v_async", comm, d_local, N, h_lens, d_out, ld_out, &ro, &Tmax);
    if (rc) return rc;
    ATL_REQUIRE(ticket, "atl_allgather_time_v_async: ticket is NULL");
    atl_ctx *ctx = comm->ctx;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = async_setup(comm))) return comm->group ? local_fail(comm->group, rc) : rc;
    // the staging buffer belongs to the communicator (the context's scratch arena is reused by the next kernel on the
    // context's stream, which this collective is meant to overlap)
    const size_t need = size_t(comm->n_ranks + 1) * size_t(std::max<int64_t>(N * Tmax, 1)) * sizeof(double);
    if (need > comm->buf_bytes) {
        ATL_HIP_TRY(hipStreamSynchronize(comm->stream));
        if (comm->buf) ATL_HIP_TRY(dev_free(comm->buf));
        comm->buf = nullptr;
        comm->buf_bytes = 0;
        hipError_t e = dev_malloc(&comm->buf, need);
        if (e != hipSuccess) {
            set_error("atl_allgather_time_v_async: hipMalloc of %zu bytes failed: %s", need, hipGetErrorString(e));
            return comm->group ? local_fail(comm->group, ATL_E_NOMEM) : ATL_E_NOMEM;
        }
        comm->buf_bytes = need;
    }
    // ordered after everything enqueued on the context's stream so far (the kernel that produced d_local)
    ATL_HIP_TRY(hipEventRecord(comm->ev_in, ctx->stream));
    ATL_HIP_TRY(hipStreamWaitEvent(comm->stream, comm->ev_in, 0));
    if (N * Tmax > 0) {
        rc = gather_place(comm, d_local, N, ro, Tmax, static_cast<double *>(comm->buf), d_out, ld_out, comm->stream,
                          "atl_allgather_time_v_async");
        if (rc) return rc;
    }
    const int64_t t = comm->tickets++;
    ATL_HIP_TRY(hipEventRecord(comm->ring[t % kTickets], comm->stream));
    *ticket = t;
    return ATL_OK;
}

int atl_comm_wait(atl_comm *comm, int64_t ticket) {
    ATL_REQUIRE(comm, "atl_comm_wait: comm is NULL");
    ATL_REQUIRE(ticket >= 0 && ticket < comm->tickets, "atl_comm_wait: ticket %lld was never issued", (long long)ticket);
    // the communicator's stream is in order: a ring slot reused by a later ticket marks a later point of it
    ATL_HIP_TRY(hipStreamWaitEvent(comm->ctx->stream, comm->ring[ticket % kTickets], 0));
    return ATL_OK;
}

int atl_comm_sync(atl_comm *comm) {
    ATL_REQUIRE(comm, "atl_comm_sync: comm is NULL");
    ATL_HIP_TRY(hipSetDevice(comm->ctx->device));
    if (comm->stream) ATL_HIP_TRY(hipStreamSynchronize(comm->stream));
    return ATL_OK;
}

int atl_gather_place_v_host(const double *h_gathered, int n_ranks, int64_t N, const int64_t *h_lens, double *h_out,
                            int64_t ld_out) {
    ATL_REQUIRE(h_gathered && h_lens && h_out && N >= 0, "atl_gather_place_v_host: bad argument");
    RankOffsets ro;
    int64_t Tmax = 0;
    int rc = fill_offsets("atl_gather_place_v_host", n_ranks, h_lens, &ro, &Tmax);
    if (rc) return rc;
    ATL_REQUIRE(ld_out >= ro.off[n_ranks], "atl_gather_place_v_host: ld_out too small");
    // the kernel's grid, walked on the host: (ceil(Tmax / 256) x 256 threads, N, n_ranks)
    const int64_t gx = (Tmax + 255) / 256;
    for (int r = 0; r < n_ranks; ++r)
        for (int64_t n = 0; n < N; ++n)
            for (int64_t t = 0; t < gx * 256; ++t) {
                const Placement p = gather_placement(ro, r, n, t, N, Tmax, ld_out);
                if (p.live) h_out[p.dst] = h_gathered[p.src];
            }
    return ATL_OK;
}

int atl_allreduce_sum(atl_comm *comm, double *d_buf, int64_t n) {
    ATL_REQUIRE(comm && (n == 0 || d_buf) && n >= 0, "atl_allreduce_sum: bad argument");
    ATL_REQUIRE(!comm->aborted, "atl_allreduce_sum: the communicator was aborted");
    if (n == 0) return ATL_OK;
    atl_ctx *ctx = comm->ctx;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (!comm->group) {
        ATL_NCCL_TRY(g_rccl.AllReduce(d_buf, d_buf, size_t(n), ncclDouble, ncclSum, comm->comm, ctx->stream));
        return ATL_OK;
    }
    // local transport: gather every rank's vector, then add them in rank order (after the gather's closing waits
    // nobody still reads d_buf, so the sum may go back in place)
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, size_t(comm->n_ranks) * size_t(n) * sizeof(double), &scr);
    if (rc) return local_fail(comm->group, rc);
    rc = local_all_gather(comm, d_buf, static_cast<double *>(scr), size_t(n), ctx->stream, "atl_allreduce_sum");
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_ranks, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const double *>(scr), comm->n_ranks, n, d_buf);
    return launched("atl_allreduce_sum");
}

}  // extern "C"
