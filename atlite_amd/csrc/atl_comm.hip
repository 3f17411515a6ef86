// Multi-GPU collectives of the hot path over RCCL (xGMI), for hosts that do not go through
// torch.distributed.  One process per GPU; the time axis is sharded, so the only
// exchanges are an all-gather of the small (shapes x T_r) result blocks and an all-reduce of
// time sums.  librccl.so.1 is opened lazily: the library loads (and every single-GPU entry point
// works) without it.  The reference has no distributed path to mirror.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
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

// gathered [rank][n][t] -> out[n][rank * T_r + t]
__global__ __launch_bounds__(256) void k_gather_place(const double *__restrict__ g, int n_ranks, int64_t N,
                                                      int64_t T_r, double *__restrict__ out, int64_t ld_out) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    const int r = blockIdx.z;
    if (t < T_r) out[n * ld_out + int64_t(r) * T_r + t] = g[(int64_t(r) * N + n) * T_r + t];
}

// ragged: gathered [rank][n][Tmax] (rank r uses its first len_r columns) -> out[n][off_r + t]
constexpr int kMaxRanks = 64;
struct RankOffsets {
    int64_t off[kMaxRanks + 1];
};
__global__ __launch_bounds__(256) void k_gather_place_v(const double *__restrict__ g, RankOffsets ro, int64_t N,
                                                        int64_t Tmax, double *__restrict__ out, int64_t ld_out) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    const int r = blockIdx.z;
    if (t < ro.off[r + 1] - ro.off[r]) out[n * ld_out + ro.off[r] + t] = g[(int64_t(r) * N + n) * Tmax + t];
}

}  // namespace

struct atl_comm {
    atl_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
};

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
    ATL_NCCL_TRY(g_rccl.CommInitRank(&c, n_ranks, id, rank));
    atl_comm *a = new atl_comm();
    a->ctx = ctx;
    a->comm = c;
    a->n_ranks = n_ranks;
    a->rank = rank;
    *out = a;
    return ATL_OK;
}

int atl_comm_destroy(atl_comm *comm) {
    if (!comm) return ATL_OK;
    if (comm->comm) {
        (void)hipStreamSynchronize(comm->ctx->stream);
        (void)g_rccl.CommDestroy(comm->comm);
    }
    delete comm;
    return ATL_OK;
}

int atl_allgather_time(atl_comm *comm, const double *d_local, int64_t N, int64_t T_r, double *d_out,
                       int64_t ld_out) {
    ATL_REQUIRE(comm && d_local && d_out, "atl_allgather_time: bad argument");
    ATL_REQUIRE(N >= 0 && T_r >= 0 && N < 65536, "atl_allgather_time: bad shape");
    ATL_REQUIRE(ld_out >= int64_t(comm->n_ranks) * T_r, "atl_allgather_time: ld_out too small");
    atl_ctx *ctx = comm->ctx;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (N * T_r == 0) return ATL_OK;
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, size_t(comm->n_ranks) * size_t(N * T_r) * sizeof(double), &scr);
    if (rc) return rc;
    ATL_NCCL_TRY(g_rccl.AllGather(d_local, scr, size_t(N * T_r), ncclDouble, comm->comm, ctx->stream));
    const dim3 grid(unsigned((T_r + 255) / 256), unsigned(N), unsigned(comm->n_ranks));
    hipLaunchKernelGGL(k_gather_place, grid, dim3(256), 0, ctx->stream, static_cast<const double *>(scr),
                       comm->n_ranks, N, T_r, d_out, ld_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("atl_allgather_time: kernel launch failed: %s", hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

int atl_allgather_time_v(atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens, double *d_out,
                         int64_t ld_out) {
    ATL_REQUIRE(comm && h_lens && d_out, "atl_allgather_time_v: bad argument");
    ATL_REQUIRE(comm->n_ranks <= kMaxRanks, "atl_allgather_time_v: at most %d ranks", kMaxRanks);
    ATL_REQUIRE(N >= 0 && N < 65536, "atl_allgather_time_v: bad shape");
    RankOffsets ro;
    int64_t Tmax = 0;
    ro.off[0] = 0;
    for (int r = 0; r < comm->n_ranks; ++r) {
        ATL_REQUIRE(h_lens[r] >= 0, "atl_allgather_time_v: negative shard length");
        ro.off[r + 1] = ro.off[r] + h_lens[r];
        Tmax = std::max(Tmax, h_lens[r]);
    }
    const int64_t T_r = h_lens[comm->rank];
    ATL_REQUIRE(ld_out >= ro.off[comm->n_ranks], "atl_allgather_time_v: ld_out too small");
    ATL_REQUIRE(d_local || N * T_r == 0, "atl_allgather_time_v: d_local is NULL");
    atl_ctx *ctx = comm->ctx;
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    if (N * Tmax == 0) return ATL_OK;
    // scratch = [send: N x Tmax][recv: n_ranks x N x Tmax]; every rank sends a full-width block
    void *scr = nullptr;
    int rc = scratch_reserve(ctx, size_t(comm->n_ranks + 1) * size_t(N * Tmax) * sizeof(double), &scr);
    if (rc) return rc;
    double *send = static_cast<double *>(scr), *recv = send + N * Tmax;
    if (T_r < Tmax) ATL_HIP_TRY(hipMemsetAsync(send, 0, size_t(N * Tmax) * sizeof(double), ctx->stream));
    if (N * T_r > 0)
        ATL_HIP_TRY(hipMemcpy2DAsync(send, size_t(Tmax) * sizeof(double), d_local, size_t(T_r) * sizeof(double),
                                     size_t(T_r) * sizeof(double), size_t(N), hipMemcpyDeviceToDevice, ctx->stream));
    ATL_NCCL_TRY(g_rccl.AllGather(send, recv, size_t(N * Tmax), ncclDouble, comm->comm, ctx->stream));
    const dim3 grid(unsigned((Tmax + 255) / 256), unsigned(N), unsigned(comm->n_ranks));
    hipLaunchKernelGGL(k_gather_place_v, grid, dim3(256), 0, ctx->stream, recv, ro, N, Tmax, d_out, ld_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("atl_allgather_time_v: kernel launch failed: %s", hipGetErrorString(e));
        return ATL_E_HIP;
    }
    return ATL_OK;
}

int atl_allreduce_sum(atl_comm *comm, double *d_buf, int64_t n) {
    ATL_REQUIRE(comm && (n == 0 || d_buf) && n >= 0, "atl_allreduce_sum: bad argument");
    if (n == 0) return ATL_OK;
    ATL_HIP_TRY(hipSetDevice(comm->ctx->device));
    ATL_NCCL_TRY(g_rccl.AllReduce(d_buf, d_buf, size_t(n), ncclDouble, ncclSum, comm->comm, comm->ctx->stream));
    return ATL_OK;
}

}  // extern "C"
