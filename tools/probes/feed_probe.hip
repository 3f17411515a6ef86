// Can a RUNNING kernel whose waves fill every wave slot be fed by the host?  (round 6: the fed k_inflate launch, atl_ingest.hip)
// A spinner kernel - one wave per workgroup, LDS sized so that 32 workgroups share a CU, more workgroups than the device holds -
// waits for a flag in device memory (s_sleep between polls, 3 s time-out).  The host then tries the ways it has to set that flag
// and to move bulk data while the spinner occupies the machine, and reports for each whether (and when) it took effect:
//   small H2D DMA (4 bytes, page-locked source) on a normal / a high-priority stream, a 64 MiB H2D DMA, hipStreamWriteValue32,
//   a blocking hipMemcpy, and a flag in page-locked HOST memory that the kernel polls across PCIe.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/feed_probe tools/probes/feed_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);         \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

__global__ __launch_bounds__(64) void spinner(const volatile uint32_t *flag, unsigned long long timeout_ticks, uint32_t *n_timeout,
                                              uint32_t *n_ok) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;  // (keeps the allocation)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    for (;;) {
        const uint32_t v = __hip_atomic_load((const uint32_t *)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v) {
            ok = true;
            break;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
        for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(127);
    }
    if (threadIdx.x == 0) atomicAdd(ok ? n_ok : n_timeout, 1u);
}

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
    const unsigned lds_bytes = argc > 1 ? unsigned(atoi(argv[1])) : 5120u;  // 32 workgroups per CU
    const unsigned grid = argc > 2 ? unsigned(atoi(argv[2])) : 10240u;
    CHECK(hipSetDevice(0));
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sk, sn, sp;
    CHECK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, greatest));
    uint32_t *d_flag, *d_cnt, *h_one, *h_cnt, *h_flag;
    uint8_t *d_big, *h_big;
    const size_t big = size_t(64) << 20;
    CHECK(hipMalloc(&d_flag, 256));
    CHECK(hipMalloc(&d_cnt, 256));
    CHECK(hipMalloc(&d_big, big));
    CHECK(hipHostMalloc(&h_one, 256, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_cnt, 256, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_flag, 256, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_big, big, hipHostMallocDefault));
    memset(h_big, 1, big);
    h_one[0] = 1;
    const unsigned long long timeout = 300000000ull;  // 3 s of the 100 MHz clock
    printf("spinner: %u workgroups of one wave, %u B of LDS each; stream priorities %d .. %d\n", grid, lds_bytes, least, greatest);
    const char *names[] = {"4-byte H2D DMA, normal-priority stream", "4-byte H2D DMA, high-priority stream", "64 MiB H2D DMA then 4-byte DMA (high priority)",
                           "hipStreamWriteValue32 (high-priority stream)", "blocking hipMemcpy of 4 bytes", "flag in page-locked HOST memory, written by the CPU",
                           "4-byte H2D DMA, high priority, spinner grid small enough to leave slots free"};
    for (int test = 0; test < 7; ++test) {
        CHECK(hipMemset(d_flag, 0, 256));
        CHECK(hipMemset(d_cnt, 0, 256));
        h_flag[0] = 0;
        CHECK(hipDeviceSynchronize());
        const uint32_t *flag = test == 5 ? h_flag : d_flag;
        const unsigned g = test == 6 ? 4096u : grid;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spinner, dim3(g), dim3(64), lds_bytes, sk, flag, timeout, d_cnt, d_cnt + 1);
        CHECK(hipGetLastError());
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const auto t1 = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        double t_op = 0;
        switch (test) {
            case 0: e = hipMemcpyAsync(d_flag, h_one, 4, hipMemcpyHostToDevice, sn); if (e == hipSuccess) e = hipStreamSynchronize(sn); break;
            case 1: case 6: e = hipMemcpyAsync(d_flag, h_one, 4, hipMemcpyHostToDevice, sp); if (e == hipSuccess) e = hipStreamSynchronize(sp); break;
            case 2:
                e = hipMemcpyAsync(d_big, h_big, big, hipMemcpyHostToDevice, sp);
                if (e == hipSuccess) e = hipStreamSynchronize(sp);
                printf("    (64 MiB DMA done after %.1f ms)\n", ms_since(t1));
                if (e == hipSuccess) e = hipMemcpyAsync(d_flag, h_one, 4, hipMemcpyHostToDevice, sp);
                if (e == hipSuccess) e = hipStreamSynchronize(sp);
                break;
            case 3: e = hipStreamWriteValue32(sp, d_flag, 1, 0); if (e == hipSuccess) e = hipStreamSynchronize(sp); break;
            case 4: e = hipMemcpy(d_flag, h_one, 4, hipMemcpyHostToDevice); break;
            case 5: __atomic_store_n(h_flag, 1u, __ATOMIC_RELEASE); break;
        }
        t_op = ms_since(t1);
        if (e != hipSuccess) {
            printf("    operation failed: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
        }
        CHECK(hipStreamSynchronize(sk));
        const double t_all = ms_since(t0);
        CHECK(hipMemcpy(h_cnt, d_cnt, 8, hipMemcpyDeviceToHost));
        printf("%-78s op returned after %8.1f ms, kernel done after %8.1f ms: %u workgroups saw the flag, %u timed out\n", names[test], t_op, t_all,
               h_cnt[1], h_cnt[0]);
        fflush(stdout);
    }
    return 0;
}
