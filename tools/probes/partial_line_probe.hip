// Probe (gfx950): does the memory system fetch whole 128-byte lines whatever part of a line a wave asks for?
// A large buffer is read ONCE with nontemporal 16-byte loads - the fused kernel's load - but only lanes whose
// (lane & 7) is below K take part: K = 8, 4, 2, 1 asks for 128, 64, 32, 16 bytes of every 128-byte line, as a tile
// row of the fused kernel does when the plan's coverage mask switches the other lanes off (ragged polygon edges).
// Run under  rocprofv3 --pmc FETCH_SIZE  /  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum  (separate passes): if
// the fetched bytes stay at the full buffer for K < 8, HBM fills are whole lines and the 1.46x traffic of BASELINE's
// star polygons over their covered cells is line granularity, irreducible by masking.
//   hipcc --offload-arch=gfx950 -O3 -o partial_line_probe partial_line_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256) void k_partial(const double *__restrict__ in, size_t n_lines, double *__restrict__ out) {
    const size_t tid = size_t(blockIdx.x) * 256 + threadIdx.x;
    const size_t line = tid >> 3;
    const int sub = int(tid & 7);
    double acc = 0.0;
    // 8 lines per thread, a chip-wide stride apart (every line of the buffer is touched exactly once)
    const size_t stride = size_t(gridDim.x) * 32;
    for (int r = 0; r < 8; ++r) {
        const size_t l = line + size_t(r) * stride;
        if (sub < K && l < n_lines) {
            const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(in + l * 16 + size_t(sub) * 2));
            acc += v.x + v.y;
        }
    }
    if (acc == 1.2345e300) out[tid] = acc;
}

template <int K>
float run(const double *in, size_t n_lines, double *out, int reps) {
    const unsigned grid = unsigned((n_lines / 8 * 8 + 255) / 256);  // 8 lanes per line, 8 lines per thread
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k_partial<K><<<grid, 256>>>(in, n_lines, out);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) k_partial<K><<<grid, 256>>>(in, n_lines, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? size_t(atoi(argv[1])) : 8;
    const size_t bytes = gib << 30, n_lines = bytes / 128;
    double *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    hipMemset(in, 0, bytes);
    hipDeviceSynchronize();
    const int reps = 5;
    const float t8 = run<8>(in, n_lines, out, reps), t4 = run<4>(in, n_lines, out, reps), t2 = run<2>(in, n_lines, out, reps),
                t1 = run<1>(in, n_lines, out, reps);
    printf("buffer %zu GiB = %zu lines of 128 B, every line touched once per launch, %d timed launches each\n", gib, n_lines, reps);
    const float ts[4] = {t8, t4, t2, t1};
    const int ks[4] = {8, 4, 2, 1};
    for (int i = 0; i < 4; ++i)
        printf("K=%d  asks for %3d B of every line  %.3f ms  %.0f GB/s on the bytes asked for  %.0f GB/s if whole lines move\n", ks[i],
               16 * ks[i], ts[i], double(n_lines) * 16 * ks[i] / (ts[i] * 1e-3) / 1e9, double(bytes) / (ts[i] * 1e-3) / 1e9);
    return 0;
}
