// Probe (gfx950): what a 2 : 1 read / write stream can reach on this chip - the ceiling of the per-cell series kernels
// (wind: 16 B read + 8 B written per cell-step).  c[i] = a[i] + b[i] over three 11.2 GB cubes with the library's own
// access shapes: 16-byte nontemporal loads and stores per lane, in several work decompositions:
//   A  one element pair per thread, one pass (grid = n / 2 / 256 blocks)
//   B  the series kernel's shape: a thread walks SLOTS slots of its cell pair (stride S between them), G loads in flight
//   C  like B with plain (temporal) stores;  D  like B with sc1 stores
// and, for reference, the read-only (acc += a + b) and the 1 : 1 copy (c = a) streams of the same shapes.
//   hipcc --offload-arch=gfx950 -O3 -o mix_probe mix_probe.hip && ./mix_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_flat(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c, size_t n2) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n2) return;
    const d2 x = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(a) + i);
    const d2 y = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(b) + i);
    __builtin_nontemporal_store(x + y, reinterpret_cast<d2 *>(c) + i);
}

// MODE 0: c = a + b (nt stores)  1: plain stores  2: read only  3: copy c = a  4: sc1 stores
template <int SLOTS, int G, int MODE>
__global__ __launch_bounds__(256) void k_slots(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c, size_t S2, size_t T) {
    const size_t cell = size_t(blockIdx.x) * 256 + threadIdx.x;  // pair index inside a slot
    if (cell >= S2) return;
    const size_t t0 = size_t(blockIdx.y) * SLOTS;
    d2 acc = {0.0, 0.0};
    for (size_t t = t0; t < t0 + SLOTS && t < T; t += G) {
        d2 x[G], y[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t tt = t + g < T ? t + g : T - 1;
            x[g] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(a) + tt * S2 + cell);
            if (MODE != 3) y[g] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(b) + tt * S2 + cell);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (t + g >= T) break;
            d2 *dst = reinterpret_cast<d2 *>(c) + (t + g) * S2 + cell;
            const d2 r = MODE == 3 ? x[g] : x[g] + y[g];
            if (MODE == 0 || MODE == 3) __builtin_nontemporal_store(r, dst);
            else if (MODE == 1) *dst = r;
            else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(r) : "memory");
            else acc += r;
        }
    }
    if (MODE == 2 && acc.x == 1.2345e300) c[cell] = acc.x;
}

// E: flat order, a block walks K consecutive 512-pair... (256 threads x 2 cells) chunks of the SAME slot, G in flight (k_cells_series_flat's shape)
template <int K, int G>
__global__ __launch_bounds__(256) void k_flat_loop(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c, size_t n2) {
    const size_t base = size_t(blockIdx.x) * K * 256 + threadIdx.x;
    for (int k = 0; k < K; k += G) {
        d2 x[G], y[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t i = base + size_t(k + g) * 256;
            const size_t j = i < n2 ? i : n2 - 1;
            x[g] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(a) + j);
            y[g] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(b) + j);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t i = base + size_t(k + g) * 256;
            if (i < n2) __builtin_nontemporal_store(x[g] + y[g], reinterpret_cast<d2 *>(c) + i);
        }
    }
}

template <class F>
float timed(F &&launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 6; ++i) launch();  // clocks up
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t T = 8760, S = 160000, n = T * S, S2 = S / 2;
    double *a, *b, *c;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess || hipMalloc(&c, n * 8) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    hipMemset(a, 0, n * 8);
    hipMemset(b, 0, n * 8);
    hipDeviceSynchronize();
    const double gb3 = 3.0 * n * 8 / 1e9, gb2 = 2.0 * n * 8 / 1e9;
    const int reps = 8;
    auto report = [&](const char *name, float ms, double gb) { printf("%-58s %7.3f ms  %6.0f GB/s  %5.1f %% of 8 TB/s\n", name, ms, gb / ms * 1e3, gb / ms * 1e3 / 80.0); fflush(stdout); };
    report("A  flat, one pair per thread, c = a + b (nt)", timed([&] { k_flat<<<unsigned((n / 2 + 255) / 256), 256>>>(a, b, c, n / 2); }, reps), gb3);
#define RUNE(K, G, name) report(name, timed([&] { k_flat_loop<K, G><<<unsigned((n / 2 + size_t(K) * 256 - 1) / (size_t(K) * 256)), 256>>>(a, b, c, n / 2); }, reps), gb3)
    RUNE(1, 1, "E  flat, 1 chunk per block (= A)");
    RUNE(2, 2, "E  flat, 2 chunks per block, 2 in flight");
    RUNE(4, 4, "E  flat, 4 chunks per block, 4 in flight");
    RUNE(8, 4, "E  flat, 8 chunks per block, 4 in flight");
    RUNE(8, 1, "E  flat, 8 chunks per block, 1 in flight");
    RUNE(8, 8, "E  flat, 8 chunks per block, 8 in flight");
    RUNE(32, 4, "E  flat, 32 chunks per block, 4 in flight");
    const dim3 bx(unsigned((S2 + 255) / 256));
#define RUN(SL, G, MODE, name, gb) report(name, timed([&] { k_slots<SL, G, MODE><<<dim3(bx.x, unsigned((T + SL - 1) / SL)), 256>>>(a, b, c, S2, T); }, reps), gb)
    RUN(32, 4, 0, "B  32 slots per block, 4 in flight, c = a + b (nt stores)", gb3);
    RUN(32, 8, 0, "B  32 slots per block, 8 in flight", gb3);
    RUN(64, 4, 0, "B  64 slots per block, 4 in flight", gb3);
    RUN(8, 4, 0, "B   8 slots per block, 4 in flight", gb3);
    RUN(8760, 4, 0, "B  every slot by one block (long walks), 4 in flight", gb3);
    RUN(32, 4, 1, "C  32 / 4, plain stores", gb3);
    RUN(32, 4, 4, "D  32 / 4, sc1 stores", gb3);
    RUN(32, 4, 2, "   32 / 4, read only (2 cubes)", gb2);
    RUN(32, 4, 3, "   32 / 4, copy c = a (1 : 1)", gb2);
    return 0;
}
