// Probe of v_mfma_f64_16x16x4_f64 operand / result layouts on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_layout.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Hypothesis checked: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[4 (l / 16) + r][l % 16] in
// register r of lane l - or D[4 r + l / 16][l % 16].  Prints the mapping found against a host product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A is 16 x 4 row-major
    const double b = B[(l / 16) * 16 + l % 16];  // B is 4 x 16 row-major
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];  // raw: register r of lane l
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i + 0.001 * i * i; hB[i] = 2.0 - 0.11 * i + 0.0007 * i * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // discover the mapping: which (i, j) does register r of lane l hold?
    int bad = 0, hyp = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            int fi = -1, fj = -1;
            for (int i = 0; i < 16 && fi < 0; ++i)
                for (int j = 0; j < 16; ++j)
                    if (fabs(hD[l * 4 + r] - ref[i * 16 + j]) <= 1e-12 * fabs(ref[i * 16 + j])) { fi = i; fj = j; break; }
            if (fi < 0) ++bad;
            if (fi == 4 * r + l / 16 && fj == l % 16) ++hyp;
            if (l % 16 == 3 || l < 2) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, fi, fj);
        }
    printf("unmatched %d; D[4 r + l / 16][l %% 16] holds for %d of 256\n", bad, hyp);
    return bad != 0;
}
