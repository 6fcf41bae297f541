// Does v_mfma_f32_16x16x4_f32 round like two v_mfma_f32_32x32x2_f32 in a row?  (Needed for a one-board Connect-4 tower on
// 16-row tiles that must return the SAME BITS as the 32-row-tile variants: the engine mixes variants by batch size.)
// Both are fed the same K = 64 dot products in the same channel order c_0, c_1, ...: the 32x32x2 form takes (c_2s, c_2s+1)
// in step s (lane half h supplies k = h), the 16x16x4 form (c_4j .. c_4j+3) in step j (lane quarter q supplies k = q).
// If every step is a chain of fp32 fused multiply-adds in k order, the results are bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_order mfma_order.hip && ./mfma_order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;

// A[m][k] (m < 32), B[k][n] (n < 32); D32[m][n] via 32x32x2, D16 for m, n < 16 via 16x16x4
__global__ void k_both(const float *A, const float *B, float *D32, float *D16) {
    const int l = threadIdx.x;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < K / 2; ++s)          // A operand: lane l -> A[l & 31][2 s + (l >> 5)]; B: B[2 s + (l >> 5)][l & 31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + 2 * s + (l >> 5)], B[(2 * s + (l >> 5)) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D32[((v & 3) + 8 * (v >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[v];
    f32x4 a4 = {0, 0, 0, 0};
    for (int j = 0; j < K / 4; ++j)          // A: A[l & 15][4 j + (l >> 4)]; B: B[4 j + (l >> 4)][l & 15]
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * K + 4 * j + (l >> 4)], B[(4 * j + (l >> 4)) * 32 + (l & 15)], a4, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D16[(4 * (l >> 4) + j) * 16 + (l & 15)] = a4[j];
}

int main() {
    float *hA = (float *)malloc(32 * K * 4), *hB = (float *)malloc(K * 32 * 4), hD32[1024], hD16[256];
    float *dA, *dB, *d32, *d16;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&d32, 4096); hipMalloc(&d16, 1024);
    long bad = 0, badfma = 0, total = 0;
    srand(1);
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 32 * K; ++i) {
            hA[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
            hB[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
        }
        hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 32 * 4, hipMemcpyHostToDevice);
        k_both<<<1, 64>>>(dA, dB, d32, d16);
        hipMemcpy(hD32, d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(hD16, d16, 1024, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                ++total;
                if (memcmp(&hD32[m * 32 + n], &hD16[m * 16 + n], 4)) ++bad;
                float r = 0.0f;
                for (int k = 0; k < K; ++k) r = fmaf(hA[m * K + k], hB[k * 32 + n], r);
                if (memcmp(&r, &hD16[m * 16 + n], 4)) ++badfma;
            }
    }
    printf("16x16x4 vs 32x32x2 over the same K order: %ld of %ld results differ\n", bad, total);
    printf("16x16x4 vs a host chain of fmaf in k order: %ld of %ld differ\n", badfma, total);
    return 0;
}
