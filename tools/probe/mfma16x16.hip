// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950 (the Gomoku tower's 16-row tiles):
//   A (16 x 4): lane l supplies A[l & 15][l >> 4];  B (4 x 16): lane l supplies B[l >> 4][l & 15];
//   D (16 x 16): register j of lane l holds D[4 * (l >> 4) + j][l & 15].
// Also timed: back-to-back issue rate with 10 independent accumulators (cycles per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o mfma16x16 mfma16x16.hip && ./mfma16x16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *A, const float *B, float *D) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = acc[j];
}
__global__ void k_rate(int n, unsigned long long *out, float *sink) {
    f32x4 a[10];
    for (int i = 0; i < 10; ++i) a[i] = (f32x4){0, 0, 0, 0};
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 40)
#pragma unroll
        for (int k = 0; k < 40; ++k) a[k % 10] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[k % 10], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 10; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 1234.5f) sink[0] = s;
}
// dependent chains: NACC accumulators used round-robin (the Gomoku tower: the MFMA two back wrote this one's C operand)
template <int NACC>
__global__ void k_dep(int n, unsigned long long *out, float *sink) {
    f32x4 a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = (f32x4){0, 0, 0, 0};
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 40)
#pragma unroll
        for (int k = 0; k < 40; ++k) a[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[k % NACC], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 1234.5f) sink[0] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_f16 (the split-fp16 kernel): NACC accumulators round-robin
template <int NACC>
__global__ void k_dep16(int n, unsigned long long *out, float *sink) {
    f32x16 a[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) a[i][v] = 0;
    h8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (_Float16)(threadIdx.x * 1e-3f); y[k] = (_Float16)1.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 24)
#pragma unroll
        for (int k = 0; k < 24; ++k) a[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a[k % NACC], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += a[i][v];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 1234.5f) sink[0] = s;
}
__global__ void k_rate32(int n, unsigned long long *out, float *sink) {
    f32x16 a[4];
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) a[i][v] = 0;
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 20)
#pragma unroll
        for (int k = 0; k < 20; ++k) a[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[k & 3], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += a[i][v];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 1234.5f) sink[0] = s;
}
int main() {
    float hA[64], hB[64], hD[256], *dA, *dB, *dD; unsigned long long *dt, ht;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11 - 5); hB[i] = (float)((i * 5) % 13 - 6); }
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024); hipMalloc(&dt, 8);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float r = 0;
            for (int k = 0; k < 4; ++k) r += hA[m * 4 + k] * hB[k * 16 + n];
            bad += r != hD[m * 16 + n];
        }
    printf("layout: %d mismatches of 256\n", bad);
    const int N = 96000;
    k_rate<<<1, 64>>>(N, dt, dD); k_rate<<<1, 64>>>(N, dt, dD);
    hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
    printf("rate: %.2f ticks per v_mfma_f32_16x16x4_f32 (the 64-cycle 32x32x2 form measures 72 of these ticks)\n", (double)ht / N);
#define DEP(NA) k_dep<NA><<<1, 64>>>(N, dt, dD); k_dep<NA><<<1, 64>>>(N, dt, dD); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); \
    printf("dependent chains, %d accumulators round-robin: %.2f ticks per MFMA\n", NA, (double)ht / N);
    DEP(1) DEP(2) DEP(4) DEP(5)
#define DEP16(NA) k_dep16<NA><<<1, 64>>>(N, dt, dD); k_dep16<NA><<<1, 64>>>(N, dt, dD); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); \
    printf("v_mfma_f32_32x32x16_f16, %d accumulators round-robin: %.2f ticks per MFMA\n", NA, (double)ht / N);
    DEP16(1) DEP16(2) DEP16(3) DEP16(4)
    k_rate32<<<1, 64>>>(N, dt, dD); k_rate32<<<1, 64>>>(N, dt, dD);
    hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
    printf("rate: %.2f ticks per v_mfma_f32_32x32x2_f32 (20 per loop iteration, 4 accumulators)\n", (double)ht / N);
    return 0;
}
