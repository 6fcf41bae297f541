// Issue-rate probe for v_mfma_f32_32x32x2_f32 on gfx950: cycles per MFMA for (V0) register operands only,
// (V1) A operands re-read from LDS every k-step, (V2) + B operands streamed from global memory, (V3) the same with the
// operand loads issued one unrolled block (8 k-steps) ahead.  One or two 256-thread workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ w, float *out, unsigned long long *cyc, int iters, int span) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 257 * 65; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) acc[m][n][v] = 0.0f;
    const float *a0 = lds + ((tid >> 6) * 64 + (lane & 31)) * 65 + (lane >> 5);
    const float *a1 = a0 + 32 * 65;
    const float *wp = w + lane * 2;
    float ra0 = a0[0], ra1 = a1[0], rb0 = wp[0], rb1 = wp[1];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 3) {            // one 16-byte load per lane feeds TWO k-steps
#pragma unroll
            for (int s = 0; s < 32; s += 2) {
                const float4 t = *reinterpret_cast<const float4 *>(w + ((size_t)((it % span) * 16 + s / 2) * 64 + lane) * 4);
                const float x0 = a0[2 * s], x1 = a1[2 * s], x2 = a0[2 * s + 2], x3 = a1[2 * s + 2];
                acc[0][0] = MF(x0, t.x, acc[0][0]);
                acc[0][1] = MF(x0, t.y, acc[0][1]);
                acc[1][0] = MF(x1, t.x, acc[1][0]);
                acc[1][1] = MF(x1, t.y, acc[1][1]);
                acc[0][0] = MF(x2, t.z, acc[0][0]);
                acc[0][1] = MF(x2, t.w, acc[0][1]);
                acc[1][0] = MF(x3, t.z, acc[1][0]);
                acc[1][1] = MF(x3, t.w, acc[1][1]);
            }
            continue;
        }
        if (V == 5) {            // 4 row tiles x 1 column tile per wave: one 16-byte load per lane feeds FOUR k-steps = 16 MFMAs
            const float *a2 = a0 + 64 * 65 > lds + 224 * 65 ? a0 : a0, *a3 = a1;
#pragma unroll
            for (int s = 0; s < 32; s += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(w + ((size_t)((it % span) * 8 + s / 4) * 64 + lane) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float bk = k == 0 ? t.x : k == 1 ? t.y : k == 2 ? t.z : t.w;
                    acc[0][0] = MF(a0[2 * (s + k)], bk, acc[0][0]);
                    acc[0][1] = MF(a1[2 * (s + k)], bk, acc[0][1]);
                    acc[1][0] = MF(a2[2 * (s + k) + 1], bk, acc[1][0]);
                    acc[1][1] = MF(a3[2 * (s + k) + 1], bk, acc[1][1]);
                }
            }
            continue;
        }
        if (V == 4) {            // B from LDS too (pre-staged): no vector-memory instruction in the loop
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float2 t = *reinterpret_cast<const float2 *>(lds + 8192 + (s * 64 + lane) * 2);
                const float x0 = a0[2 * s], x1 = a1[2 * s];
                acc[0][0] = MF(x0, t.x, acc[0][0]);
                acc[0][1] = MF(x0, t.y, acc[0][1]);
                acc[1][0] = MF(x1, t.x, acc[1][0]);
                acc[1][1] = MF(x1, t.y, acc[1][1]);
            }
            continue;
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            float x0 = ra0, x1 = ra1, y0 = rb0, y1 = rb1;
            if (V >= 1) { x0 = a0[2 * s]; x1 = a1[2 * s]; }
            if (V >= 2) { const float2 t = *reinterpret_cast<const float2 *>(wp + (size_t)((it % span) * 32 + s) * 128); y0 = t.x; y1 = t.y; }
            acc[0][0] = MF(x0, y0, acc[0][0]);
            acc[0][1] = MF(x0, y1, acc[0][1]);
            acc[1][0] = MF(x1, y0, acc[1][0]);
            acc[1][1] = MF(x1, y1, acc[1][1]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) sum += acc[m][n][v];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char *name, int wg_per_cu, const float *w, float *out, unsigned long long *cyc, int span = 8) {
    const int grid = 256 * wg_per_cu, iters = 72;
    const size_t lds = 257 * 65 * 4 + (wg_per_cu == 1 ? 70000 : 0);   // force one workgroup per CU by LDS size
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 200; ++rep) probe<V><<<grid, 256, lds>>>(w, out, cyc, iters, span);   // warm the clock
    hipDeviceSynchronize();
    probe<V><<<grid, 256, lds>>>(w, out, cyc, iters, span);
    hipDeviceSynchronize();
    unsigned long long h[512];
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < grid; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
    const double mf = (double)iters * 32 * 4;
    printf("%-32s %d WG/CU: %.1f cycles per MFMA per wave (mean), %.1f (slowest)  -> pipe use %.0f %%\n", name, wg_per_cu,
           s / grid / mf, mx / mf, 100.0 * 64.0 * wg_per_cu / (s / grid / mf));
}

int main() {
    float *w, *out; unsigned long long *cyc;
    hipMalloc(&w, 72 * 32 * 128 * 4 + 4096); hipMemset(w, 0, 72 * 32 * 128 * 4 + 4096);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    for (int k = 1; k <= 2; ++k) {
        run<0>("registers only", k, w, out, cyc);
        run<1>("A from LDS", k, w, out, cyc);
        run<2>("A LDS, B global 128 KB loop", k, w, out, cyc, 8);
        run<2>("A LDS, B global 1.2 MB stream", k, w, out, cyc, 72);
        run<3>("A LDS, B global 16 B / 2 steps", k, w, out, cyc, 72);
        run<4>("A LDS, B LDS", k, w, out, cyc, 72);
        run<5>("4 A tiles, B 16 B / 4 steps", k, w, out, cyc, 72);
    }
    return 0;
}
