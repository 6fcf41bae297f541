// Issue-rate probe for v_mfma_f32_32x32x16_f16 on gfx950: how much do dependent accumulations cost?
//   chain 1 : every MFMA writes a different accumulator than its predecessor (4 accumulators round-robin)
//   chain 3 : three MFMAs in a row on one accumulator, then the next accumulator   (the f16x2 kernel's hi*hi, hi*lo, lo*hi)
//   chain 12: twelve in a row on one accumulator (tile-major order of the tap-skipping f16x2 loop)
// Operands in registers, or A re-read from LDS (two ds_read_b128 per three MFMAs) as in the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int CHAIN, int LDSA>
__global__ __launch_bounds__(256, 2) void probe(float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 257 * 72 * 2; i += 256) reinterpret_cast<_Float16 *>(lds)[i] = (_Float16)((i & 7) * 0.125f);
    __syncthreads();
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
    h8 b0, b1, a0, a1;
    for (int j = 0; j < 8; ++j) { b0[j] = (_Float16)(0.5f + j); b1[j] = (_Float16)(0.25f * j); a0[j] = (_Float16)(1.0f + lane % 3); a1[j] = (_Float16)0.5f; }
    const unsigned char *ar = lds + ((tid >> 6) * 64 + (lane & 31)) * 144 + (lane >> 5) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {          // 16 groups of 3 MFMAs
            h8 x = a0, y = a1;
            if (LDSA) { x = *reinterpret_cast<const h8 *>(ar + (g & 3) * 32 + (g >> 2) * 32 * 144); y = *reinterpret_cast<const h8 *>(ar + 37008 + (g & 3) * 32 + (g >> 2) * 32 * 144); }
            if (CHAIN == 1) {
                acc[(3 * g) & 3] = MF(x, b0, acc[(3 * g) & 3]);
                acc[(3 * g + 1) & 3] = MF(x, b1, acc[(3 * g + 1) & 3]);
                acc[(3 * g + 2) & 3] = MF(y, b0, acc[(3 * g + 2) & 3]);
            } else {
                const int m = CHAIN == 3 ? (g & 3) : (g >> 2);
                acc[m] = MF(x, b0, acc[m]);
                acc[m] = MF(x, b1, acc[m]);
                acc[m] = MF(y, b0, acc[m]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int m = 0; m < 4; ++m) for (int v = 0; v < 16; ++v) sum += acc[m][v];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAIN, int LDSA>
static void run(const char *name, int wg_per_cu, float *out, unsigned long long *cyc) {
    const int grid = 256 * wg_per_cu, iters = 200;
    const size_t lds = 257 * 72 * 2 * 2 + (wg_per_cu == 1 ? 70000 : 0);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<CHAIN, LDSA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 300; ++rep) probe<CHAIN, LDSA><<<grid, 256, lds>>>(out, cyc, iters);
    hipDeviceSynchronize();
    probe<CHAIN, LDSA><<<grid, 256, lds>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[512];
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; ++i) s += h[i];
    const double mf = (double)iters * 48;
    printf("%-34s %d WG/CU: %.1f cycles per MFMA per wave -> pipe use %.0f %%\n", name, wg_per_cu, s / grid / mf, 100.0 * 32.0 * wg_per_cu / (s / grid / mf));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    for (int k = 1; k <= 2; ++k) {
        run<1, 0>("registers, no dependent neighbours", k, out, cyc);
        run<3, 0>("registers, chains of 3", k, out, cyc);
        run<12, 0>("registers, chains of 12", k, out, cyc);
        run<1, 1>("A from LDS, no dependent neighbours", k, out, cyc);
        run<3, 1>("A from LDS, chains of 3", k, out, cyc);
        run<12, 1>("A from LDS, chains of 12", k, out, cyc);
    }
    return 0;
}
