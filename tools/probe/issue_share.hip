// Do vector-ALU / LDS instructions of one wave take issue time away from the fp32 MFMAs of ANOTHER wave on the same SIMD?
// One workgroup of 8 waves on one CU (two per SIMD).  Waves 0-3 run back-to-back v_mfma_f32_32x32x2_f32 (four independent
// accumulators); waves 4-7 run, by mode: nothing / a v_fma_f32 stream / a ds_read_b128 stream / a ds_write_b32 stream /
// MFMAs as well.  Every wave reports its own duration in core clocks (s_memtime) and the SIMD it ran on.
//   hipcc --offload-arch=gfx950 -O3 -o issue_share issue_share.hip && ./issue_share
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_stream(int n, float *sink) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < n; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0.0f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    if (s == 12345.678f) sink[0] = s;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma16_stream(int n, float *sink) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f16x8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (_Float16)(threadIdx.x * 1e-3f); y[k] = (_Float16)1.0f; }
    for (int i = 0; i < n; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
    }
    float s = 0.0f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    if (s == 12345.678f) sink[0] = s;
}
// the same wave interleaves K v_fma between consecutive MFMAs (F16: the fp16 MFMA)
template <int K, bool F16>
__device__ __forceinline__ void mixed_stream(int n, float *sink) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    f16x8 hx, hy;
    for (int k = 0; k < 8; ++k) { hx[k] = (_Float16)x; hy[k] = (_Float16)1.0f; }
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = threadIdx.x + k;
    const float m = 1.0000001f, c = 1e-9f;
#define ONE(acc) \
    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc, 0, 0, 0); \
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0); \
    _Pragma("unroll") for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k & 7]) : "v"(m), "v"(c));
    for (int i = 0; i < n; i += 4) { ONE(a0) ONE(a1) ONE(a2) ONE(a3) }
#undef ONE
    float s = 0.0f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    for (int k = 0; k < 8; ++k) s += r[k];
    if (s == 12345.678f) sink[0] = s;
}
__device__ __forceinline__ void fma_stream(int n, float *sink) {
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = threadIdx.x + k;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(m), "v"(c));
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += r[k];
    if (s == 12345.678f) sink[0] = s;
}
__device__ __forceinline__ void lds_read_stream(int n, const float *lds, float *sink) {
    float4 acc = make_float4(0, 0, 0, 0);
    const float4 *p = reinterpret_cast<const float4 *>(lds) + (threadIdx.x & 63) * 17;       // conflict-free 16-byte reads
    for (int i = 0; i < n; i += 4) {
        float4 t0, t1, t2, t3;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"((uint32_t)(uintptr_t)p));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(t1) : "v"((uint32_t)(uintptr_t)p));
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(t2) : "v"((uint32_t)(uintptr_t)p));
        asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(t3) : "v"((uint32_t)(uintptr_t)p));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc.x += t0.x + t1.x + t2.x + t3.x;
    }
    if (acc.x == 12345.678f) sink[0] = acc.x;
}
__device__ __forceinline__ void lds_write_stream(int n, float *lds) {
    const uint32_t p = (uint32_t)(uintptr_t)(lds + (threadIdx.x & 63) * 69);
    const float v = threadIdx.x;
    for (int i = 0; i < n; i += 4) {
        asm volatile("ds_write_b32 %0, %1" ::"v"(p), "v"(v) : "memory");
        asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(p), "v"(v) : "memory");
        asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(p), "v"(v) : "memory");
        asm volatile("ds_write_b32 %0, %1 offset:12" ::"v"(p), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// role_a / role_b: 0 nothing, 1 MFMA, 2 v_fma, 3 ds_read_b128, 4 ds_write_b32, 5 fp16 MFMA, 10.. MFMA with v_fma in between
__global__ __launch_bounds__(512, 1) void k_share(int role_a, int n_a, int role_b, int n_b, unsigned long long *out, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const int role = w < 4 ? role_a : role_b, n = w < 4 ? n_a : n_b;
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) mfma_stream(n, sink);
    else if (role == 2) fma_stream(n, sink);
    else if (role == 3) lds_read_stream(n, lds, sink);
    else if (role == 4) lds_write_stream(n, lds);
    else if (role == 5) mfma16_stream(n, sink);
    else if (role == 10) mixed_stream<2, false>(n, sink);
    else if (role == 11) mixed_stream<4, false>(n, sink);
    else if (role == 12) mixed_stream<8, false>(n, sink);
    else if (role == 13) mixed_stream<12, false>(n, sink);
    else if (role == 20) mixed_stream<1, true>(n, sink);
    else if (role == 21) mixed_stream<2, true>(n, sink);
    else if (role == 22) mixed_stream<4, true>(n, sink);
    else if (role == 23) mixed_stream<6, true>(n, sink);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { out[w * 2] = t1 - t0; out[w * 2 + 1] = (hwid >> 4) & 3; }
}

static void run(const char *what, int ra, int na, int rb, int nb, unsigned long long *d_out, float *sink) {
    unsigned long long h[16];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_share, dim3(1), dim3(512), 8192 * 4, 0, ra, na, rb, nb, d_out, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s", what);
    for (int w = 0; w < 8; ++w) printf(" %s%llu@%llu", w == 4 ? "| " : "", h[w * 2], h[w * 2 + 1]);
    printf("\n");
}

int main() {
    unsigned long long *d_out; float *sink;
    hipMalloc(&d_out, 16 * 8); hipMalloc(&sink, 64);
    const int NM = 16000;            // MFMAs: 64 cycles each = 1.02 M cycles (s_memtime ticks at 100 MHz: ~42.7 k ticks at 2.4 GHz)
    const int NF = 256000;           // v_fma: 4 cycles each (one pass per 16 lanes) = 1.02 M cycles
    const int NR = 64000;            // ds_read_b128: 64 lanes x 16 B = 8 LDS cycles each at 128 B/clk
    const int NW = 128000;           // ds_write_b32
    printf("durations in s_memtime ticks, wave@SIMD; waves 0-3 | waves 4-7\n");
    run("MFMA alone", 1, NM, 0, 0, d_out, sink);
    run("v_fma alone (waves 4-7)", 0, 0, 2, NF, d_out, sink);
    run("MFMA + v_fma", 1, NM, 2, NF, d_out, sink);
    run("ds_read_b128 alone", 0, 0, 3, NR, d_out, sink);
    run("MFMA + ds_read_b128", 1, NM, 3, NR, d_out, sink);
    run("ds_write_b32 alone", 0, 0, 4, NW, d_out, sink);
    run("MFMA + ds_write_b32", 1, NM, 4, NW, d_out, sink);
    run("MFMA + MFMA", 1, NM, 1, NM, d_out, sink);
    run("v_fma + v_fma", 2, NF, 2, NF, d_out, sink);
    const int NH = 32000;            // fp16 MFMAs: 32 cycles each
    run("MFMA16 alone", 5, NH, 0, 0, d_out, sink);
    run("MFMA16 + v_fma", 5, NH, 2, NF, d_out, sink);
    run("MFMA16 + ds_write_b32", 5, NH, 4, NW, d_out, sink);
    run("MFMA16 + MFMA16", 5, NH, 5, NH, d_out, sink);
    run("one wave: MFMA, 2 v_fma each", 10, NM, 0, 0, d_out, sink);
    run("one wave: MFMA, 4 v_fma each", 11, NM, 0, 0, d_out, sink);
    run("one wave: MFMA, 8 v_fma each", 12, NM, 0, 0, d_out, sink);
    run("one wave: MFMA, 12 v_fma each", 13, NM, 0, 0, d_out, sink);
    run("two waves: MFMA, 8 v_fma each", 12, NM, 12, NM, d_out, sink);
    run("one wave: MFMA16, 1 v_fma each", 20, NH, 0, 0, d_out, sink);
    run("one wave: MFMA16, 2 v_fma each", 21, NH, 0, 0, d_out, sink);
    run("one wave: MFMA16, 4 v_fma each", 22, NH, 0, 0, d_out, sink);
    run("one wave: MFMA16, 6 v_fma each", 23, NH, 0, 0, d_out, sink);
    run("two waves: MFMA16, 4 v_fma each", 22, NH, 22, NH, d_out, sink);
    return 0;
}
