// Cycles per ds_read_b128 / ds_write_b64 / ds_write_b128 of ONE wave for the activation-fragment address patterns of the
// inference kernels: lane l touches byte (l & 31) * row_bytes + (l >> 5) * half_bytes (+ an optional row permutation).
// A conflict-free 64-lane 16-byte access takes 8 cycles (128 bytes per clock).
//   hipcc --offload-arch=gfx950 -O3 -o lds_conflict lds_conflict.hip && ./lds_conflict
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>      // 0 ds_read_b128, 1 ds_write_b64, 2 ds_write_b128, 3 ds_read_b64
__global__ void k(int n, int row_bytes, int half_bytes, int lanes16, unsigned long long *out, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 40960; i += 64) reinterpret_cast<float *>(lds)[i] = i;
    __syncthreads();
    const int l = threadIdx.x;
    // lanes16: lane = row (l & 15), quarter (l >> 4) (the 16-row tiles); else row (l & 31), half (l >> 5)
    const uint32_t addr = lanes16 ? (l & 15) * row_bytes + (l >> 4) * half_bytes : (l & 31) * row_bytes + (l >> 5) * half_bytes;
    float4 acc = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            if (MODE == 0) { v4f t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr)); asm volatile("" :: "v"(t)); }
            else if (MODE == 3) { v2f t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr)); asm volatile("" :: "v"(t)); }
            else if (MODE == 1) { v2f t = {1, 2}; asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(t) : "memory"); }
            else { v4f t = {1, 2, 3, 4}; asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(t) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) out[0] = t1 - t0;
    if (acc.x == 123.0f) sink[0] = acc.x;
}

int main() {
    unsigned long long *d, h; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 64);
    const int N = 80000;
    struct Case { const char *what; int mode, row, half, l16; } cases[] = {
        {"read_b128  fp32 64ch  row 272 B, half +16", 0, 272, 16, 0},
        {"read_b128  fp32 128ch row 528 B, half +16", 0, 528, 16, 0},
        {"read_b128  fp32 256ch row 1040 B, quarter +16 (16-row tiles)", 0, 1040, 16, 1},
        {"read_b128  f16 64ch   row 144 B, half +16", 0, 144, 16, 0},
        {"read_b128  f16 128ch  row 272 B, half +16", 0, 272, 16, 0},
        {"read_b128  f16 64ch   row 160 B, half +16", 0, 160, 16, 0},
        {"read_b128  f16 64ch   row 176 B, half +16", 0, 176, 16, 0},
        {"read_b128  f16 64ch   row 208 B, half +16", 0, 208, 16, 0},
        {"read_b128  contiguous 16 B per lane", 0, 16, 512, 0},
        {"write_b64  f16 64ch   row 144 B, half +8", 1, 144, 8, 0},
        {"write_b64  f16 64ch   row 176 B, half +8", 1, 176, 8, 0},
        {"write_b64  f16 64ch   row 208 B, half +8", 1, 208, 8, 0},
        {"write_b128 fp32 64ch  row 272 B, half +16", 2, 272, 16, 0},
        {"write_b128 fp32 256ch row 1040 B, quarter +16 (16-row tiles)", 2, 1040, 16, 1},
    };
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            if (c.mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 163840, 0, N, c.row, c.half, c.l16, d, sink);
            else if (c.mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 163840, 0, N, c.row, c.half, c.l16, d, sink);
            else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 163840, 0, N, c.row, c.half, c.l16, d, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-66s %6.2f cycles per access\n", c.what, (double)h / N);
    }
    return 0;
}
