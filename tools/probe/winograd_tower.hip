// Probe for VERDICT r02 item 5: would a Winograd F(2x2, 3x3) tower on v_mfma_f32_16x16x4_f32 beat the direct,
// tap-skipping tower of k_resnet_forward<Connect4, 2, 4, 0, 2, 2> (64 channels, 6 boards per workgroup) by >= 15 %?
//
// Work of one convolution layer for the 6 boards of a workgroup, per SIMD (= per wave, 4 waves):
//   direct (shipped)  960 v_mfma_f32_32x32x2_f32 (30 of 36 tile-taps x 32 k-steps), 64 pipe cycles each = 61 440 pipe cycles;
//                     measured 66.5 cycles per MFMA in the statically scheduled form (tools/probe/tower_loop.hip V6),
//                     68.3 - 71.0 in the compiler-scheduled forms (V1 / V2)            -> 63 840 cycles per layer
//   Winograd          6 boards x 12 output tiles (2 x 2; a 6 x 7 board is 3 x 4 tiles, one padded column) = 72 tiles = five
//                     16-row tiles (8 padding rows); 16 transform positions x 5 row tiles x 4 column tiles (16 channels) x
//                     16 k-steps = 5120 v_mfma_f32_16x16x4_f32 per workgroup = 1280 per wave, 32 pipe cycles each
//                     = 40 960 pipe cycles (1.5 x fewer than the direct form's 61 440)
// What the MFMA count does not show, and this probe measures:
//   * a wave must hold ALL 16 positions of its (row tile, column tile) pairs to do the output transform in registers
//     (the transformed-domain results of a layer are 327 KB per workgroup: not an LDS job), so it owns one column tile x
//     five row tiles x 16 positions = 80 accumulator tiles = 320 registers (AccVGPRs; nothing left for the residual);
//   * every B operand is a transformed activation V = B^T d B: the wave reads the 4 x 4 patch of its tile (16 ds_read_b128
//     = 4 channels) and spends 32 v_add_f32 per channel on it -- 128 adds per 64 MFMAs, redone by all four waves (each
//     needs every V for its own column tile); beside fp32 MFMAs a vector instruction costs matrix-pipe time one for one
//     (tools/probe/issue_share.hip);
//   * the transformed weights are 16 x 64 x 64 floats = 262 KB per layer instead of 147 KB.
// The kernel below is that inner loop (real transform arithmetic, weights from global memory, activations from LDS,
// compiler-scheduled like tower_loop V1/V2) for ONE layer-equivalent of one workgroup; it prints cycles per wave and layer.
//   hipcc --offload-arch=gfx950 -O3 -o winograd_tower winograd_tower.hip && ./winograd_tower
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

constexpr int C = 64, POS = 16, RT = 5, KQ = C / 16;      // k-quads: a quad = 4 k-steps = 16 input channels (4 per lane quarter)
constexpr int STRIDE = C + 4;                               // activation row stride (floats), 16-byte aligned, odd multiple of 4
constexpr int ROWS = 6 * 8 * 9 + 1;                         // 6 boards, each stored with a one-cell zero border: 8 x 9 cells

__device__ __forceinline__ float comp(const float4 &t, int k) { return k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w)); }

// TRANSFORM = false: the same loop with the raw patch values fed to the MFMAs (no adds): what the adds cost
template <bool TRANSFORM>
__global__ __launch_bounds__(256) void k_wino(const float *__restrict__ U, float *out, unsigned long long *cyc, int layers) {
    extern __shared__ __attribute__((aligned(16))) float act[];          // [ROWS][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq4 = lane >> 4, tl = lane & 15;
    for (int i = tid; i < ROWS * STRIDE; i += 256) act[i] = (float)((i * 7) % 13 - 6) * 0.125f;
    __syncthreads();
    f32x4 acc[POS][RT];
#pragma unroll
    for (int p = 0; p < POS; ++p)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[p][t] = (f32x4){0, 0, 0, 0};
    // top-left cell (with border) of this lane's output tile in row tile t: tile index 16 t + tl -> board, tile row, tile col
    int base[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int tile = 16 * t + tl, b = tile / 12, ty = (tile % 12) / 4, tx = tile % 4;
        base[t] = tile < 72 ? ((b * 8 + 2 * ty) * 9 + 2 * tx) * STRIDE + 4 * kq4 : (ROWS - 1) * STRIDE + 4 * kq4;
    }
    const float *wl = U + ((size_t)wave * 64 + lane) * 4;                 // [layer][kq][pos][col tile 4][lane 64][4 k-steps]
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int layer = 0; layer < layers; ++layer) {
#pragma unroll 1
        for (int q = 0; q < KQ; ++q) {
            float4 w[POS];
#pragma unroll
            for (int p = 0; p < POS; ++p) w[p] = *reinterpret_cast<const float4 *>(wl + ((size_t)(layer * KQ + q) * POS + p) * 4 * 64 * 4);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                float4 d[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const float4 *>(act + base[t] + (i * 9 + j) * STRIDE + 16 * q);
                float4 v[4][4];
                if constexpr (TRANSFORM) {
                    float4 r[4][4];           // r = B^T d : rows  d0 - d2, d1 + d2, d2 - d1, d1 - d3
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r[0][j] = make_float4(d[0][j].x - d[2][j].x, d[0][j].y - d[2][j].y, d[0][j].z - d[2][j].z, d[0][j].w - d[2][j].w);
                        r[1][j] = make_float4(d[1][j].x + d[2][j].x, d[1][j].y + d[2][j].y, d[1][j].z + d[2][j].z, d[1][j].w + d[2][j].w);
                        r[2][j] = make_float4(d[2][j].x - d[1][j].x, d[2][j].y - d[1][j].y, d[2][j].z - d[1][j].z, d[2][j].w - d[1][j].w);
                        r[3][j] = make_float4(d[1][j].x - d[3][j].x, d[1][j].y - d[3][j].y, d[1][j].z - d[3][j].z, d[1][j].w - d[3][j].w);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {     // v = r B : the same along the columns
                        v[i][0] = make_float4(r[i][0].x - r[i][2].x, r[i][0].y - r[i][2].y, r[i][0].z - r[i][2].z, r[i][0].w - r[i][2].w);
                        v[i][1] = make_float4(r[i][1].x + r[i][2].x, r[i][1].y + r[i][2].y, r[i][1].z + r[i][2].z, r[i][1].w + r[i][2].w);
                        v[i][2] = make_float4(r[i][2].x - r[i][1].x, r[i][2].y - r[i][1].y, r[i][2].z - r[i][1].z, r[i][2].w - r[i][1].w);
                        v[i][3] = make_float4(r[i][1].x - r[i][3].x, r[i][1].y - r[i][3].y, r[i][1].z - r[i][3].z, r[i][1].w - r[i][3].w);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[i][j] = d[i][j];
                }
#pragma unroll
                for (int p = 0; p < POS; ++p)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[p][t] = MF16(comp(w[p], k), comp(v[p / 4][p % 4], k), acc[p][t]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int p = 0; p < POS; ++p)
#pragma unroll
        for (int t = 0; t < RT; ++t) s += acc[p][t][0] + acc[p][t][1] + acc[p][t][2] + acc[p][t][3];
    out[tid] = s;
    if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
    const int layers = 8;
    const size_t nU = (size_t)layers * KQ * POS * 4 * 64 * 4;
    std::vector<float> hU(nU);
    for (size_t i = 0; i < nU; ++i) hU[i] = (float)((i * 11) % 17 - 8) * 0.0625f;
    float *dU, *dOut; unsigned long long *dC, hC[4];
    hipMalloc(&dU, nU * 4); hipMalloc(&dOut, 256 * 4); hipMalloc(&dC, 32);
    hipMemcpy(dU, hU.data(), nU * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)ROWS * STRIDE * 4;
    hipFuncSetAttribute((const void *)k_wino<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k_wino<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("LDS %zu bytes (6 bordered boards), weights %zu KB per layer\n", lds, nU * 4 / layers / 1024);
    for (int rep = 0; rep < 3; ++rep) {
        k_wino<true><<<1, 256, lds>>>(dU, dOut, dC, layers);
        hipDeviceSynchronize();
        hipMemcpy(hC, dC, 32, hipMemcpyDeviceToHost);
        const double w = (double)(hC[0] + hC[1] + hC[2] + hC[3]) / 4 / layers;
        k_wino<false><<<1, 256, lds>>>(dU, dOut, dC, layers);
        hipDeviceSynchronize();
        hipMemcpy(hC, dC, 32, hipMemcpyDeviceToHost);
        const double n = (double)(hC[0] + hC[1] + hC[2] + hC[3]) / 4 / layers;
        // s_memtime ticks at 100 MHz x ... on gfx950 it counts shader-clock cycles scaled as in tower_loop: report both per MFMA
        printf("rep %d: Winograd loop %.0f ticks per layer and wave = %.2f per MFMA (1280 MFMAs); without the transform adds %.0f = %.2f per MFMA\n",
               rep, w, w / 1280.0, n, n / 1280.0);
    }
    printf("direct form for comparison (tools/probe/tower_loop.hip, same tick unit): 960 MFMAs x 66.5 .. 71.0 ticks = 63 840 .. 68 160 per layer and wave\n");
    return 0;
}
