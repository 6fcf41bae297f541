// Probe for the inner loop of the fp32 residual tower (k_resnet_forward<Connect4, 2, 4, 0, 2, 2>): one 256-thread
// workgroup = 2 x 2 waves, each wave 4 row tiles x 1 column tile of a 64-channel 3x3 convolution (9 taps x 32 k-steps
// of v_mfma_f32_32x32x2_f32), tile-taps skipped as in the shipped kernel (30 of 36 per layer), activations in LDS,
// weights streamed from global memory (L2 resident, 1.2 MB).  The shipped loop reaches 82 % of the matrix pipe with one
// AND with two waves per SIMD; the variants below change one thing at a time:
//   V0  shipped form: per tap 8 x 16-byte weight loads up front, then per row tile a block of 32 MFMAs whose A operands
//       come from ds_read2_b32 issued just in time (stride 65 dwords)
//   V1  A operands as ds_read_b128: activations with stride 68, k order permuted so that one 16-byte read holds the four
//       values a lane needs for four consecutive k-steps; the reads of a 16-MFMA half block are issued up front
//   V2  V1 + weights double-buffered by half tap (the next half tap's four loads are issued before this one's MFMAs)
//   V4  V2 with the four row tiles interleaved per quad of k-steps (rotating accumulators) instead of one block per tile
//   V5  one static, role-specialised instruction sequence per layer: A reads two steps ahead, next tap's weights in a second buffer
//   V6  V5 with ONE weight buffer: the tap's last tile block reloads each quad for the next tap as soon as it has used it
// ROWS = 0: LDS rows in board order (conflict-free), ROWS = 1: the shipped tap-skipping row order (bank conflicts).
//   hipcc --offload-arch=gfx950 -O3 -o tower_loop tower_loop.hip && ./tower_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

constexpr int LAYERS = 8, TAPS = 9, S = 32, NQ = 8;   // k-steps per tap, quads per tap
constexpr int WG_ROWS = 256;

struct Params {
    const float *w;          // [layer][tap][quad][col tile 2][lane 64][4]
    const int *perm;         // [256] GEMM row -> LDS row
    float *out;
    unsigned long long *cyc;
};

__device__ __forceinline__ float comp(const float4 &t, int k) { return k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w)); }

// tile-tap masks of the shipped 2 x 2 grid: row group 0 = [left, right, interior, interior], 1 = [top, bottom, interior, interior]
__device__ __forceinline__ unsigned need_of(int group, int tile) {
    if (tile >= 2) return 0x1FFu;
    if (group == 0) return tile == 0 ? (0x1FFu & ~0x049u) : (0x1FFu & ~0x124u);     // lacks dx = -1 : taps 0,3,6 / dx = +1 : 2,5,8
    return tile == 0 ? (0x1FFu & ~0x007u) : (0x1FFu & ~0x1C0u);                     // lacks dy = -1 : taps 0,1,2 / dy = +1 : 6,7,8
}


// ---- V5: role-specialised static schedule -------------------------------------------------------------------------
// The tile-tap masks are a static property of the row order, so the layer body is specialised on the wave's row group
// and written as one straight-line sequence of 240 steps (tile-tap x quad of k-steps).  Step i issues the 16-byte A read
// of step i + 2 (three rotating buffers) and then its own four MFMAs; the eight weight quads of the NEXT tap are loaded
// into a second buffer while the current tap is computed, one load at the head of every fourth step.
// __builtin_amdgcn_sched_barrier(0) between steps keeps the compiler from re-serialising reads and MFMAs.
struct Step { int tap, tile, q, first_of_tap, last_tile; };
template <int GROUP>
struct Sched {
    Step s[240 + 2];
    int n;
    constexpr Sched() : s{}, n(0) {
        for (int tap = 0; tap < 9; ++tap) {
            int first = 1;
            for (int tile = 3; tile >= 0; --tile) {
                unsigned need = tile >= 2 ? 0x1FFu : (GROUP == 0 ? (tile == 0 ? (0x1FFu & ~0x049u) : (0x1FFu & ~0x124u))
                                                                  : (tile == 0 ? (0x1FFu & ~0x007u) : (0x1FFu & ~0x1C0u)));
                if (!((need >> tap) & 1u)) continue;
                for (int q = 0; q < 8; ++q) { s[n].tap = tap; s[n].tile = tile; s[n].q = q; s[n].first_of_tap = first && q == 0; s[n].last_tile = 0; ++n; }
                first = 0;
            }
            for (int q = 0; q < 8; ++q) s[n - 8 + q].last_tile = 1;        // the tap's last tile block
        }
        s[n] = s[n - 1]; s[n + 1] = s[n - 1];       // harmless over-reads past the end
    }
};

template <int GROUP>
struct SchedHolder { static constexpr Sched<GROUP> SC{}; };

template <int GROUP, int STRIDE, int PAR, int I>
__device__ __forceinline__ void step_v5(f32x16 (&acc)[4], const float *lds, const int (&ridx)[4], const float *wl,
                                        float4 (&wb)[2][8], float4 (&xa)[3]) {
    constexpr Step st = SchedHolder<GROUP>::SC.s[I], nx = SchedHolder<GROUP>::SC.s[I + 2];
    constexpr int cur = (st.tap + PAR) & 1;
    constexpr bool wload = SchedHolder<GROUP>::SC.s[I - st.q].first_of_tap != 0;      // quad q of the tap's first tile
    xa[(I + 2) % 3] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[nx.tile] + ((nx.tap / 3 - 1) * 7 + (nx.tap % 3 - 1)) * STRIDE + 8 * nx.q);
    if constexpr (wload)
        wb[cur ^ 1][st.q] = *reinterpret_cast<const float4 *>(wl + (size_t)((st.tap + 1) * 8 + st.q) * 2 * 64 * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[st.tile] = MF(comp(xa[I % 3], k), comp(wb[cur][st.q], k), acc[st.tile]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < 240) step_v5<GROUP, STRIDE, PAR, I + 1>(acc, lds, ridx, wl, wb, xa);
}

template <int GROUP, int STRIDE, int I>
__device__ __forceinline__ void step_v6(f32x16 (&acc)[4], const float *lds, const int (&ridx)[4], const float *wl,
                                        float4 (&wb)[8], float4 (&xa)[3]) {
    constexpr Step st = SchedHolder<GROUP>::SC.s[I], nx = SchedHolder<GROUP>::SC.s[I + 2];
    xa[(I + 2) % 3] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[nx.tile] + ((nx.tap / 3 - 1) * 7 + (nx.tap % 3 - 1)) * STRIDE + 8 * nx.q);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[st.tile] = MF(comp(xa[I % 3], k), comp(wb[st.q], k), acc[st.tile]);
    // the tap's last tile has consumed quad q: the same registers take the next tap's quad q (needed eight steps later)
    if constexpr (st.last_tile != 0)
        wb[st.q] = *reinterpret_cast<const float4 *>(wl + (size_t)((st.tap + 1) * 8 + st.q) * 2 * 64 * 4);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < 240) step_v6<GROUP, STRIDE, I + 1>(acc, lds, ridx, wl, wb, xa);
}

template <int GROUP, int STRIDE>
__device__ __forceinline__ void layer_v6(f32x16 (&acc)[4], const float *lds, const int (&ridx)[4], const float *wl, float4 (&wb)[8]) {
    float4 xa[3];
    constexpr Step s0 = SchedHolder<GROUP>::SC.s[0], s1 = SchedHolder<GROUP>::SC.s[1];
    xa[0] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[s0.tile] + ((s0.tap / 3 - 1) * 7 + (s0.tap % 3 - 1)) * STRIDE + 8 * s0.q);
    xa[1] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[s1.tile] + ((s1.tap / 3 - 1) * 7 + (s1.tap % 3 - 1)) * STRIDE + 8 * s1.q);
    step_v6<GROUP, STRIDE, 0>(acc, lds, ridx, wl, wb, xa);
}

template <int GROUP, int STRIDE, bool XR, int PAR>
__device__ __forceinline__ void layer_v5(f32x16 (&acc)[4], const float *lds, const int (&ridx)[4], const float *wl, float4 (&wb)[2][8]) {
    float4 xa[3];
    constexpr Step s0 = SchedHolder<GROUP>::SC.s[0], s1 = SchedHolder<GROUP>::SC.s[1];
    xa[0] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[s0.tile] + ((s0.tap / 3 - 1) * 7 + (s0.tap % 3 - 1)) * STRIDE + 8 * s0.q);
    xa[1] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[s1.tile] + ((s1.tap / 3 - 1) * 7 + (s1.tap % 3 - 1)) * STRIDE + 8 * s1.q);
    step_v5<GROUP, STRIDE, PAR, 0>(acc, lds, ridx, wl, wb, xa);
}

template <int V, int STRIDE, bool XR = false>
__global__ __launch_bounds__(256, 2) void probe(Params P) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, wave = tid >> 6;
    const int group = wave >> 1, wc = wave & 1;
    for (int i = tid; i < (WG_ROWS + 16) * STRIDE; i += 256) lds[i] = (float)((i * 7) & 15) * 0.0625f - 0.4f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.0f;
    int ridx[4];
    unsigned need[4];
    for (int t = 0; t < 4; ++t) {
        const int r = P.perm[group * 128 + t * 32 + (lane & 31)];
        ridx[t] = r * STRIDE + (V == 0 ? half : 4 * half);
        need[t] = need_of(group, t);
    }
    const float *wbase = P.w + ((size_t)wc * 64 + lane) * 4;
    f32x16 xres[4];                                 // the shipped kernel keeps the residual in 64 registers across a block
    if constexpr (XR) for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) xres[t][v] = lds[(t * 16 + v) * 3 + lane];
    float4 wb[2][8];
    if constexpr (V == 5 || V == 6) {
#pragma unroll
        for (int q = 0; q < 8; ++q) wb[0][q] = *reinterpret_cast<const float4 *>(wbase + (size_t)q * 2 * 64 * 4);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int layer = 0; layer < LAYERS; ++layer) {
        const float *wl = wbase + (size_t)layer * TAPS * NQ * 2 * 64 * 4;
        if constexpr (XR) if (layer & 1) for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) { acc[t][v] += xres[t][v]; xres[t][v] = fmaxf(acc[t][v], 0.0f); }
        if constexpr (V == 6) {
            if (group == 0) layer_v6<0, STRIDE>(acc, lds, ridx, wl, wb[0]); else layer_v6<1, STRIDE>(acc, lds, ridx, wl, wb[0]);
            __syncthreads();
            continue;
        }
        if constexpr (V == 5) {
            // nine taps per layer: the buffer parity of tap 0 alternates from layer to layer
            if (group == 0) { if (layer & 1) layer_v5<0, STRIDE, XR, 1>(acc, lds, ridx, wl, wb); else layer_v5<0, STRIDE, XR, 0>(acc, lds, ridx, wl, wb); }
            else { if (layer & 1) layer_v5<1, STRIDE, XR, 1>(acc, lds, ridx, wl, wb); else layer_v5<1, STRIDE, XR, 0>(acc, lds, ridx, wl, wb); }
            __syncthreads();
            continue;
        }
        if constexpr (V == 0) {
            for (int tap = 0; tap < TAPS; ++tap) {
                const int off = ((tap / 3 - 1) * 7 + (tap % 3 - 1)) * STRIDE;
                const float *wt = wl + (size_t)tap * NQ * 2 * 64 * 4;
                float4 t[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) t[q] = *reinterpret_cast<const float4 *>(wt + (size_t)q * 2 * 64 * 4);
#pragma unroll
                for (int tile = 3; tile >= 0; --tile) {
                    if (!((need[tile] >> tap) & 1u)) continue;
                    const float *a = lds + 8 * STRIDE + ridx[tile] + off;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[tile] = MF(a[2 * (4 * q + k)], comp(t[q], k), acc[tile]);
                }
            }
        } else if constexpr (V == 1) {
            for (int tap = 0; tap < TAPS; ++tap) {
                const int off = ((tap / 3 - 1) * 7 + (tap % 3 - 1)) * STRIDE;
                const float *wt = wl + (size_t)tap * NQ * 2 * 64 * 4;
                float4 t[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) t[q] = *reinterpret_cast<const float4 *>(wt + (size_t)q * 2 * 64 * 4);
#pragma unroll
                for (int tile = 3; tile >= 0; --tile) {
                    if (!((need[tile] >> tap) & 1u)) continue;
                    const float *a = lds + 8 * STRIDE + ridx[tile] + off;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float4 x[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float4 *>(a + 8 * (4 * h + q));
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[tile] = MF(comp(x[q], k), comp(t[4 * h + q], k), acc[tile]);
                    }
                }
            }
        } else {
            // weights double-buffered by half tap: hb[cur] holds the four quads of half-tap i
            float4 hb[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hb[0][q] = *reinterpret_cast<const float4 *>(wl + (size_t)q * 2 * 64 * 4);
#pragma unroll 1
            for (int tap = 0; tap < TAPS; ++tap) {
                const int off = ((tap / 3 - 1) * 7 + (tap % 3 - 1)) * STRIDE;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // next half tap (past the layer's end: the next layer's first, the buffer is padded)
                    const float *wn = wl + (size_t)((tap * 2 + h + 1) * 4) * 2 * 64 * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hb[(h + 1) & 1][q] = *reinterpret_cast<const float4 *>(wn + (size_t)q * 2 * 64 * 4);
                    if constexpr (V == 4) {
                        float4 x[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
#pragma unroll
                            for (int tile = 0; tile < 4; ++tile)
                                x[tile] = *reinterpret_cast<const float4 *>(lds + 8 * STRIDE + ridx[tile] + off + 8 * (4 * h + q));
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int tile = 0; tile < 4; ++tile)
                                    if ((need[tile] >> tap) & 1u) acc[tile] = MF(comp(x[tile], k), comp(hb[h & 1][q], k), acc[tile]);
                        }
                    } else {
#pragma unroll
                        for (int tile = 3; tile >= 0; --tile) {
                            if (!((need[tile] >> tap) & 1u)) continue;
                            const float *a = lds + 8 * STRIDE + ridx[tile] + off + 32 * h;
                            float4 x[4];
                            x[0] = *reinterpret_cast<const float4 *>(a);
#pragma unroll
                            for (int q = 1; q < 4; ++q) x[q] = *reinterpret_cast<const float4 *>(a + 8 * q);
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[tile] = MF(comp(x[q], k), comp(hb[h & 1][q], k), acc[tile]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) sum += acc[t][v];
    P.out[(size_t)blockIdx.x * 256 + tid] = sum;
    if (lane == 0) P.cyc[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
}

static void build_row_order(int *q) {     // the shipped 2 x 2 table (azsp_resnet.hip build_row_order, second half)
    constexpr int R = 6, C = 7, CE = 42, TBB = 6;
    std::vector<int> left, right, top, bottom, interior, rest;
    for (int b = 0; b < TBB; ++b)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x) {
                const int r = b * CE + y * C + x;
                const bool corner_l = x == 0 && (y == 0 || y == R - 1) && b < 2;
                if (x == 0 && !corner_l) left.push_back(r);
                else if (x == 0) (y == 0 ? top : bottom).push_back(r);
                else if (x == C - 1) right.push_back(r);
                else if (y == 0) top.push_back(r);
                else if (y == R - 1) bottom.push_back(r);
                else interior.push_back(r);
            }
    while (right.size() > 32) { rest.push_back(right.back()); right.pop_back(); }
    for (int r = TBB * CE; r < 256; ++r) rest.push_back(r);
    interior.insert(interior.end(), rest.begin(), rest.end());
    const std::vector<int> *special[4] = {&left, &right, &top, &bottom};
    for (int grp = 0; grp < 2; ++grp) {
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 32; ++i) q[grp * 128 + k * 32 + i] = (*special[grp * 2 + k])[i];
        for (int i = 0; i < 64; ++i) q[grp * 128 + 64 + i] = interior[grp * 64 + i];
    }
}

template <int V, int STRIDE, bool XR = false>
static void run(const char *name, int rows_mode, int wg_per_cu, Params P, const int *d_perm_id, const int *d_perm_sk) {
    const int grid = 256 * wg_per_cu;
    P.perm = rows_mode ? d_perm_sk : d_perm_id;
    const size_t lds = (size_t)(WG_ROWS + 16) * STRIDE * 4 + (wg_per_cu == 1 ? 84000 : 0);    // force one workgroup per CU by LDS size
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<V, STRIDE, XR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 300; ++rep) probe<V, STRIDE, XR><<<grid, 256, lds>>>(P);     // warm the clock
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) probe<V, STRIDE, XR><<<grid, 256, lds>>>(P);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), P.cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (auto c : h) { s += c; mx = c > mx ? c : mx; }
    const double mf = (double)LAYERS * 30 * S;                 // MFMAs per wave: 30 tile-taps per layer
    const double ideal_us = mf * 64.0 * wg_per_cu / 2.4e3;     // matrix-pipe time at 2.4 GHz per launch
    printf("%-44s rows=%d %d WG/CU: %6.1f cycles/MFMA/wave (mean) %6.1f (slowest) pipe %5.1f %% | launch %.1f us (%.1f %% of 2.4 GHz pipe time)\n",
           name, rows_mode, wg_per_cu, s / h.size() / mf, mx / mf, 100.0 * 64.0 * wg_per_cu / (s / h.size() / mf),
           ms * 1e3 / 20, 100.0 * ideal_us / (ms * 1e3 / 20));
}

int main() {
    const size_t wn = (size_t)(LAYERS + 1) * TAPS * NQ * 2 * 64 * 4;
    std::vector<float> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.0f - 0.5f;
    float *w, *out; unsigned long long *cyc; int *pid, *psk;
    hipMalloc(&w, wn * 4); hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 4 * 8);
    int id[256], sk[256];
    for (int i = 0; i < 256; ++i) id[i] = i;
    build_row_order(sk);
    hipMalloc(&pid, sizeof id); hipMemcpy(pid, id, sizeof id, hipMemcpyHostToDevice);
    hipMalloc(&psk, sizeof sk); hipMemcpy(psk, sk, sizeof sk, hipMemcpyHostToDevice);
    Params P{w, pid, out, cyc};
    for (int k = 1; k <= 2; ++k)
        for (int rows = 0; rows <= 1; ++rows) {
            run<0, 65>("V0 shipped: ds_read2_b32 just in time", rows, k, P, pid, psk);
            run<1, 68>("V1 ds_read_b128, 4 reads per half block", rows, k, P, pid, psk);
            run<2, 68>("V2 V1 + weights double-buffered per half tap", rows, k, P, pid, psk);
            run<4, 68>("V4 V2 with tiles interleaved per quad", rows, k, P, pid, psk);
            run<0, 65, true>("V0 + 64 residual registers live", rows, k, P, pid, psk);
            run<5, 68>("V5 static schedule, A and B pipelined", rows, k, P, pid, psk);
            run<5, 68, true>("V5 + 64 residual registers live", rows, k, P, pid, psk);
            run<6, 68>("V6 static schedule, one weight buffer refreshed in place", rows, k, P, pid, psk);
            run<6, 68, true>("V6 + 64 residual registers live", rows, k, P, pid, psk);
        }
    return 0;
}
