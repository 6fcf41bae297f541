// Batched policy+value inference on gfx950 matrix cores: the whole ResNet tower of
// src/alphazero_simple/resnet.py:30-157 (stem, residual blocks, policy head, value head, masked
// softmax, tanh) fused into ONE kernel, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
//
// Mapping:
//   * a workgroup owns TB boards = TB*cells rows of the implicit GEMM (Connect-4: 6 boards = 252 of 256 rows for large
//     batches, 3 boards up to 768 boards, 1 board up to ~one workgroup per CU) and ALL output channels; its waves form
//     an RW x CW grid of row groups x column groups.  The 64- and 128-channel Connect-4 kernels use RW = 2 and one wave
//     per 32-channel column tile: a wave then owns 4 (or 2, or 1) row tiles x ONE column tile, and the weights of four
//     k-steps arrive in one 16-byte load per lane (a vector-memory instruction in the MFMA stream costs matrix-pipe
//     time); a tap's weights are loaded first, then one straight-line block of MFMAs runs per row tile.
//   * activations never leave the CU: the layer input lives in LDS as act[row][C+1] (row stride C+1 dwords ->
//     conflict-free ds_read_b32 of one channel across 32 rows), the residual copy lives in registers in the MFMA C/D
//     layout, and each conv writes its output back over the same LDS buffer after a barrier.  A 3x3 tap is an LDS row
//     offset; taps that fall off the board read a shared all-zero row (validity mask precomputed per lane).
//   * 6-board workgroups take their GEMM rows in an order that groups the cells by the taps they lack (left / right
//     column, top / bottom row, interior), so that a whole 32-row tile can skip its off-board taps: -16.7 % MFMAs,
//     bit-identical results (build_row_order).
//   * weights (BatchNorm folded on the host at upload) are pre-permuted into B-fragment order; they stream from L2
//     (whole 64-channel tower = 1.2 MB).
//   * heads: policy 1x1 conv on the same MFMA path, value 3x3 conv (3 channels) on v_mfma_f32_4x4x1_16B_f32; their
//     ReLU outputs go to LDS, the two Linear layers are per-board dot products reduced in a fixed order, then masked
//     softmax / tanh.  Every board's result is independent of which tile / row / variant it landed in
//     (batch-invariant, tested bitwise), so leaf order does not influence the search.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "azsp_internal.h"
#include "azsp_resnet_host.h"

namespace azsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


struct NetGeom {
    int rows, cols, cells, A, AP;
    int C, Cp, NT, S;          // channels, padded to 32, column tiles, k-steps per tap (Cp/2)
    int blocks, TB;            // residual blocks, boards per workgroup
};

struct ResNetDev {
    NetGeom g{};
    int game = 0;
    float *d_blob = nullptr;   // all permuted parameters
    size_t blob_floats = 0;
    // offsets (in floats) into d_blob
    size_t o_stem_w, o_stem_b, o_conv_w, o_conv_b, o_pol_w, o_pol_b, o_val_w, o_val_w4, o_val_b, o_fcp_w, o_fcp_b, o_fcv_w, o_fcv_b;
    size_t conv_w_stride, conv_b_stride;
    double flops_per_board = 0;
    ResNet16Dev *f16 = nullptr;   // split-fp16 variant (AZSP_FLAG_F16_SPLIT)
    int32_t *d_perm = nullptr;    // [2][256] row orders of the tap-skipping Connect-4 kernels: 4x1 and 2x2 wave grids
    size_t o_conv_w4 = 0;         // quad-k-step copy of the conv weights (64-channel Connect-4 nets)
    bool has_w4 = false;
};

struct NetParams {
    const float *stem_w, *stem_b, *conv_w, *conv_b, *pol_w, *pol_b, *val_w, *val_w4, *val_b, *fcp_w, *fcp_b, *fcv_w, *fcv_b;
    int conv_w_stride, conv_b_stride, blocks;
    unsigned long long *dbg;      // diagnostic phase timestamps (AZSP_RESNET_TRACE), else nullptr
    const int32_t *perm;          // [256] GEMM row -> board-cell row of the tap-skipping row order (see build_row_order)
    const float *conv_w4;         // conv weights as [tap][s / 4][column tile][lane][4 k-steps] (one 16-byte load per lane
                                  // feeds four k-steps of a wave that owns ONE column tile), else nullptr
};
// phase timestamp of wave 0 of each workgroup, 100 MHz constant clock; compiled only into the TR = true
// diagnostic instantiation (AZSP_RESNET_TRACE), the product kernels carry no stamp code
#define STAMP(k) do { if constexpr (TR) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + (k)] = wall_clock64(); } } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void load_w(const float *p, float (&w)[NT]) {
    if constexpr (NT == 1) { w[0] = p[0]; }
    else if constexpr (NT == 2) { float2 t = *reinterpret_cast<const float2 *>(p); w[0] = t.x; w[1] = t.y; }
    else if constexpr (NT == 3) { w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; }
    else {
#pragma unroll
        for (int i = 0; i < NT; i += 4) { float4 t = *reinterpret_cast<const float4 *>(p + i); w[i] = t.x; w[i + 1] = t.y; w[i + 2] = t.z; w[i + 3] = t.w; }
    }
}

// One 3x3 (TAPS = 9) or 1x1 (TAPS = 1, centre tap) convolution as implicit GEMM over the LDS-resident
// activations.  acc[mt][nt] += A(rows of tile mt, K) * B(K, channels of tile nt).
//   wl   : permuted weights [tap][S][64][NTW]
//   ridx : per row tile, float index of this lane's row (+ lane>>5) in `act`
//   vmask: per row tile, 9-bit tap validity
template <int NTW, int KB, int WST>
__device__ __forceinline__ void load_wblock(const float *p, float (&w)[KB][NTW]) {
#pragma unroll
    for (int i = 0; i < KB; ++i) load_w<NTW>(p + (size_t)i * 64 * WST, w[i]);
}

template <int MT, int NTW, int KB>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[MT][NTW], const float *const (&arow)[MT], int s0,
                                           const float (&w)[KB][NTW]) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        float a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt][2 * (s0 + i)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], w[i][nt], acc[mt][nt], 0, 0, 0);
    }
}

// One tap of a convolution for the row tiles selected by MASK (bit mt set = row tile mt takes part).
template <int MT, int NTW, int S, int WST, int MASK>
__device__ __forceinline__ void tap_mfma(f32x16 (&acc)[MT][NTW], const float *const (&arow)[MT], const float *wt) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float w[NTW];
        load_w<NTW>(wt + (size_t)s * 64 * WST, w);
        float a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if ((MASK >> mt) & 1) a[mt] = arow[mt][2 * s];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((MASK >> mt) & 1) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], w[nt], acc[mt][nt], 0, 0, 0);
            }
    }
}

// One tap for a wave that owns MT row tiles and ONE column tile: the weights of four k-steps come in one 16-byte
// load per lane (layout [s / 4][column tile][lane][4]), i.e. one vector-memory instruction per 4 * MT MFMAs.  A global
// load in the MFMA stream costs matrix-pipe time that a second resident wave does not win back (measured with
// tools/probe/mfma_rate.hip: 8 B per k-step -> 88 % of the pipe with two waves per SIMD, 16 B per four k-steps -> 100 %).
// MASK < 0: `rt` (wave-uniform) says which row tiles take part.
template <int MT, int S, int NT, int MASK>
__device__ __forceinline__ void tap_mfma_quad(f32x16 (&acc)[MT][1], const float *const (&arow)[MT], const float *wq, uint32_t rt) {
    if constexpr (MT == 1) {
        // One row tile per wave (small batches: a workgroup's serial chain sets the launch time): there are registers
        // to spare, so ALL weight loads of the tap are issued before its first MFMA -- one L2 round trip per tap instead
        // of one per load (measured: the compiler's own schedule waits for them one at a time, 590 ns each).
        float4 t[S / 4];
#pragma unroll
        for (int q = 0; q < S / 4; ++q) t[q] = *reinterpret_cast<const float4 *>(wq + (size_t)q * NT * 64 * 4);
#pragma unroll
        for (int q = 0; q < S / 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float b = k == 0 ? t[q].x : (k == 1 ? t[q].y : (k == 2 ? t[q].z : t[q].w));
                if (MASK >= 0 ? (MASK & 1) != 0 : (rt & 1u) != 0)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[0][2 * (4 * q + k)], b, acc[0][0], 0, 0, 0);
            }
        return;
    }
#pragma unroll
    for (int s = 0; s < S; s += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(wq + (size_t)(s / 4) * NT * 64 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float b = k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (MASK >= 0 ? ((MASK >> mt) & 1) != 0 : ((rt >> mt) & 1u) != 0)
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[mt][2 * (s + k)], b, acc[mt][0], 0, 0, 0);
            }
        }
    }
}

// Row tile T of one tap over NQ quads of k-steps whose weights are already in registers (q0 = first quad).
template <int MT, int NQ, int T>
__device__ __forceinline__ void tap_block(f32x16 (&acc)[MT][1], const float *arow_t, const float4 (&t)[NQ], int q0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float b = k == 0 ? t[q].x : (k == 1 ? t[q].y : (k == 2 ? t[q].z : t[q].w));
            acc[T][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow_t[2 * (4 * (q0 + q) + k)], b, acc[T][0], 0, 0, 0);
        }
}

// `need[mt]` (wave-uniform): taps for which at least one row of row tile mt has an on-board input.  The other
// taps multiply an all-zero A tile -- adding exactly 0 -- and are skipped (SK kernels order their rows so that
// whole row tiles share their off-board taps; everywhere else every tap is needed by some row).
template <class G, int MT, int NTW, int TAPS, int S, int PF, int WST = NTW, bool SK = false>
__device__ __forceinline__ void conv_mfma(f32x16 (&acc)[MT][NTW], const float *__restrict__ wl, const float *act,
                                          const int (&ridx)[MT], const uint32_t (&vmask)[MT], int stride, int zero_idx,
                                          int lane, const uint32_t (&need)[MT], const float *__restrict__ wq = nullptr) {
    if constexpr (PF == 0 && NTW == 1 && TAPS == 9 && WST > 1) {
      if (wq != nullptr) {
        // waves that own ONE column tile: wq = this wave's column of the quad-step weights
        for (int tap = 0; tap < 9; ++tap) {
            uint32_t rt = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rt |= ((need[mt] >> tap) & 1u) << mt;
            if (!rt) continue;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const float *arow[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                arow[mt] = act + (((vmask[mt] >> tap) & 1u) ? ridx[mt] + (dy * G::COLS + dx) * stride : zero_idx);
            const float *wt = wq + (size_t)tap * (S / 4) * WST * 64 * 4;
            if constexpr (MT > 1) {
                // Eight quads (32 k-steps) of weights at a time in registers, then one straight-line block per row tile
                // that needs the tap.  (One interleaved body per tile mask was 8-15 % slower: the alternative bodies
                // meeting in one loop made the compiler keep 40-60 residual registers in scratch and schedule worse.)
                constexpr int NQ = (S / 4) % 8 == 0 ? 8 : (S / 4);
#pragma unroll
                for (int q0 = 0; q0 < S / 4; q0 += NQ) {
                    float4 t[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) t[q] = *reinterpret_cast<const float4 *>(wt + (size_t)(q0 + q) * WST * 64 * 4);
                    if constexpr (MT > 3) if (rt & 8u) tap_block<MT, NQ, (MT > 3 ? 3 : 0)>(acc, arow[MT > 3 ? 3 : 0], t, q0);
                    if constexpr (MT > 2) if (rt & 4u) tap_block<MT, NQ, (MT > 2 ? 2 : 0)>(acc, arow[MT > 2 ? 2 : 0], t, q0);
                    if (rt & 1u) tap_block<MT, NQ, 0>(acc, arow[0], t, q0);
                    if (rt & 2u) tap_block<MT, NQ, 1>(acc, arow[1], t, q0);
                }
                continue;
            }
            if (rt == (1u << MT) - 1u) tap_mfma_quad<MT, S, WST, (1 << MT) - 1>(acc, arow, wt, rt);
            else if (MT == 4 && rt == 0xEu) tap_mfma_quad<MT, S, WST, 0xE>(acc, arow, wt, rt);
            else if (MT == 4 && rt == 0xDu) tap_mfma_quad<MT, S, WST, 0xD>(acc, arow, wt, rt);
            else tap_mfma_quad<MT, S, WST, -1>(acc, arow, wt, rt);        // partly filled workgroups
        }
        return;
      }
    }
    if constexpr (PF == 0 && SK && MT == 2) {
        for (int tap = 0; tap < TAPS; ++tap) {
            const int t9 = (TAPS == 9) ? tap : 4;
            const bool n0 = (need[0] >> t9) & 1u, n1 = (need[1] >> t9) & 1u;
            if (!(n0 || n1)) continue;
            const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
            const float *arow[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                arow[mt] = act + (((vmask[mt] >> t9) & 1u) ? ridx[mt] + (dy * G::COLS + dx) * stride : zero_idx);
            const float *wt = wl + ((size_t)tap * S * 64 + lane) * WST;
            if (n0 && n1) tap_mfma<MT, NTW, S, WST, 3>(acc, arow, wt);
            else if (n0) tap_mfma<MT, NTW, S, WST, 1>(acc, arow, wt);
            else tap_mfma<MT, NTW, S, WST, 2>(acc, arow, wt);
        }
        return;
    }
    if constexpr (PF == 0) {
        // compiler-scheduled form: one load + MT LDS reads + MT*NTW MFMAs per k-step, fully unrolled per tap
        for (int tap = 0; tap < TAPS; ++tap) {
            const int t9 = (TAPS == 9) ? tap : 4;
            const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
            const float *arow[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                arow[mt] = act + (((vmask[mt] >> t9) & 1u) ? ridx[mt] + (dy * G::COLS + dx) * stride : zero_idx);
            const float *wt = wl + ((size_t)tap * S * 64 + lane) * WST;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float w[NTW];
                load_w<NTW>(wt + (size_t)s * 64 * WST, w);
                float a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt][2 * s];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], w[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        return;
    }
    // Weights stream through two register blocks of KB k-steps each: while the MFMAs of one block
    // issue (KB * MT * NTW * 64 cycles), the loads of the next block (possibly the next tap's first)
    // are in flight, so L2 latency never reaches the matrix pipe.  The buffer past the last block
    // of the layer is prefetched too (the blob is padded accordingly) and simply not used.
    constexpr int KB = 8, NB = S / KB;
    static_assert(S % (2 * KB) == 0, "k-steps per tap must be a multiple of 16");
    const float *wp = wl + (size_t)lane * WST;
    constexpr size_t BLK = (size_t)KB * 64 * WST;
    float wA[KB][NTW], wB[KB][NTW];
    load_wblock<NTW, KB, WST>(wp, wA);
    for (int tap = 0; tap < TAPS; ++tap) {
        const int t9 = (TAPS == 9) ? tap : 4;
        const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
        const float *arow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            arow[mt] = act + (((vmask[mt] >> t9) & 1u) ? ridx[mt] + (dy * G::COLS + dx) * stride : zero_idx);
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
            load_wblock<NTW, KB, WST>(wp + BLK, wB);
            mfma_block<MT, NTW, KB>(acc, arow, b * KB, wA);
            load_wblock<NTW, KB, WST>(wp + 2 * BLK, wA);
            mfma_block<MT, NTW, KB>(acc, arow, (b + 1) * KB, wB);
            wp += 2 * BLK;
        }
    }
}

// C/D layout of v_mfma_f32_32x32x2_f32: register v of lane l holds
//   row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5),  col = l & 31
__device__ __forceinline__ int cd_row(int v, int lane) { return (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5); }

template <class G, int NT, int MT, int PF, int CW = 1, int RW = 4, bool TR = false>
__global__ __launch_bounds__(64 * RW * CW, (RW * CW == 4 && CW == 1 && MT == 1 && NT <= 2 ? 3 : (RW * CW == 16 ? 1 : (RW * CW > 8 ? RW * CW / 4 : (RW * CW == 8 && RW * MT == 8 ? 1 : 2))))) void k_resnet_forward(NetParams P, const u64 *__restrict__ mine, const u64 *__restrict__ theirs,
                                                         const int32_t *__restrict__ leaf_slot, const int32_t *__restrict__ n_leaf_p,
                                                         float *__restrict__ out_p, float *__restrict__ out_v) {
    constexpr int CELLS = G::CELLS, COLS = G::COLS, ROWS = G::ROWS, A = G::A;
    constexpr int WG_ROWS = RW * 32 * MT;          // RW row groups x MT row tiles of 32 (x CW column groups of waves)
    constexpr int WROWS = 32 * MT;                 // rows per wave
    constexpr int NTHREADS = 64 * RW * CW;
    constexpr int TB = WG_ROWS / CELLS;            // boards per workgroup
    constexpr int CP = 32 * NT, S = CP / 2, STRIDE = CP + 1;
    constexpr int NTW = NT / CW;                   // column tiles per wave: CW waves share a row group
    static_assert(NT % CW == 0, "column tiles must divide over the column waves");
    constexpr int ZERO_IDX = WG_ROWS * STRIDE;
    // tap skipping: 6 Connect-4 boards per workgroup in the row order of build_row_order()
    constexpr bool SK = std::is_same<G, Connect4>::value && PF == 0 && ((MT == 2 && RW == 4) || (MT == 4 && RW == 2 && CW == NT));
    extern __shared__ float smem[];
    float *act = smem;                             // [WG_ROWS + 1][STRIDE]
    constexpr int AP = G::AP;                      // actions padded to a multiple of 8
    float *part = smem + (WG_ROWS + 1) * STRIDE;   // [WG_ROWS][8] FC partial sums, then [TB][AP] logits
    float *hval = part + WG_ROWS * 8 + TB * AP;    // [WG_ROWS][4] value-head activations
    int *s_perm = reinterpret_cast<int *>(hval + WG_ROWS * 4);   // [WG_ROWS] GEMM row -> board-cell row (SK only)

    const int n_leaf = *n_leaf_p;
    const int board0 = blockIdx.x * TB;
    if (board0 >= n_leaf) return;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = (tid >> 6) / CW, wc = (tid >> 6) % CW;    // row group, column group
    const int col0 = wc * NTW * 32;

    STAMP(0);
    for (int i = tid; i < STRIDE; i += NTHREADS) act[ZERO_IDX + i] = 0.0f;
    if constexpr (SK) {
        for (int i = tid; i < WG_ROWS; i += NTHREADS) s_perm[i] = P.perm[i];
        __syncthreads();
    }
    // LDS row of GEMM row g (activations are stored by board cell; GEMM rows may be permuted)
    auto lds_row = [&](int g) -> int { if constexpr (SK) return s_perm[g]; else return g; };

    // ---- per-lane row geometry -------------------------------------------------------------------
    int ridx[MT];
    uint32_t vmask[MT];
    typename G::B bm[MT], bt[MT];
    int ry[MT], rx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = lds_row(wave * WROWS + mt * 32 + (lane & 31));
        const int b = r / CELLS, pos = r % CELLS;
        const int y = pos / COLS, x = pos % COLS;
        const bool real = r < TB * CELLS && board0 + b < n_leaf;
        uint32_t m = 0;
        if (real)
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS) m |= 1u << t;
            }
        vmask[mt] = m;
        ridx[mt] = r * STRIDE + half;
        ry[mt] = y;
        rx[mt] = x;
        bm[mt] = board_zero<G::W>();
        bt[mt] = board_zero<G::W>();
        if (real) {
            const int slot = leaf_slot[board0 + b];
            bm[mt] = board_load<G::W>(mine, slot);
            bt[mt] = board_load<G::W>(theirs, slot);
        }
    }

    uint32_t need[MT];            // wave-uniform: taps some row of the tile needs
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        uint32_t nd = 0;
        for (int t = 0; t < 9; ++t) if (__ballot((vmask[mt] >> t) & 1u)) nd |= 1u << t;
        need[mt] = nd;
    }

    f32x16 acc[MT][NTW], xres[MT][NTW];
    STAMP(1);

    // ---- stem: conv3x3(3 -> C) + BN + ReLU, A fragments generated from the bitboards ---------------
    // (_states_to_tensor resnet.py:115-128: planes [empty, mine(+1), theirs(-1)]; input_conv :44-48)
    {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float bias = P.stem_b[col0 + nt * 32 + (lane & 31)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[mt][nt][v] = bias;
        }
        for (int s = 0; s < 14; ++s) {
            const int kk = 2 * s + half;               // K index = tap*3 + plane, 27 real + 1 zero
            const int tap = kk / 3, plane = kk % 3;
            float w[NTW];
            load_w<NTW>(P.stem_w + ((size_t)s * 64 + lane) * NT + wc * NTW, w);
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float val = 0.0f;
                if (kk < 27 && ((vmask[mt] >> tap) & 1u)) {
                    const int yy = ry[mt] + tap / 3 - 1, xx = rx[mt] + tap % 3 - 1;
                    const int bit = G::bit_of(yy, xx);
                    const int mi = board_bit(bm[mt], bit) ? 1 : 0, th = board_bit(bt[mt], bit) ? 1 : 0;
                    const int on = plane == 0 ? (1 - mi) * (1 - th) : (plane == 1 ? mi : th);
                    val = (float)on;
                }
                a[mt] = val;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], w[nt], acc[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float o = fmaxf(acc[mt][nt][v], 0.0f);
                    xres[mt][nt][v] = o;
                    act[lds_row(wave * WROWS + mt * 32 + cd_row(v, lane)) * STRIDE + col0 + nt * 32 + (lane & 31)] = o;
                }
    }
    __syncthreads();
    STAMP(2);
    if constexpr (TR) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + 56] = __builtin_amdgcn_s_memtime(); }   // core-clock counter

    // ---- residual blocks (ResBlock.forward resnet.py:21-27) -----------------------------------------
    for (int layer = 0; layer < 2 * P.blocks; ++layer) {
        const bool second = layer & 1;
        const float *wl = P.conv_w + (size_t)layer * P.conv_w_stride;
        const float *bl = P.conv_b + (size_t)layer * P.conv_b_stride;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float bias = bl[col0 + nt * 32 + (lane & 31)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[mt][nt][v] = second ? xres[mt][nt][v] + bias : bias;   // x += residual
        }
        conv_mfma<G, MT, NTW, 9, S, PF, NT, SK>(acc, wl + wc * NTW, act, ridx, vmask, STRIDE, ZERO_IDX + half, lane, need,
                                                (CW == NT && P.conv_w4) ? P.conv_w4 + (size_t)layer * P.conv_w_stride + ((size_t)wc * 64 + lane) * 4 : nullptr);
        if (layer < 16) STAMP(3 + 3 * layer);
        __syncthreads();                               // every wave is done reading the layer input
        if (layer < 16) STAMP(4 + 3 * layer);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float o = fmaxf(acc[mt][nt][v], 0.0f);
                    if (second) xres[mt][nt][v] = o;
                    act[lds_row(wave * WROWS + mt * 32 + cd_row(v, lane)) * STRIDE + col0 + nt * 32 + (lane & 31)] = o;
                }
        __syncthreads();
        if (layer < 16) STAMP(5 + 3 * layer);
    }

    if constexpr (TR) {
        if (tid == 0) {
            P.dbg[(size_t)blockIdx.x * 64 + 57] = __builtin_amdgcn_s_memtime();
            P.dbg[(size_t)blockIdx.x * 64 + 58] = wall_clock64();
        }
    }
    // ---- heads: policy conv1x1(C->32)+BN+ReLU (resnet.py:56-59), value conv3x3(C->3)+BN+ReLU (:65-68)
    // The policy conv is ONE 32-channel column tile.  Where the row tiles of a row group divide evenly over its column
    // waves each takes its share; otherwise the first column wave computes all of them.
    constexpr int PSPLIT = (MT % CW == 0) ? CW : 1, PMT = MT / PSPLIT;
    const int pt0 = PSPLIT > 1 ? wc * PMT : 0;             // first row tile (of this wave's row group) handled here
    const bool pol_here = PSPLIT > 1 || wc == 0;
    f32x16 accp[PMT][1];
    int pridx[PMT];
    uint32_t pvmask[PMT], pneed[PMT];
    {
        const float bp = P.pol_b[lane & 31];
#pragma unroll
        for (int i = 0; i < PMT; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) accp[i][0][v] = bp;
            pridx[i] = ridx[0]; pvmask[i] = vmask[0]; pneed[i] = need[0];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) if (mt == pt0 + i) { pridx[i] = ridx[mt]; pvmask[i] = vmask[mt]; pneed[i] = need[mt]; }
        }
    }
    if (pol_here)
        conv_mfma<G, PMT, 1, 1, S, PF, 1, false>(accp, P.pol_w, act, pridx, pvmask, STRIDE, ZERO_IDX + half, lane, pneed);
    STAMP(51);
    // The value conv has 3 output channels.  As a 32-wide MFMA tile it would cost +5.8 % MFMA time for 0.6 % of the
    // FLOPs; on the vector ALU it took 16 us of a 370 us workgroup (scalar weight loads and LDS reads sharing one
    // wait counter).  It runs on v_mfma_f32_4x4x1_16B_f32 instead: 16 independent 4x4 blocks per instruction, a
    // block = 4 board cells x 4 output channels (3 real), K = 1, so 64 cells advance one input channel per 8-cycle
    // instruction; every cell's sum is a fixed-order fma chain per kernel row (batch-invariant).
    // Operand layout (probed on gfx950): A row i of block b from lane 4b+i, B column j from lane 4b+j,
    // D[i][j] in VGPR i of lane 4b+j.
    constexpr bool VMFMA = (WG_ROWS % 64 == 0) && (RW * CW * 64 >= WG_ROWS);
    float hv0 = 0.0f, hv1 = 0.0f, hv2 = 0.0f;
    if constexpr (VMFMA) {
        constexpr int NCHUNK = WG_ROWS / 64;           // 64-row chunks; wave w owns chunk w
        const int wv = tid >> 6;
        if (wv < NCHUNK) {
            const int r = wv * 64 + lane;
            const int pos = r % CELLS, y = pos / COLS, x = pos % COLS;
            const bool real = r < TB * CELLS && board0 + r / CELLS < n_leaf;
            const int j = lane & 3;
            const float bj = j < VAL_CH ? P.val_b[j] : 0.0f;
            // three independent accumulation chains, one per kernel row (dy = -1, 0, +1), summed at the end in a
            // fixed order: dependent 4x4x1 MFMAs issue only every ~4 passes, independent ones back to back
            f32x4 ar[3] = {{bj, bj, bj, bj}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float *xr[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const bool ok = real && yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS;
                xr[t] = act + (ok ? (r + (t / 3 - 1) * COLS + (t % 3 - 1)) * STRIDE : ZERO_IDX);
            }
            const float *wt = P.val_w4 + j * 4;                                // [tap][c / 4][j][c % 4]
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll 2
                for (int c = 0; c < CP; c += 4) {
                    float4 w[3];
                    const float *a[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        w[dy] = *reinterpret_cast<const float4 *>(wt + ((size_t)(dy * 3 + dx) * CP + c) * 4);
                        a[dy] = dx == 0 ? xr[dy * 3] : (dx == 1 ? xr[dy * 3 + 1] : xr[dy * 3 + 2]);
                    }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) ar[dy] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[dy][c + 0], w[dy].x, ar[dy], 0, 0, 0);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) ar[dy] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[dy][c + 1], w[dy].y, ar[dy], 0, 0, 0);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) ar[dy] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[dy][c + 2], w[dy].z, ar[dy], 0, 0, 0);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) ar[dy] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[dy][c + 3], w[dy].w, ar[dy], 0, 0, 0);
                }
            }
            f32x4 av;
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = (ar[0][i] + ar[1][i]) + ar[2][i];
            const int rb = wv * 64 + (lane & ~3);
#pragma unroll
            for (int i = 0; i < 4; ++i) hval[(rb + i) * 4 + j] = fmaxf(av[i], 0.0f);   // hval is not read before the barriers below
        }
    } else if (tid < TB * CELLS) {
        // other wave grids: one thread per board cell on the vector ALU, weights wave-uniform; the SAME three
        // fma chains (one per kernel row, bias in the first, summed (r0 + r1) + r2) as the MFMA form, so that every
        // variant of the kernel returns the same bits for a position
        const int pos = tid % CELLS, y = pos / COLS, x = pos % COLS;
        const bool real = board0 + tid / CELLS < n_leaf;
        float hr[3][3] = {{P.val_b[0], P.val_b[1], P.val_b[2]}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
        for (int dx = 0; dx < 3; ++dx) {
            const float *xr[3], *wv[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                const bool ok = real && yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS;
                xr[dy] = act + (ok ? (tid + (dy - 1) * COLS + (dx - 1)) * STRIDE : ZERO_IDX);
                wv[dy] = P.val_w + (size_t)(dy * 3 + dx) * CP * 4;
            }
#pragma unroll 4
            for (int c = 0; c < CP; ++c)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float a = xr[dy][c];
                    hr[dy][0] = fmaf(a, wv[dy][c * 4 + 0], hr[dy][0]);
                    hr[dy][1] = fmaf(a, wv[dy][c * 4 + 1], hr[dy][1]);
                    hr[dy][2] = fmaf(a, wv[dy][c * 4 + 2], hr[dy][2]);
                }
        }
        hv0 = (hr[0][0] + hr[1][0]) + hr[2][0];
        hv1 = (hr[0][1] + hr[1][1]) + hr[2][1];
        hv2 = (hr[0][2] + hr[1][2]) + hr[2][2];
    }
    STAMP(52);
    __syncthreads();
    // policy-head activations over the (now dead) tower activations, value-head ones beside them
#pragma unroll
    for (int i = 0; i < PMT; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = lds_row(wave * WROWS + (pt0 + i) * 32 + cd_row(v, lane));
            if (pol_here) act[row * STRIDE + (lane & 31)] = fmaxf(accp[i][0][v], 0.0f);
        }
    if (!VMFMA && tid < TB * CELLS) {
        hval[tid * 4 + 0] = fmaxf(hv0, 0.0f);
        hval[tid * 4 + 1] = fmaxf(hv1, 0.0f);
        hval[tid * 4 + 2] = fmaxf(hv2, 0.0f);
    }
    __syncthreads();
    STAMP(53);

    // ---- Linear layers: Flatten is (c, y, x) order (resnet.py:60-61, :69-70) -------------------------
    float *logits = part + WG_ROWS * 8;            // [TB][AP]; value logit of board b at vlogit[b]
    float *vlogit = logits + TB * AP - 0;          // (for A <= 7 the value logit lives in logits[b][7])
    if constexpr (A <= 7) {
        static_assert(TB * CELLS <= NTHREADS, "one thread per board cell");
        // thread r (< TB*CELLS) = one board cell: partial dot products over channels, fixed order
        if (tid < TB * CELLS) {
            const int pos = tid % CELLS;
            float pa[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) pa[o] = 0.0f;
            const float *hrow = act + tid * STRIDE;
            for (int c = 0; c < POL_CH; ++c) {
                const float h = hrow[c];
                const float4 w0 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8);
                const float4 w1 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8 + 4);
                pa[0] = fmaf(h, w0.x, pa[0]); pa[1] = fmaf(h, w0.y, pa[1]); pa[2] = fmaf(h, w0.z, pa[2]); pa[3] = fmaf(h, w0.w, pa[3]);
                pa[4] = fmaf(h, w1.x, pa[4]); pa[5] = fmaf(h, w1.y, pa[5]); pa[6] = fmaf(h, w1.z, pa[6]);
            }
            float pv = 0.0f;
#pragma unroll
            for (int c = 0; c < VAL_CH; ++c) pv = fmaf(hval[tid * 4 + c], P.fcv_w[c * CELLS + pos], pv);
            pa[7] = pv;
#pragma unroll
            for (int o = 0; o < 8; ++o) part[tid * 8 + o] = pa[o];
        }
        __syncthreads();
        if (tid < TB * 8) {
            const int b = tid >> 3, o = tid & 7;
            float sum = 0.0f;
            for (int pos = 0; pos < CELLS; ++pos) sum += part[(b * CELLS + pos) * 8 + o];
            logits[tid] = sum + (o < 7 ? (o < A ? P.fcp_b[o] : 0.0f) : P.fcv_b[0]);
        }
        vlogit = nullptr;
    } else {
        // wide action spaces (Gomoku: 81): one thread per (board, action) walks the 32 x CELLS inputs in
        // flatten order; weights are stored [input][AP] so that the threads of a board read them coalesced
#pragma unroll 1
        for (int t = tid; t < TB * (A + 1); t += NTHREADS) {
            const int b = t / (A + 1), o = t % (A + 1);
            float sum = 0.0f;
            if (o < A) {
#pragma unroll 1
                for (int c = 0; c < POL_CH; ++c)
#pragma unroll 3
                    for (int pos = 0; pos < CELLS; ++pos)
                        sum = fmaf(act[(b * CELLS + pos) * STRIDE + c], P.fcp_w[((size_t)c * CELLS + pos) * AP + o], sum);
                logits[b * AP + o] = sum + P.fcp_b[o];
            } else {
#pragma unroll 1
                for (int c = 0; c < VAL_CH; ++c)
#pragma unroll 1
                    for (int pos = 0; pos < CELLS; ++pos)
                        sum = fmaf(hval[(b * CELLS + pos) * 4 + c], P.fcv_w[c * CELLS + pos], sum);
                part[b] = sum + P.fcv_b[0];
            }
        }
        vlogit = part;
    }
    __syncthreads();
    STAMP(54);
    // ---- masked softmax + tanh (ResNet.predict resnet.py:143-151; legal = the game's valid moves) --------
    if (tid < TB && board0 + tid < n_leaf) {
        const int j = board0 + tid;
        const int slot = leaf_slot[j];
        const typename G::B bmine = board_load<G::W>(mine, slot), btheirs = board_load<G::W>(theirs, slot);
        const float *lg = logits + tid * AP;
        float mx = -INFINITY;
#pragma unroll 1
        for (int a = 0; a < A; ++a) if (G::legal(bmine, btheirs, a)) mx = fmaxf(mx, lg[a]);
        float den = 0.0f;
#pragma unroll 1
        for (int a = 0; a < A; ++a) den += G::legal(bmine, btheirs, a) ? expf(lg[a] - mx) : 0.0f;
#pragma unroll 1
        for (int a = 0; a < AP; ++a)
            out_p[(size_t)j * AP + a] = (a < A && G::legal(bmine, btheirs, a)) ? expf(lg[a] - mx) / den : 0.0f;
        out_v[j] = tanhf(vlogit ? vlogit[tid] : lg[7]);
    }
    STAMP(55);
}

// ------------------------------------------------------------------------------------------------
// host: geometry, BN folding, weight permutation
// ------------------------------------------------------------------------------------------------
static bool geom_for(int game, int blocks, int channels, NetGeom &g) {
    // The reference network is Connect-4 only (ResNet.predict hard-codes Connect4Game, resnet.py:135).
    if (game == 0) { g.rows = Connect4::ROWS; g.cols = Connect4::COLS; g.cells = Connect4::CELLS; g.A = Connect4::A; g.AP = Connect4::AP; }
    // Gomoku 9x9 (not a reference network: the same architecture on the plugin game, legal = empty cells)
    else if (game == 2) { g.rows = Gomoku9::ROWS; g.cols = Gomoku9::COLS; g.cells = Gomoku9::CELLS; g.A = Gomoku9::A; g.AP = Gomoku9::AP; }
    else return false;
    if (blocks < 0 || channels <= 0) return false;
    g.C = channels;
    g.Cp = (channels + 31) / 32 * 32;
    g.NT = g.Cp / 32;
    if (g.NT == 5 || g.NT == 7) { g.NT += 1; g.Cp += 32; }       // wave grids exist for 1, 2, 3, 4, 6 and 8 column tiles
    if (g.NT < 1 || g.NT > 8) return false;      // up to 256 channels
    g.S = g.Cp / 2;
    g.blocks = blocks;
    g.TB = 256 / g.cells;
    return true;
}

size_t resnet_blob_floats(int game, int blocks, int channels) {
    NetGeom g;
    if (!geom_for(game, blocks, channels, g)) return 0;
    const size_t C = g.C, cells = g.cells;
    size_t n = C * 3 * 9 + C + 4 * C;                               // input_conv.0 + input_conv.1
    n += (size_t)blocks * 2 * (C * C * 9 + C + 4 * C);              // residual blocks
    n += POL_CH * C + POL_CH + 4 * POL_CH + (size_t)g.A * POL_CH * cells + g.A;   // policy head
    n += VAL_CH * C * 9 + VAL_CH + 4 * VAL_CH + VAL_CH * cells + 1;               // value head
    return n;
}

// B-fragment order for a KxK conv: dst[((tap*S + s)*64 + lane)*NTW + nt] = W[oc = nt*32 + (lane&31)][ic = 2s + (lane>>5)][tap]
static void permute_conv(const std::vector<float> &w, int oc, int ic, int k, int S, int NTW, float *dst) {
    const int taps = k * k;
    for (int tap = 0; tap < taps; ++tap)
        for (int s = 0; s < S; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int nt = 0; nt < NTW; ++nt) {
                    const int o = nt * 32 + (lane & 31), i = 2 * s + (lane >> 5);
                    float val = 0.0f;
                    if (o < oc && i < ic) val = w[((size_t)o * ic + i) * taps + tap];
                    dst[(((size_t)tap * S + s) * 64 + lane) * NTW + nt] = val;
                }
}

// Row order of the tap-skipping kernels (6 Connect-4 boards = 252 cells + 4 padding rows in 8 row tiles of 32).
// A 3x3 tap that falls off the board multiplies zeros; 74 of the 378 (cell, tap) pairs of a 6x7 board do.  A
// 32-row MFMA tile can only drop a tap that ALL its rows lack, so the cells are grouped by which taps they lack:
//   tile 0 of wave 0: 32 cells of the left column   (lack dx = -1: 3 taps)
//   tile 0 of wave 1: 32 cells of the right column  (lack dx = +1)
//   tile 0 of wave 2: the 30 inner cells of the top rows + 2 top-left corners     (lack dy = -1)
//   tile 0 of wave 3: the 30 inner cells of the bottom rows + 2 bottom-left corners (lack dy = +1)
//   tile 1 of every wave: interior cells, the 4 right-column cells left over and the 4 padding rows (all 9 taps)
// Every wave then issues 15 instead of 18 tile-taps per layer (-16.7 % MFMAs), identically for all waves, and the
// skipped products are exact zeros, so results do not change by a bit.
static void build_row_order(int32_t *perm) {
    constexpr int R = Connect4::ROWS, C = Connect4::COLS, CE = Connect4::CELLS, TBB = 6;
    std::vector<int> left, right, top, bottom, interior, rest;
    for (int b = 0; b < TBB; ++b)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x) {
                const int r = b * CE + y * C + x;
                const bool corner_l = x == 0 && (y == 0 || y == R - 1) && b < 2;    // 2 top-left + 2 bottom-left corners
                if (x == 0 && !corner_l) left.push_back(r);
                else if (x == 0) (y == 0 ? top : bottom).push_back(r);
                else if (x == C - 1) right.push_back(r);
                else if (y == 0) top.push_back(r);
                else if (y == R - 1) bottom.push_back(r);
                else interior.push_back(r);
            }
    while (right.size() > 32) { rest.push_back(right.back()); right.pop_back(); }
    for (int r = TBB * CE; r < 256; ++r) rest.push_back(r);
    interior.insert(interior.end(), rest.begin(), rest.end());      // 120 + 4 + 4 = 128
    const std::vector<int> *special[4] = {&left, &right, &top, &bottom};
    for (int w = 0; w < 4; ++w) {
        for (int i = 0; i < 32; ++i) perm[w * 64 + i] = (*special[w])[i];
        for (int i = 0; i < 32; ++i) perm[w * 64 + 32 + i] = interior[w * 32 + i];
    }
    // second table: 2 x 2 wave grid, a row group = 4 row tiles: [left, right, interior, interior | top, bottom, interior, interior]
    int32_t *q = perm + 256;
    for (int grp = 0; grp < 2; ++grp) {
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 32; ++i) q[grp * 128 + k * 32 + i] = (*special[grp * 2 + k])[i];
        for (int i = 0; i < 64; ++i) q[grp * 128 + 64 + i] = interior[grp * 64 + i];
    }
}

ResNetDev *resnet_create(int game, int blocks, int channels, bool f16_split) {
    NetGeom g;
    if (!geom_for(game, blocks, channels, g)) return nullptr;
    ResNetDev *n = new ResNetDev();
    n->g = g;
    n->game = game;
    size_t off = 0;
    auto reserve = [&](size_t cnt) { size_t o = off; off += (cnt + 3) & ~(size_t)3; return o; };
    n->o_stem_w = reserve((size_t)14 * 64 * g.NT);
    n->o_stem_b = reserve(g.Cp);
    n->conv_w_stride = ((size_t)9 * g.S * 64 * g.NT + 3) & ~(size_t)3;
    n->conv_b_stride = g.Cp;
    n->o_conv_w = reserve(n->conv_w_stride * 2 * blocks);
    n->o_conv_b = reserve(n->conv_b_stride * 2 * blocks);
    if ((game == 0 && (g.NT == 2 || g.NT == 4)) || (game == 2 && g.NT >= 3)) { n->has_w4 = true; n->o_conv_w4 = reserve(n->conv_w_stride * 2 * blocks); }
    n->o_pol_w = reserve((size_t)g.S * 64);
    n->o_pol_b = reserve(32);
    n->o_val_w = reserve((size_t)9 * g.Cp * 4);
    n->o_val_w4 = reserve((size_t)9 * g.Cp * 4);
    n->o_val_b = reserve(32);
    n->o_fcp_w = reserve((size_t)POL_CH * g.cells * g.AP);
    n->o_fcp_b = reserve(g.AP);
    n->o_fcv_w = reserve((size_t)VAL_CH * g.cells);
    n->o_fcv_b = reserve(4);
    off += 8 * 64 * 4;                      // slack for the one-block weight prefetch past the last array
    n->blob_floats = off;
    if (hipMalloc((void **)&n->d_blob, off * sizeof(float)) != hipSuccess) { delete n; return nullptr; }
    if (game == 0) {
        int32_t perm[512];
        build_row_order(perm);
        if (hipMalloc((void **)&n->d_perm, sizeof perm) != hipSuccess ||
            hipMemcpy(n->d_perm, perm, sizeof perm, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(n->d_blob); delete n; return nullptr; }
    }
    const double C = g.C, cells = g.cells;
    n->flops_per_board = 2.0 * cells * (3 * 9 * C + 2.0 * blocks * C * C * 9 + C * POL_CH + C * 9 * VAL_CH) +
                         2.0 * (POL_CH * cells * g.A + VAL_CH * cells);
    if (f16_split) {
        n->f16 = resnet16_create(game, blocks, channels);
        if (!n->f16) { (void)hipFree(n->d_blob); delete n; return nullptr; }
    }
    return n;
}

void resnet_destroy(ResNetDev *n) {
    if (!n) return;
    if (n->d_blob) (void)hipFree(n->d_blob);
    if (n->d_perm) (void)hipFree(n->d_perm);
    if (n->f16) resnet16_destroy(n->f16);
    delete n;
}

double resnet_flops_per_board(const ResNetDev *n) { return n ? n->flops_per_board : 0.0; }

int resnet_load(ResNetDev *n, const float *blob, size_t n_floats, hipStream_t s) {
    const NetGeom &g = n->g;
    if (n_floats != resnet_blob_floats(n->game, g.blocks, g.C)) return -1;
    std::vector<float> host(n->blob_floats, 0.0f), w, b;
    const float *p = blob;
    // stem: K index kk = tap*3 + plane
    {
        ConvBN c = take_conv_bn(p, g.C, 3, 3);
        fold(c, w, b);
        float *dst = host.data() + n->o_stem_w;
        for (int sidx = 0; sidx < 14; ++sidx)
            for (int lane = 0; lane < 64; ++lane)
                for (int nt = 0; nt < g.NT; ++nt) {
                    const int kk = 2 * sidx + (lane >> 5), o = nt * 32 + (lane & 31);
                    float val = 0.0f;
                    if (kk < 27 && o < g.C) val = w[((size_t)o * 3 + kk % 3) * 9 + kk / 3];
                    dst[((size_t)sidx * 64 + lane) * g.NT + nt] = val;
                }
        for (int o = 0; o < g.C; ++o) host[n->o_stem_b + o] = b[o];
    }
    for (int l = 0; l < 2 * g.blocks; ++l) {
        ConvBN c = take_conv_bn(p, g.C, g.C, 3);
        fold(c, w, b);
        permute_conv(w, g.C, g.C, 3, g.S, g.NT, host.data() + n->o_conv_w + (size_t)l * n->conv_w_stride);
        if (n->has_w4) {            // [tap][s / 4][nt][lane][s & 3] from [tap][s][lane][nt]
            const float *src = host.data() + n->o_conv_w + (size_t)l * n->conv_w_stride;
            float *dst = host.data() + n->o_conv_w4 + (size_t)l * n->conv_w_stride;
            for (int tap = 0; tap < 9; ++tap)
                for (int sidx = 0; sidx < g.S; ++sidx)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int nt = 0; nt < g.NT; ++nt)
                            dst[((((size_t)tap * (g.S / 4) + sidx / 4) * g.NT + nt) * 64 + lane) * 4 + (sidx & 3)] =
                                src[(((size_t)tap * g.S + sidx) * 64 + lane) * g.NT + nt];
        }
        for (int o = 0; o < g.C; ++o) host[n->o_conv_b + (size_t)l * n->conv_b_stride + o] = b[o];
    }
    {
        ConvBN c = take_conv_bn(p, POL_CH, g.C, 1);
        fold(c, w, b);
        permute_conv(w, POL_CH, g.C, 1, g.S, 1, host.data() + n->o_pol_w);
        for (int o = 0; o < POL_CH; ++o) host[n->o_pol_b + o] = b[o];
        const float *fw = take(p, (size_t)g.A * POL_CH * g.cells), *fb = take(p, g.A);
        for (int c2 = 0; c2 < POL_CH; ++c2)
            for (int pos = 0; pos < g.cells; ++pos)
                for (int o = 0; o < g.A; ++o)
                    host[n->o_fcp_w + ((size_t)c2 * g.cells + pos) * g.AP + o] = fw[(size_t)o * POL_CH * g.cells + (size_t)c2 * g.cells + pos];
        for (int o = 0; o < g.A; ++o) host[n->o_fcp_b + o] = fb[o];
    }
    {
        ConvBN c = take_conv_bn(p, VAL_CH, g.C, 3);
        fold(c, w, b);
        for (int tap = 0; tap < 9; ++tap)           // vector-ALU layout: [tap][channel][4]
            for (int c2 = 0; c2 < g.C; ++c2)
                for (int o = 0; o < VAL_CH; ++o)
                    host[n->o_val_w + ((size_t)tap * g.Cp + c2) * 4 + o] = w[((size_t)o * g.C + c2) * 9 + tap];
        for (int tap = 0; tap < 9; ++tap)           // 4x4x1-MFMA layout: [tap][channel / 4][out][channel % 4]
            for (int c2 = 0; c2 < g.C; ++c2)
                for (int o = 0; o < VAL_CH; ++o)
                    host[n->o_val_w4 + (((size_t)tap * g.Cp + (c2 & ~3)) + o) * 4 + (c2 & 3)] = w[((size_t)o * g.C + c2) * 9 + tap];
        for (int o = 0; o < VAL_CH; ++o) host[n->o_val_b + o] = b[o];
        const float *fw = take(p, (size_t)VAL_CH * g.cells), *fb = take(p, 1);
        for (int i = 0; i < VAL_CH * g.cells; ++i) host[n->o_fcv_w + i] = fw[i];
        host[n->o_fcv_b] = fb[0];
    }
    if ((size_t)(p - blob) != n_floats) return -2;
    if (hipMemcpyAsync(n->d_blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return -3;
    if (hipStreamSynchronize(s) != hipSuccess) return -4;
    if (n->f16) return resnet16_load(n->f16, blob, n_floats, s);
    return 0;
}

// Tracing is on for a thread while AZSP_RESNET_TRACE is set or resnet_trace_next() armed it for one launch
// (azsp_probe_clock); the last traced launch's in-kernel clock and span are kept for that probe.
static thread_local bool g_trace_armed = false;
static thread_local double g_trace_ghz = 0.0, g_trace_span_us = 0.0;
void resnet_trace_next() { g_trace_armed = true; g_trace_ghz = 0.0; g_trace_span_us = 0.0; }
void resnet_trace_result(double *ghz, double *span_us) { if (ghz) *ghz = g_trace_ghz; if (span_us) *span_us = g_trace_span_us; }

// Diagnostic: run the launch with per-workgroup phase timestamps, wait for it and (AZSP_RESNET_TRACE) print the
// mean duration of every phase over the workgroups that ran (microseconds; 100 MHz clock) to stderr.
template <class Launch>
static void trace_forward(const NetParams &P, int grid, Launch launch, hipStream_t s, bool print) {
    NetParams Q = P;
    unsigned long long *d = nullptr;
    const size_t n = (size_t)grid * 64;
    if (hipMalloc((void **)&d, n * 8) != hipSuccess) { launch(P); return; }
    (void)hipMemsetAsync(d, 0, n * 8, s);
    Q.dbg = d;
    launch(Q);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    {   // in-kernel clock over the residual tower: core-clock ticks / 100 MHz ticks, mean over the workgroups
        double ghz = 0;
        int cnt = 0;
        unsigned long long a = ~0ull, b = 0;
        for (int w = 0; w < grid; ++w) {
            const unsigned long long *t = &h[(size_t)w * 64];
            if (!t[55]) continue;
            a = t[0] < a ? t[0] : a;
            b = t[55] > b ? t[55] : b;
            if (t[58] > t[2]) { ghz += (double)(t[57] - t[56]) / (double)(t[58] - t[2]) * 0.1; ++cnt; }
        }
        g_trace_ghz = cnt ? ghz / cnt : 0.0;
        g_trace_span_us = b > a ? (b - a) * 0.01 : 0.0;
    }
    static int calls = 0;
    static const bool verbose_probe = getenv("AZSP_PROBE_VERBOSE") != nullptr;     // print armed (warm) launches too
    if (!(print && ++calls <= 3) && !(verbose_probe && !print)) return;
    double sum[64] = {};
    unsigned long long t_min = ~0ull, t_max = 0;
    int ran = 0;
    for (int w = 0; w < grid; ++w) {
        const unsigned long long *t = &h[(size_t)w * 64];
        if (!t[55]) continue;
        ++ran;
        t_min = t[0] < t_min ? t[0] : t_min;
        t_max = t[55] > t_max ? t[55] : t_max;
        unsigned long long prev = t[0];
        for (int k = 1; k < 56; ++k) if (t[k]) { sum[k] += (double)(t[k] - prev); prev = t[k]; }
    }
    if (!ran) return;
    fprintf(stderr, "[azsp trace] %d workgroups, span %.1f us; mean phase durations (us):", ran, (t_max - t_min) * 0.01);
    for (int k = 1; k < 56; ++k) if (sum[k] > 0) fprintf(stderr, " %d:%.2f", k, sum[k] / ran * 0.01);
    // in-kernel clock over the residual tower: core-clock ticks / 100 MHz ticks (median would be better; mean is fine here)
    double ghz = 0;
    int cnt = 0;
    for (int w = 0; w < grid; ++w) {
        const unsigned long long *t = &h[(size_t)w * 64];
        if (t[55] && t[58] > t[2]) { ghz += (double)(t[57] - t[56]) / (double)(t[58] - t[2]) * 0.1; ++cnt; }
    }
    if (cnt) fprintf(stderr, " | tower clock %.3f GHz", ghz / cnt);
    fprintf(stderr, "\n");
}

template <class G, int NT, int MT, int PF, int CW = 1, int RW = 4>
static void launch_forward(const NetParams &P, const u64 *mine, const u64 *theirs, const int32_t *leaf_slot,
                           const int32_t *n_leaf, int max_leaves, float *out_p, float *out_v, hipStream_t s) {
    constexpr int WG_ROWS = RW * 32 * MT;
    constexpr int TB = WG_ROWS / G::CELLS;
    static_assert(TB >= 1 && TB * G::CELLS <= 64 * RW * CW, "one thread per board cell in the head phases");
    constexpr int STRIDE = 32 * NT + 1;
    const size_t lds = ((size_t)(WG_ROWS + 1) * STRIDE + WG_ROWS * 8 + TB * G::AP + WG_ROWS * 4 + WG_ROWS) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resnet_forward<G, NT, MT, PF, CW, RW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    const int grid = (max_leaves + TB - 1) / TB;
    if constexpr (std::is_same<G, Connect4>::value && NT == 2 && (CW == 1 || MT == 4)) {     // traced instantiations: the 64-channel Connect-4 kernels
        static const bool trace_env = getenv("AZSP_RESNET_TRACE") != nullptr;
        const bool armed = g_trace_armed;
        g_trace_armed = false;
        if (trace_env || armed) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resnet_forward<G, NT, MT, PF, CW, RW, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            trace_forward(P, grid, [&](const NetParams &Q) {
                hipLaunchKernelGGL((k_resnet_forward<G, NT, MT, PF, CW, RW, true>), dim3(grid), dim3(64 * RW * CW), lds, s, Q, mine, theirs, leaf_slot, n_leaf, out_p, out_v); }, s, trace_env);
            return;
        }
    }
    hipLaunchKernelGGL((k_resnet_forward<G, NT, MT, PF, CW, RW>), dim3(grid), dim3(64 * RW * CW), lds, s, P, mine, theirs, leaf_slot, n_leaf, out_p, out_v);
}

void resnet_forward(ResNetDev *n, int game, const u64 *mine, const u64 *theirs, const int32_t *leaf_slot, const int32_t *n_leaf,
                    int max_leaves, float *out_p, float *out_v, hipStream_t s) {
    if (n->f16) { resnet16_forward(n->f16, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); return; }
    NetParams P;
    const float *d = n->d_blob;
    P.stem_w = d + n->o_stem_w; P.stem_b = d + n->o_stem_b;
    P.conv_w = d + n->o_conv_w; P.conv_b = d + n->o_conv_b;
    P.pol_w = d + n->o_pol_w; P.pol_b = d + n->o_pol_b;
    P.val_w = d + n->o_val_w; P.val_w4 = d + n->o_val_w4; P.val_b = d + n->o_val_b;
    P.fcp_w = d + n->o_fcp_w; P.fcp_b = d + n->o_fcp_b;
    P.fcv_w = d + n->o_fcv_w; P.fcv_b = d + n->o_fcv_b;
    P.conv_w_stride = (int)n->conv_w_stride;
    P.conv_b_stride = (int)n->conv_b_stride;
    P.blocks = n->g.blocks;
    P.dbg = nullptr;
    P.perm = n->d_perm;
    P.conv_w4 = n->has_w4 ? d + n->o_conv_w4 : nullptr;
    if (game == 2) {       // Gomoku 9x9: 3 boards (243 rows) per workgroup up to 64 channels, 1 board (81 of 128 rows) above
#define FWG(NTV, MTV, PFV) launch_forward<Gomoku9, NTV, MTV, PFV>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s)
        // wide nets: one board (81 rows -> 3 row tiles) per workgroup, one wave per 32-channel column tile
        // (each wave owns one column tile: weights as one 16-byte load per four k-steps; AZSP_RESNET_GQUAD=0 = the
        //  per-k-step double-buffered stream)
        static const bool gquad = [] { const char *e = getenv("AZSP_RESNET_GQUAD"); return !e || atoi(e) != 0; }();
        if (n->g.NT == 1) FWG(1, 2, 0); else if (n->g.NT == 2) FWG(2, 2, 0);
        else if (n->g.NT == 3) { if (gquad && P.conv_w4) launch_forward<Gomoku9, 3, 3, 0, 3, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); else launch_forward<Gomoku9, 3, 3, 1, 3, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); }
        else if (n->g.NT == 4) { if (gquad && P.conv_w4) launch_forward<Gomoku9, 4, 3, 0, 4, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); else launch_forward<Gomoku9, 4, 3, 1, 4, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); }
        else if (n->g.NT == 6) { if (gquad && P.conv_w4) launch_forward<Gomoku9, 6, 3, 0, 6, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); else launch_forward<Gomoku9, 6, 3, 1, 6, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); }
        else { if (gquad && P.conv_w4) launch_forward<Gomoku9, 8, 3, 0, 8, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); else launch_forward<Gomoku9, 8, 3, 1, 8, 1>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); }
#undef FWG
        return;
    }
    // Variant selection (all variants give bitwise the same outputs; measured choices):
    //   large batches  : 2 row tiles per wave = 6 boards per workgroup, rows in the tap-skipping order, weight stream
    //                    scheduled by the compiler (PF = 0)
    //   <= 768 boards  : latency-bound (one workgroup's serial chain): 3 boards per workgroup on a 2 x 2 / 2 x 4 wave grid
    //                    (32 and 96 channels: 1 row tile per wave with the explicit double-buffered weight stream, PF = 1)
    //   <= 320 boards (64 channels) / 512 (128 channels): ONE board per workgroup, each wave one row tile x one column tile
    // AZSP_RESNET_MT / _PF / _CW / _WIDE6 override the choice for tuning.
    static const int mt_env = [] { const char *e = getenv("AZSP_RESNET_MT"); return e ? atoi(e) : 0; }();
    const int mt = mt_env ? mt_env : (max_leaves <= 768 ? 1 : 2);
    static const int pf = [] { const char *e = getenv("AZSP_RESNET_PF"); return e ? atoi(e) : -1; }();
#define FWD(NTV, MTV, PFV) launch_forward<Connect4, NTV, MTV, PFV>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s)
    const bool small = max_leaves <= 768;
    if (n->g.NT == 1) { if (mt == 1) { if (pf == 0) FWD(1, 1, 0); else FWD(1, 1, 1); } else FWD(1, 2, 0); }
    else if (n->g.NT == 2) {
        static const int cw_env = [] { const char *e = getenv("AZSP_RESNET_CW"); return e ? atoi(e) : 0; }();
        // AZSP_RESNET_CW=2 also splits the channels over two waves (8 waves per 3 boards).  Measured: no gain --
        // 0.201 vs 0.189 ms per launch at 100 boards -- the single-workgroup latency floor is set by barriers,
        // epilogues and the weight stream, not by the MFMA chain; kept as a knob, off by default.
        if (mt == 1 && cw_env == 2) { launch_forward<Connect4, 2, 1, 1, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); return; }
        // up to ~one workgroup per CU a launch takes as long as ONE workgroup, and that time grows with the boards it
        // holds: one board per 4-wave workgroup (2 x 2 grid, 42 of 64 rows used) -- 0.112 vs 0.185 ms for 100 boards
        static const int tb1 = [] { const char *e = getenv("AZSP_RESNET_TB1"); return e ? atoi(e) : 320; }();
        if (mt == 1 && tb1 && P.conv_w4 && max_leaves <= tb1) {
            launch_forward<Connect4, 2, 1, 0, 2, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
            return;
        }
        // up to 768 boards: 3 boards per workgroup on the 2 x 2 grid (two row tiles x one column tile per wave, block path):
        // 0.176 vs 0.188 ms for 512 boards with the double-buffered 4 x 1 stream (AZSP_RESNET_MID=0 selects that)
        static const bool mid_old = [] { const char *e = getenv("AZSP_RESNET_MID"); return e && atoi(e) == 0; }();
        if (mt == 1 && P.conv_w4 && !mid_old && mt_env == 0 && pf < 0) { launch_forward<Connect4, 2, 2, 0, 2, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); return; }
        if (mt == 1 && cw_env == 3) { launch_forward<Connect4, 2, 1, 0, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); return; }   // 8 waves, quad-step weights
        // (a 2 x 2 grid with quad-step weight loads was also measured for small batches: 0.200 vs 0.183 ms per 100-board
        //  launch -- the explicit double-buffered stream of the 4 x 1 grid hides more latency where one workgroup owns a CU)
        if (mt == 1) { if (pf == 0 || (pf < 0 && !small)) FWD(2, 1, 0); else FWD(2, 1, 1); }
        else {
            // 2 x 2 wave grid (4 row tiles x 1 column tile per wave, weights 16 B per four k-steps) unless AZSP_RESNET_GRID=4x1
            static const bool grid41 = [] { const char *e = getenv("AZSP_RESNET_GRID"); return e && !strcmp(e, "4x1"); }();
            if (pf == 1) FWD(2, 2, 1);
            else if (grid41 || !P.conv_w4) FWD(2, 2, 0);
            else { NetParams Q = P; Q.perm = P.perm + 256; launch_forward<Connect4, 2, 4, 0, 2, 2>(Q, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s); }
        }
    }
    else if (n->g.NT == 3) { if (pf == 0) FWD(3, 1, 0); else FWD(3, 1, 1); }
    // (splitting the four column tiles of a 128-channel net over four waves for small batches was measured:
    //  2.33 vs 1.90 ms per 100-board launch -- the single-workgroup latency is set by the weight stream, which
    //  gets four times more load instructions per MFMA, not by the MFMA chain)
    else if (n->g.NT == 4) {
        // large batches: 6 boards per workgroup on 8 waves (4 row groups x 2 column groups), one workgroup per CU, so
        // that the tap-skipping row order applies; small batches keep 3 boards per 4-wave workgroup
        static const int wide6 = [] { const char *e = getenv("AZSP_RESNET_WIDE6"); return e ? atoi(e) : 1; }();
        static const bool grid42 = [] { const char *e = getenv("AZSP_RESNET_GRID"); return e && !strcmp(e, "4x2"); }();
        if (wide6 && !small && !grid42 && P.conv_w4) {      // 2 row groups x 4 column waves, 4 row tiles x 1 column tile each
            NetParams Q = P;
            Q.perm = P.perm + 256;
            launch_forward<Connect4, 4, 4, 0, 4, 2>(Q, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
        }
        else if (wide6 && !small) launch_forward<Connect4, 4, 2, 0, 2, 4>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
        else if (small && P.conv_w4 && max_leaves <= (getenv("AZSP_RESNET_TB1") ? atoi(getenv("AZSP_RESNET_TB1")) : 512))   // one board per 8-wave workgroup: 0.86 vs 1.88 ms (9 blocks, 100 boards)
            launch_forward<Connect4, 4, 1, 0, 4, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
        else if (small && P.conv_w4 && !(getenv("AZSP_RESNET_MID") && atoi(getenv("AZSP_RESNET_MID")) == 0))   // 513..768 boards: 3 boards on a 2 x 4 grid, block path (1.20 vs 1.72 ms)
            launch_forward<Connect4, 4, 2, 0, 4, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
        else if (small && P.conv_w4 && getenv("AZSP_RESNET_CW") && atoi(getenv("AZSP_RESNET_CW")) == 4)   // 16 waves per 3 boards, quad-step weights
            launch_forward<Connect4, 4, 1, 0, 4, 4>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
        else if (pf == 0) FWD(4, 1, 0); else FWD(4, 1, 1);
    }
    // 192 / 256 channels: 3 boards (128 rows) per workgroup, 2 row groups x NT/2 column groups of waves
    else if (n->g.NT == 6) launch_forward<Connect4, 6, 2, 1, 3, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
    else launch_forward<Connect4, 8, 2, 1, 4, 2>(P, mine, theirs, leaf_slot, n_leaf, max_leaves, out_p, out_v, s);
#undef FWD
}

}  // namespace azsp
