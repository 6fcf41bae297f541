// Device side of the fused ResNet inference kernel (included by the azsp_resnet_k*.hip translation units, which hold
// the instantiations, and by azsp_resnet.hip, which holds the host side and the variant table).
//
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
//     time).
//   * activations never leave the CU: the layer input lives in LDS as act[row][C + 4] (16-byte aligned rows; the K
//     order of the GEMM is chosen so that a lane's four values for four k-steps are one ds_read_b128, see koff), the
//     residual copy lives in registers in the MFMA C/D layout, and each conv writes its output back over the same LDS
//     buffer after a barrier.  A 3x3 tap is an LDS row offset; taps that fall off the board read a shared all-zero row.
//   * 6-board workgroups take their GEMM rows in an order that groups the cells by the taps they lack (left / right
//     column, top / bottom row, interior), so that a whole 32-row tile can skip its off-board taps: -16.7 % MFMAs,
//     bit-identical results (build_row_order); their tower runs as one static, software-pipelined instruction
//     sequence per row group (tower_layer).
//   * weights (BatchNorm folded on the host at upload) are pre-permuted into B-fragment order; they stream from L2
//     (whole 64-channel tower = 1.2 MB).
//   * heads: policy 1x1 conv on the same MFMA path, value 3x3 conv (3 channels) on v_mfma_f32_4x4x1_16B_f32; their
//     ReLU outputs go to LDS, the two Linear layers are per-board dot products reduced in a fixed order, then masked
//     softmax / tanh.  Every board's result is independent of which tile / row / variant it landed in
//     (batch-invariant, tested bitwise), so leaf order does not influence the search.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>
#include <vector>

#include <hip/hip_ext.h>

#include "azsp_internal.h"
#include "azsp_resnet_host.h"
#include "azsp_tree_dev.h"

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
    size_t o_conv_w16 = 0, o_conv_b16 = 0;   // slot-ordered copies for the Connect-4 one-board kernels on 16-row tiles
    bool has_w16 = false;
    int g_words = 1;              // 64-bit words per side of a position (game trait W)
    int forced_variant = -1;      // index into the variant table if AZSP_RESNET_VARIANT named one at creation, else -1
    double *d_flop_acc = nullptr; // executed matrix-pipe FLOP of every launch since the last reset (both kernels add to it)
};

// FUSED rounds: what the inference kernel needs to carry its games on after evaluating them (advance_game)
struct FusedArgs {
    View v;                      // lo / n_act of the slot group, counters, tables
    int32_t *next_list;          // the leaf list the NEXT launch evaluates (indexed from the group's first slot)
    int32_t *next_count;         // its length counter (zero on entry)
    int32_t *zero_count;         // the counter of the launch after next: zeroed here
};

struct NetParams {
    const float *stem_w, *stem_b, *conv_w, *conv_b, *pol_w, *pol_b, *val_w, *val_w4, *val_b, *fcp_w, *fcp_b, *fcv_w, *fcv_b;
    int conv_w_stride, conv_b_stride, blocks;
    unsigned long long *dbg;      // diagnostic phase timestamps (AZSP_RESNET_TRACE), else nullptr
    const int32_t *perm;          // [256] GEMM row -> board-cell row of the tap-skipping row order (see build_row_order)
    const float *conv_w4;         // conv weights as [tap][s / 4][column tile][lane][4 k-steps] (one 16-byte load per lane
                                  // feeds four k-steps of a wave that owns ONE column tile), else nullptr
    const float *conv_w16;        // Connect-4 one-board kernels (16-row tiles): conv weights as [tap][16-channel input group][16-channel
                                  // output tile][lane][4] in SLOT order (t16_chan), conv_b16: the biases in slot order; else nullptr
    const float *conv_b16;
    double *flop_acc;             // device accumulator of EXECUTED matrix-pipe FLOP (one atomic per launch), or nullptr
    double wg_flops;              // what one workgroup of this variant executes (resnet_exec_flops_per_board x boards per workgroup)
};
// phase timestamp of wave 0 of each workgroup, 100 MHz constant clock; compiled only into the TR = true
// diagnostic instantiation (AZSP_RESNET_TRACE), the product kernels carry no stamp code
#define STAMP(k) do { if constexpr (TR) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + (k)] = wall_clock64(); } } while (0)

// ------------------------------------------------------------------------------------------------
// layout rules shared by host and device
// ------------------------------------------------------------------------------------------------
// K order of every convolution: k-step s of an MFMA (K = 2: lane half h = lane >> 5 supplies k = h) takes input
// channel 8 * (s / 4) + 4 * h + s % 4.  The four values a lane needs for four consecutive k-steps are then 16
// contiguous bytes of an activation row: one ds_read_b128 instead of four ds_read_b32.  koff(s) is the float offset
// from the lane's half base (row * stride + 4 * h); the host permutes the weights to the same order (permute_conv).
__host__ __device__ constexpr int koff(int s) { return 8 * (s >> 2) + (s & 3); }
// activation row stride in floats: a multiple of 4 (16-byte aligned rows) with stride / 4 odd, so that the 16 lanes
// ds_read_b128 services together hit 16 different 16-byte slots when they read consecutive rows
__host__ __device__ constexpr int act_stride(int nt) { return 32 * nt + 4; }

// LDS carve of k_resnet_forward, in floats (one definition for the kernel and for the launch)
template <class G, int NT, int MT, int CW, int RW>
struct LdsLayout {
    static constexpr int WG_ROWS = RW * 32 * MT, TB = WG_ROWS / G::CELLS, STRIDE = act_stride(NT);
    static constexpr bool PART_ALIAS = 32 * NT >= 64;          // FC partial sums inside the activation rows (columns 32..39)
    static constexpr int LOGITS = (WG_ROWS + 1) * STRIDE;      // after act[WG_ROWS + 1][STRIDE] (last row = zeros)
    static constexpr int VL = LOGITS + TB * G::AP;
    static constexpr int HVAL = VL + ((TB + 7) & ~7);
    static constexpr int PERM = HVAL + WG_ROWS * (std::is_same<G, Gomoku9>::value ? 16 : 4);   // (9x9: + the value conv's channel slices)
    static constexpr int PART = PERM + WG_ROWS;
    static constexpr int TOTAL = PART + (PART_ALIAS ? 0 : WG_ROWS * 8);
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// Every convolution puts the WEIGHTS on the A side of v_mfma_f32_32x32x2_f32 and the activations on the B side
// (D = W^T * act^T; both operands have the same lane layout, so nothing else changes): a lane's 16 results of a 32 x 32
// tile then belong to ONE board cell (GEMM row `lane & 31` of the row tile) and 16 output channels -- channel
// (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5) of the column tile in register v -- i.e. four runs of four consecutive
// channels, which an epilogue stores with four 16-byte LDS writes at its own row (no per-value address, no row
// permutation lookup) where the cell-per-register layout needed sixteen 4-byte ones.
__device__ __forceinline__ f32x16 conv_mma(float a_act, float w, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(w, a_act, c, 0, 0, 0);
}
// The same MFMA as conv_mma with the result in registers of its own, as inline assembly: the statically scheduled tower
// starts a tile's accumulators from the block input (second convolution of a block) or from the bias (first) at the
// tile's first MFMA, under a branch on the layer's parity -- written with the builtin on both arms, the optimiser merges
// the two MFMAs behind a 16-register select, which is the copy this form exists to avoid.
__device__ __forceinline__ f32x16 conv_mma_from(float a_act, float w, const f32x16 &c) {
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(w), "v"(a_act), "v"(c));
    return d;
}
// max(x, 0) as ONE instruction (fmaxf compiles to a canonicalising v_max_f32 x, x, x in front of it; the result is the
// same: a NaN input gives 0 either way)
__device__ __forceinline__ float relu(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// the 16 per-channel constants (bias) of a lane for the column tile whose channel 0 is at b
__device__ __forceinline__ f32x16 channel_vec(const float *b, int half) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = *reinterpret_cast<const float4 *>(b + 8 * i + 4 * half);
        r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    }
    return r;
}
// a lane's 16 results to its row: dst = act + row * STRIDE + 4 * (lane >> 5) + first channel of the column tile
__device__ __forceinline__ void store_tile(float *dst, const f32x16 &o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(dst + 8 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

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
        for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt][koff(s0 + i)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = conv_mma(a[mt], w[i][nt], acc[mt][nt]);
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
        for (int mt = 0; mt < MT; ++mt) if ((MASK >> mt) & 1) a[mt] = arow[mt][koff(s)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((MASK >> mt) & 1) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = conv_mma(a[mt], w[nt], acc[mt][nt]);
            }
    }
}

// `need[mt]` (wave-uniform): taps for which at least one row of row tile mt has an on-board input.  The other
// taps multiply an all-zero A tile -- adding exactly 0 -- and are skipped (SK kernels order their rows so that
// whole row tiles share their off-board taps; everywhere else every tap is needed by some row).
template <class G, int MT, int NTW, int TAPS, int S, int PF, int WST = NTW, bool SK = false>
__device__ __forceinline__ void conv_mfma(f32x16 (&acc)[MT][NTW], const float *__restrict__ wl, const float *act,
                                          const int (&ridx)[MT], const uint32_t (&vmask)[MT], int stride, int zero_idx,
                                          int lane, const uint32_t (&need)[MT]) {
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
                for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt][koff(s)];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = conv_mma(a[mt], w[nt], acc[mt][nt]);
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

// ------------------------------------------------------------------------------------------------
// statically scheduled tower layer (6-board Connect-4 workgroups, 2 x CW wave grid, one column tile per wave)
// ------------------------------------------------------------------------------------------------
// Which tile-taps can be skipped is a static property of the row order (build_row_order): row group 0 owns the tiles
// [left column, right column, interior, interior], row group 1 [top row, bottom row, interior, interior]; a tile of
// left-column cells lacks the three taps with dx = -1, and so on (30 of 36 tile-taps remain per layer).  The layer is
// therefore ONE straight-line instruction sequence per row group: a list of steps (tap, chunk of eight weight quads,
// tile, quad), four MFMAs each.  Step i first issues the 16-byte activation read of step i + 2 (three rotating
// buffers: LDS latency never reaches the matrix pipe, also across tile and tap boundaries), then its four MFMAs.
// The weights of a chunk sit in eight 16-byte registers; the chunk's LAST tile reloads each quad for the next chunk
// (the next tap, or the next layer's first) right after it has used it -- eight steps = 2048 matrix-pipe cycles ahead
// of its first use, in place, no second buffer.  __builtin_amdgcn_sched_barrier(0) between steps keeps the compiler
// from re-serialising reads and MFMAs.  Measured (tools/probe/tower_loop.hip, one workgroup per CU): 66.5 cycles per
// v_mfma_f32_32x32x2_f32 against 71.0 for the compiler-scheduled block form with the same register pressure.
struct TStep { int tap, tile, q, last_tile, first; };      // first: the tile's first step of the layer
__host__ __device__ constexpr bool tile_needs(int group, int tile, int tap) {
    if (tile >= 2) return true;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if (group == 0) return tile == 0 ? dx != -1 : dx != 1;      // left column / right column
    return tile == 0 ? dy != -1 : dy != 1;                      // top row / bottom row
}
template <int GROUP, int QPT>          // QPT = weight quads (of four k-steps) per tap = channels / 8
struct TowerSched {
    static constexpr int NSTEP = 30 * QPT;
    TStep s[NSTEP + 2];
    constexpr TowerSched() : s{} {
        int n = 0;
        bool seen[4] = {false, false, false, false};
        for (int tap = 0; tap < 9; ++tap)
            for (int ch = 0; ch < QPT / 8; ++ch) {
                for (int tile = 3; tile >= 0; --tile) {
                    if (!tile_needs(GROUP, tile, tap)) continue;
                    for (int q = 0; q < 8; ++q) {
                        s[n].tap = tap; s[n].tile = tile; s[n].q = ch * 8 + q; s[n].last_tile = 0; s[n].first = seen[tile] ? 0 : 1;
                        seen[tile] = true;
                        ++n;
                    }
                }
                for (int q = 0; q < 8; ++q) s[n - 8 + q].last_tile = 1;
            }
        s[n] = s[n - 1];                   // the two reads past the end repeat the last one
        s[n + 1] = s[n - 1];
    }
};
template <int GROUP, int QPT>
struct TowerHolder { static constexpr TowerSched<GROUP, QPT> SC{}; };

__device__ __forceinline__ float f4c(const float4 &t, int k) { return k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w)); }

template <class G, int STRIDE>
__device__ __forceinline__ const float *tower_abase(const float *act, int ridx_t, uint32_t vmask_t, int tap, int zero_idx) {
    return act + (((vmask_t >> tap) & 1u) ? ridx_t + ((tap / 3 - 1) * G::COLS + (tap % 3 - 1)) * STRIDE : zero_idx);
}

template <class G, int GROUP, int QPT, int NT, int STRIDE, int I>
__device__ __forceinline__ void tower_step(f32x16 (&acc)[4][1], const f32x16 (&xin)[4][1], const f32x16 &bv, bool second, const float *act,
                                           const int (&ridx)[4], const uint32_t (&vmask)[4],
                                           int zero_idx, const float *wq, float4 (&wb)[8], float4 (&xa)[3], const float *&rbase) {
    constexpr TStep st = TowerHolder<GROUP, QPT>::SC.s[I], nx = TowerHolder<GROUP, QPT>::SC.s[I + 2];
    if constexpr ((nx.q & 7) == 0 && I + 2 < TowerSched<GROUP, QPT>::NSTEP)          // step i + 2 opens a new (tap, tile) block
        rbase = tower_abase<G, STRIDE>(act, ridx[nx.tile], vmask[nx.tile], nx.tap, zero_idx);
    xa[(I + 2) % 3] = *reinterpret_cast<const float4 *>(rbase + 8 * nx.q);
    if constexpr (st.first != 0) {         // the tile's accumulators start here: from block input + bias, or from the bias
        if (second) acc[st.tile][0] = conv_mma_from(xa[I % 3].x, wb[st.q & 7].x, xin[st.tile][0]);
        else acc[st.tile][0] = conv_mma(xa[I % 3].x, wb[st.q & 7].x, bv);
    }
#pragma unroll
    for (int k = st.first != 0 ? 1 : 0; k < 4; ++k)
        acc[st.tile][0] = conv_mma(f4c(xa[I % 3], k), f4c(wb[st.q & 7], k), acc[st.tile][0]);
    if constexpr (st.last_tile != 0)       // this quad is done for the chunk: its registers take the next chunk's quad
        wb[st.q & 7] = *reinterpret_cast<const float4 *>(wq + (size_t)(st.tap * QPT + st.q + 8) * NT * 64 * 4);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < TowerSched<GROUP, QPT>::NSTEP)
        tower_step<G, GROUP, QPT, NT, STRIDE, I + 1>(acc, xin, bv, second, act, ridx, vmask, zero_idx, wq, wb, xa, rbase);
}

// One 3x3 convolution of the tower for the row group GROUP.  wq: this lane's pointer to quad 0 of the layer
// ([tap][quad][column tile][lane][4]); wb holds the layer's first chunk on entry and the NEXT layer's first on exit.
// The accumulators need no initialisation: each tile's first MFMA takes its C operand from xin (second = the block's
// second convolution: block input + bias) or from bv (the bias, the same 16 registers for the four tiles).
template <class G, int GROUP, int QPT, int NT, int STRIDE>
__device__ __forceinline__ void tower_layer(f32x16 (&acc)[4][1], const f32x16 (&xin)[4][1], const f32x16 &bv, bool second, const float *act,
                                            const int (&ridx)[4], const uint32_t (&vmask)[4],
                                            int zero_idx, const float *wq, float4 (&wb)[8]) {
    constexpr TStep s0 = TowerHolder<GROUP, QPT>::SC.s[0], s1 = TowerHolder<GROUP, QPT>::SC.s[1];
    static_assert(s0.tap == s1.tap && s0.tile == s1.tile, "the first two steps share a block");
    const float *rbase = tower_abase<G, STRIDE>(act, ridx[s0.tile], vmask[s0.tile], s0.tap, zero_idx);
    float4 xa[3];
    xa[0] = *reinterpret_cast<const float4 *>(rbase + 8 * s0.q);
    xa[1] = *reinterpret_cast<const float4 *>(rbase + 8 * s1.q);
    __builtin_amdgcn_sched_barrier(0);
    tower_step<G, GROUP, QPT, NT, STRIDE, 0>(acc, xin, bv, second, act, ridx, vmask, zero_idx, wq, wb, xa, rbase);
}

// ------------------------------------------------------------------------------------------------
// rolled form of the same pipeline, for every other wave that owns MT row tiles x ONE column tile (no tap skipping)
// ------------------------------------------------------------------------------------------------
// A 3x3 convolution is 9 * QPT / QC chunks of QC weight quads; every chunk runs the same static body of MT * QC steps
// (tile-major), so the loop over chunks stays rolled.  As in tower_step: the 16-byte activation read of step i + 2 is
// issued before the MFMAs of step i (four rotating buffers; MT * QC is a multiple of 4), also across the loop's back
// edge, and the chunk's last tile reloads each weight quad for the next chunk as soon as it has used it.
template <int QPT> struct RolledGeo { static constexpr int QC = QPT % 8 == 0 ? 8 : QPT % 12 == 0 ? 12 : QPT; };

template <class G, int STRIDE>
__device__ __forceinline__ int tower_aindex(int ridx_t, uint32_t vmask_t, int tap, int zero_idx) {       // float index into act
    return ((vmask_t >> tap) & 1u) ? ridx_t + ((tap / 3 - 1) * G::COLS + (tap % 3 - 1)) * STRIDE : zero_idx;
}

template <class G, int MT, int QPT, int NT, int STRIDE>
__device__ __forceinline__ void conv_rolled(f32x16 (&acc)[MT][1], const float *act, const int (&ridx)[MT], const uint32_t (&vmask)[MT],
                                            int zero_idx, const float *wq, float4 (&wb)[RolledGeo<QPT>::QC]) {
    constexpr int QC = RolledGeo<QPT>::QC, CPT = QPT / QC, NCHUNK = 9 * CPT, BODY = MT * QC;
    static_assert(QPT % QC == 0 && BODY % 4 == 0 && BODY >= 4, "chunking");
    float4 xa[4];
    // row bases as float indices into `act` (pointers carried around the loop would lose their LDS address space and
    // turn the reads into flat loads)
    int base[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) base[t] = tower_aindex<G, STRIDE>(ridx[t], vmask[t], 0, zero_idx);
    xa[0] = *reinterpret_cast<const float4 *>(act + base[0]);
    xa[1] = *reinterpret_cast<const float4 *>(act + base[0] + 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int c = 0; c < NCHUNK; ++c) {
        // the next chunk's first tile: where the last two steps of this chunk prefetch from
        const int cn = c + 1 < NCHUNK ? c + 1 : c;
        const int tapn = cn / CPT, q0n = (cn % CPT) * QC;
        const int nbase0 = tower_aindex<G, STRIDE>(ridx[0], vmask[0], tapn, zero_idx) + 8 * q0n;
        const int q0 = (c % CPT) * QC;
        const float *wnext = wq + (size_t)(c * QC + QC) * NT * 64 * 4;          // quad 0 of the next chunk
#pragma unroll
        for (int i = 0; i < BODY; ++i) {
            const int t = i / QC, q = i % QC;
            const int j = i + 2;                                                  // the step whose activations are requested now
            if (j < BODY) xa[j % 4] = *reinterpret_cast<const float4 *>(act + base[j / QC] + 8 * (q0 + j % QC));
            else xa[j % 4] = *reinterpret_cast<const float4 *>(act + nbase0 + 8 * (j - BODY));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc[t][0] = conv_mma(f4c(xa[i % 4], k), f4c(wb[q], k), acc[t][0]);
            if (t == MT - 1) wb[q] = *reinterpret_cast<const float4 *>(wnext + (size_t)q * NT * 64 * 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cn != c && cn % CPT == 0) {                                           // the next chunk opens a new tap
#pragma unroll
            for (int t = 0; t < MT; ++t) base[t] = tower_aindex<G, STRIDE>(ridx[t], vmask[t], tapn, zero_idx);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 16-row tiles for one 9x9 board per workgroup (Gomoku, 96 channels and more)
// ------------------------------------------------------------------------------------------------
// 81 cells in 32-row tiles are 96 GEMM rows: 15 of them padding, 16 % of the matrix-pipe time.  v_mfma_f32_16x16x4_f32 has
// the same FLOP rate as the 32x32x2 form (32 cycles, tools/probe/mfma16x16.hip), so the tower runs on FIVE 16-row tiles
// (cells 0..79) and the last cell (the bottom-right corner: four of its nine taps exist) on the vector ALU of the same
// waves -- 8 FMAs beside 40 MFMAs per 16 input channels, with the weights the MFMAs already hold:
//   A side: weights, lane l = output channel (l & 15) of a 16-channel tile, input channels 4 * (l >> 4) + j of the 16-group
//   B side: activations, lane l = cell (l & 15) of the row tile, the same input channels (one 16-byte LDS read = 4 MFMAs)
//   D: register j of lane l = output channel 4 * (l >> 4) + j of the tile, cell l & 15  -> one 16-byte store per tile
// A wave owns 32 output channels (two 16-channel tiles) x all five row tiles; the corner cell's lane (oc, kq) sums the
// products of its quarter of the input channels, the four quarters are added across lanes at the end of the layer.
// Same rolled software pipeline as conv_rolled: chunks of QC 16-channel groups, a body of 6 * QC steps (tile-major, the
// corner cell as a sixth "tile"), activations two steps ahead, the chunk's last user reloads each weight pair in place.
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4m conv_mma16(float a_act, float w, const f32x4m &c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(w, a_act, c, 0, 0, 0);
}
template <int SPT> struct Tile16Geo { static constexpr int QC = SPT % 8 == 0 ? 8 : (SPT % 6 == 0 ? 6 : SPT); };
constexpr uint32_t G9_CORNER_TAPS = 0x1Bu;            // taps (dy, dx) in {-1, 0}^2 exist for the cell (8, 8): bits 0, 1, 3, 4

template <class G, int SPT, int NT, int STRIDE>
__device__ __forceinline__ void conv_tile16(f32x4m (&acc)[5][2], float (&vacc)[2], const float *act, const int (&ridx)[5],
                                            const uint32_t (&vmask)[5], int r80, int zero_idx, const float *wq,
                                            float4 (&wb)[Tile16Geo<SPT>::QC][2]) {
    constexpr int QC = Tile16Geo<SPT>::QC, CPT = SPT / QC, NCHUNK = 9 * CPT, BODY = 6 * QC;
    constexpr size_t QSTR = (size_t)2 * NT * 64 * 4;                  // floats between consecutive 16-channel groups
    static_assert(SPT % QC == 0 && BODY % 4 == 0, "chunking");
    float4 xa[4];
    int base[6];
#pragma unroll
    for (int t = 0; t < 5; ++t) base[t] = tower_aindex<G, STRIDE>(ridx[t], vmask[t], 0, zero_idx);
    base[5] = tower_aindex<G, STRIDE>(r80, G9_CORNER_TAPS, 0, zero_idx);
    xa[0] = *reinterpret_cast<const float4 *>(act + base[0]);
    xa[1] = *reinterpret_cast<const float4 *>(act + base[0] + 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int c = 0; c < NCHUNK; ++c) {
        const int cn = c + 1 < NCHUNK ? c + 1 : c;
        const int tap = c / CPT, tapn = cn / CPT, q0n = (cn % CPT) * QC;
        const int nbase0 = tower_aindex<G, STRIDE>(ridx[0], vmask[0], tapn, zero_idx) + 16 * q0n;
        const int q0 = (c % CPT) * QC;
        const bool corner = (G9_CORNER_TAPS >> tap) & 1u;             // this tap exists for the corner cell
        const float *wnext = wq + (size_t)(c * QC + QC) * QSTR;       // group 0 of the next chunk
#pragma unroll
        for (int i = 0; i < BODY; ++i) {
            const int t = i / QC, q = i % QC;
            const int j = i + 2;                                      // the step whose activations are requested now
            if (j < BODY) xa[j % 4] = *reinterpret_cast<const float4 *>(act + base[j / QC] + 16 * (q0 + j % QC));
            else xa[j % 4] = *reinterpret_cast<const float4 *>(act + nbase0 + 16 * (j - BODY));
            if (t < 5) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[t][0] = conv_mma16(f4c(xa[i % 4], k), f4c(wb[q][0], k), acc[t][0]);
                    acc[t][1] = conv_mma16(f4c(xa[i % 4], k), f4c(wb[q][1], k), acc[t][1]);
                }
            } else {
                if (corner) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        vacc[0] = fmaf(f4c(xa[i % 4], k), f4c(wb[q][0], k), vacc[0]);
                        vacc[1] = fmaf(f4c(xa[i % 4], k), f4c(wb[q][1], k), vacc[1]);
                    }
                }
                wb[q][0] = *reinterpret_cast<const float4 *>(wnext + (size_t)q * QSTR);
                wb[q][1] = *reinterpret_cast<const float4 *>(wnext + (size_t)q * QSTR + 256);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cn != c && cn % CPT == 0) {                               // the next chunk opens a new tap
#pragma unroll
            for (int t = 0; t < 5; ++t) base[t] = tower_aindex<G, STRIDE>(ridx[t], vmask[t], tapn, zero_idx);
            base[5] = tower_aindex<G, STRIDE>(r80, G9_CORNER_TAPS, tapn, zero_idx);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 16-row tiles for ONE Connect-4 board per workgroup (small batches: a round there is this workgroup's latency)
// ------------------------------------------------------------------------------------------------
// 42 cells in 32-row tiles are 64 GEMM rows; in 16-row tiles 48: a quarter less matrix-pipe time per layer on the one CU
// the board runs on.  The variants of a network must return the SAME BITS (the engine picks them by batch size), and an
// fp32 MFMA of either shape is a chain of fused multiply-adds in k order (tools/probe/mfma_order.hip: 0 of 51 200 results
// differ), so the 16x16x4 form only has to meet the channel order of the 32-row kernels: per 8-channel block c, c+4, c+1,
// c+5, c+2, c+6, c+3, c+7 (koff).  MFMA j of a 16-channel group then takes the channels  j = 0: 0 4 1 5,  1: 2 6 3 7,
// 2: 8 12 9 13,  3: 10 14 11 15  -- lane quarter q supplies t16_chan(q, j).  Inside this kernel's tower the activations sit
// in LDS in SLOT order (slot 4 q + j of a 16-group holds channel t16_chan(q, j)), so that a quarter's four values are one
// 16-byte read; the weights' A rows are permuted the same way, which makes a lane's four results four consecutive slots
// (one 16-byte store).  One pass converts the stem's output to slot order, one converts the tower's output back.
__host__ __device__ constexpr int t16_chan(int q, int j) { return 4 * (q & 1) + (q >> 1) + 2 * (j & 1) + 8 * (j >> 1); }

// ------------------------------------------------------------------------------------------------
// The static schedule of the Connect-4 16-row-tile kernels: ONE board per workgroup (three tiles, every tap), and THREE
// boards per workgroup with static tap skipping (mid-size batches, and the workhorse of the evaluation cache, whose
// launches hold a few hundred positions)
// ------------------------------------------------------------------------------------------------
// In 32-row tiles the 126 cells of three boards cannot skip taps (no class of cells that lack the same taps fills a tile);
// in 16-row tiles they can: tile 0 = 16 left-column cells (lack dx = -1: 3 of 9 taps), 1 = 16 right-column cells, 2 = the 15
// inner cells of the top rows + a top-left corner, 3 = the same at the bottom, 4..7 = interior cells, the two right-column
// cells left over and two padding rows (build_row_order's third table).  60 of 72 tile-taps remain (the six-board kernels'
// 5/6).  Which tile-taps exist is static, so a layer is ONE straight-line sequence of 60 x SPT steps (tile-tap x 16-channel
// group, four v_mfma_f32_16x16x4_f32 each; t16_layer) as in tower_layer: step i issues the 16-byte activation read of step i + 2 (four
// rotating buffers) and its own MFMAs; at 64 channels a tap's first tile loads the NEXT tap's weight quads into the other of
// two buffers, at 128 the tap's last tile reloads each quad of the one buffer after its last use.  The one-board kernel
// runs the same sequence over its three tiles without skipping (SK = false).  Same bits as every other variant.
struct T16Step { int tap, tile, q, first, last; };
__host__ __device__ constexpr bool t16m_needs(int tile, int tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    return tile == 0 ? dx != -1 : (tile == 1 ? dx != 1 : (tile == 2 ? dy != -1 : (tile == 3 ? dy != 1 : true)));
}
// NRT row tiles; SK: the three-board tile classes above (else every tile takes every tap: the one-board kernel's three tiles)
template <int NRT, bool SK, int SPT>
struct T16Sched {
    static constexpr int NSTEP = (SK ? 60 : 9 * NRT) * SPT;
    T16Step s[NSTEP + 2];
    constexpr T16Sched() : s{} {
        int n = 0;
        for (int tap = 0; tap < 9; ++tap) {
            int first_tile = -1, last_tile = -1;
            for (int tile = NRT - 1; tile >= 0; --tile) if (!SK || t16m_needs(tile, tap)) { first_tile = tile; if (last_tile < 0) last_tile = tile; }
            for (int tile = 0; tile < NRT; ++tile) {
                if (SK && !t16m_needs(tile, tap)) continue;
                for (int q = 0; q < SPT; ++q) { s[n].tap = tap; s[n].tile = tile; s[n].q = q; s[n].first = tile == first_tile; s[n].last = tile == last_tile; ++n; }
            }
        }
        s[n] = s[n - 1]; s[n + 1] = s[n - 1];          // the reads past the end repeat the last one
    }
};
template <int NRT, bool SK, int SPT>
struct T16Holder { static constexpr T16Sched<NRT, SK, SPT> SC{}; };

template <class G, int NRT, bool SK, int SPT, int NT, int STRIDE, int I>
__device__ __forceinline__ void t16_step(f32x4m (&acc)[NRT], const float *act, const int (&rix)[NRT], const uint32_t (&vmk)[NRT], int zero_idx,
                                         const float *wq, float4 (&wb)[2][SPT], float4 (&xa)[4], int &rbase) {
    constexpr T16Step st = T16Holder<NRT, SK, SPT>::SC.s[I], nx = T16Holder<NRT, SK, SPT>::SC.s[I + 2];
    constexpr bool DB = SPT <= 4;                                     // 64 channels: two weight buffers, taps alternate (at 128 the second costs more than it hides)
    constexpr int cur = DB ? (st.tap & 1) : 0;
    constexpr size_t QSTR = (size_t)2 * NT * 64 * 4;                  // floats between consecutive 16-channel input groups
    if constexpr (nx.q == 0 && I + 2 < T16Sched<NRT, SK, SPT>::NSTEP) // step i + 2 opens a new (tile, tap) block
        rbase = tower_aindex<G, STRIDE>(rix[nx.tile], vmk[nx.tile], nx.tap, zero_idx);
    xa[(I + 2) % 4] = *reinterpret_cast<const float4 *>(act + rbase + 16 * nx.q);
    if constexpr (DB && st.first != 0)                                // the NEXT tap's (past the layer's end: the next layer's) group q, a whole
        wb[cur ^ 1][st.q] = *reinterpret_cast<const float4 *>(wq + (size_t)((st.tap + 1) * SPT + st.q) * QSTR);      // tap of MFMAs ahead of its use
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[st.tile] = conv_mma16(f4c(xa[I % 4], k), f4c(wb[cur][st.q], k), acc[st.tile]);
    if constexpr (!DB && st.last != 0)                                // one buffer: the tap's last tile reloads each quad right after using it
        wb[0][st.q] = *reinterpret_cast<const float4 *>(wq + (size_t)((st.tap + 1) * SPT + st.q) * QSTR);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < T16Sched<NRT, SK, SPT>::NSTEP) t16_step<G, NRT, SK, SPT, NT, STRIDE, I + 1>(acc, act, rix, vmk, zero_idx, wq, wb, xa, rbase);
}
template <class G, int NRT, bool SK, int SPT, int NT, int STRIDE>
__device__ __forceinline__ void t16_layer(f32x4m (&acc)[NRT], const float *act, const int (&rix)[NRT], const uint32_t (&vmk)[NRT], int zero_idx,
                                          const float *wq, float4 (&wb)[2][SPT]) {
    constexpr T16Step s0 = T16Holder<NRT, SK, SPT>::SC.s[0], s1 = T16Holder<NRT, SK, SPT>::SC.s[1];
    float4 xa[4];
    int rbase = tower_aindex<G, STRIDE>(rix[s0.tile], vmk[s0.tile], s0.tap, zero_idx);
    xa[0] = *reinterpret_cast<const float4 *>(act + rbase + 16 * s0.q);
    if constexpr (s1.q == 0) rbase = tower_aindex<G, STRIDE>(rix[s1.tile], vmk[s1.tile], s1.tap, zero_idx);
    xa[1] = *reinterpret_cast<const float4 *>(act + rbase + 16 * s1.q);
    __builtin_amdgcn_sched_barrier(0);
    t16_step<G, NRT, SK, SPT, NT, STRIDE, 0>(acc, act, rix, vmk, zero_idx, wq, wb, xa, rbase);
    if constexpr (SPT <= 4)
#pragma unroll
        for (int q = 0; q < SPT; ++q) wb[0][q] = wb[1][q];            // tap 8 used buffer 0 and filled buffer 1 with the next layer's tap 0
}

// (C/D layout of v_mfma_f32_32x32x2_f32: register v of lane l holds row (v & 3) + 8 * (v >> 2) + 4 * (l >> 5), column
// l & 31 -- with the weights on the A side the row is the output channel and the column the board cell: conv_mma above)

template <class G, int NT, int MT, int PF, int CW = 1, int RW = 4, bool TR = false, bool FUSED = false>
__global__ __launch_bounds__(64 * RW * CW, (RW * CW == 4 && CW == 1 && MT == 1 && NT <= 2 ? 3 : (RW * CW == 16 ? 1 : (RW * CW > 8 ? RW * CW / 4 : (RW * CW == 8 && RW * MT == 8 ? 1 : 2))))) void k_resnet_forward(NetParams P, const u64 *__restrict__ mine, const u64 *__restrict__ theirs, int bstride,
                                                         const int32_t *__restrict__ leaf_slot, const int32_t *__restrict__ n_leaf_p,
                                                         float *__restrict__ out_p, float *__restrict__ out_v, int p_stride, int v_stride,
                                                         typename std::conditional<FUSED, FusedArgs, int>::type F) {
    constexpr int CELLS = G::CELLS, COLS = G::COLS, ROWS = G::ROWS, A = G::A;
    constexpr int WG_ROWS = RW * 32 * MT;          // RW row groups x MT row tiles of 32 (x CW column groups of waves)
    constexpr int WROWS = 32 * MT;                 // rows per wave
    constexpr int NTHREADS = 64 * RW * CW;
    constexpr int TB = WG_ROWS / CELLS;            // boards per workgroup
    constexpr int CP = 32 * NT, S = CP / 2, STRIDE = act_stride(NT);
    constexpr int NTW = NT / CW;                   // column tiles per wave: CW waves share a row group
    static_assert(NT % CW == 0, "column tiles must divide over the column waves");
    constexpr int ZERO_IDX = WG_ROWS * STRIDE;
    // tap skipping: 6 Connect-4 boards per workgroup in the row order of build_row_order()
    constexpr bool SK = std::is_same<G, Connect4>::value && PF == 0 && ((MT == 2 && RW == 4) || (MT == 4 && RW == 2 && CW == NT));
    // the statically scheduled tower (tower_layer): 6-board workgroups on the 2 x CW wave grid, 64 or 128 channels
    constexpr bool STATIC_TOWER = SK && MT == 4 && RW == 2 && CW == NT && (NT == 2 || NT == 4);
    // every other wave grid with one column tile per wave: the rolled form of the same software pipeline (conv_rolled)
    // one 9x9 board per workgroup from 96 channels on: the tower on five 16-row tiles + the corner cell (conv_tile16)
    constexpr bool TILE16 = std::is_same<G, Gomoku9>::value && PF == 0 && CW == NT && RW == 1 && MT == 3 && NT >= 3;
    // one Connect-4 board per workgroup on three 16-row tiles (t16_layer, SK = false): PF == 2 marks these instantiations; every
    // wave owns one 16-channel output tile
    constexpr bool TILE16C = std::is_same<G, Connect4>::value && PF == 2;
    static_assert(!TILE16C || (MT == 1 && WG_ROWS / CELLS == 1 && RW * CW * 16 == 32 * NT), "one board, one 16-channel tile per wave");
    // three Connect-4 boards per workgroup on eight 16-row tiles with static tap skipping (t16m_layer): PF == 3
    constexpr bool TILE16M = std::is_same<G, Connect4>::value && PF == 3;
    static_assert(!TILE16M || (MT == 2 && WG_ROWS == 128 && RW * CW * 16 == 32 * NT), "three boards, one 16-channel tile per wave");
    constexpr int PFE = PF >= 2 ? 0 : PF;          // (the prefetch form of the other convolutions)
    constexpr bool ROLLED = !STATIC_TOWER && !TILE16 && !TILE16C && !TILE16M && PF == 0 && CW == NT && NT >= 2;
    constexpr int WBN = ROLLED ? RolledGeo<S / 4>::QC : 8;
    constexpr int SPT16 = CP / 16, QC16 = Tile16Geo<SPT16>::QC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;                             // [WG_ROWS + 1][STRIDE], 16-byte aligned rows
    constexpr int AP = G::AP;                      // actions padded to a multiple of 8
    using L = LdsLayout<G, NT, MT, CW, RW>;
    float *logits = smem + L::LOGITS;              // [TB][AP]
    float *vl = smem + L::VL;                      // [TB] value logits (wide action spaces)
    float *hval = smem + L::HVAL;                  // [WG_ROWS][4] value-head activations
    int *s_perm = reinterpret_cast<int *>(smem + L::PERM);   // [WG_ROWS] GEMM row -> board-cell row (SK only)
    // FC partial sums [WG_ROWS][8]: from 64 channels on they live in columns 32..39 of the (then dead) activation rows
    float *part = L::PART_ALIAS ? act + 32 : smem + L::PART;
    constexpr int PSTR = L::PART_ALIAS ? STRIDE : 8;

    if constexpr (FUSED) { if (blockIdx.x == 0 && threadIdx.x == 0) F.zero_count[0] = 0; }
    const int n_leaf = *n_leaf_p;
    const int board0 = blockIdx.x * TB;
    if (board0 >= n_leaf) return;
    if constexpr (!FUSED) {
        // launch gate (FwdArgs::gate): with the evaluation cache the list's length is only known on the device and varies
        // from 1 to the group's size, so the host enqueues every variant of the network and each runs only for lengths in
        // its own range (lo, hi], in units of 16 boards; 0 = no gate
        const int lo = (F & 0xffff) * 16, hi = (int)((unsigned)F >> 16) * 16;
        if (n_leaf <= lo || (hi != 0 && n_leaf > hi)) return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.flop_acc != nullptr)       // the workgroups that get past the two returns above
        atomicAdd(P.flop_acc, (double)((n_leaf + TB - 1) / TB) * P.wg_flops);
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
        ridx[mt] = r * STRIDE + 4 * half;
        ry[mt] = y;
        rx[mt] = x;
        bm[mt] = board_zero<G::W>();
        bt[mt] = board_zero<G::W>();
        if (real) {
            const int slot = leaf_slot[board0 + b];
            bm[mt] = board_load_strided<G::W>(mine, slot, bstride);
            bt[mt] = board_load_strided<G::W>(theirs, slot, bstride);
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
            const f32x16 bias = channel_vec(P.stem_b + col0 + nt * 32, half);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bias;
        }
        // the 27 binary inputs of a row (K index kk = tap * 3 + plane: empty / mine / theirs at the tapped cell), once per tile
        uint32_t inbits[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            uint32_t ib = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if ((vmask[mt] >> tap) & 1u) {
                    const int bit = G::bit_of(ry[mt] + tap / 3 - 1, rx[mt] + tap % 3 - 1);
                    const uint32_t mi = board_bit(bm[mt], bit) ? 1u : 0u, th = board_bit(bt[mt], bit) ? 1u : 0u;
                    ib |= (((1u - mi) * (1u - th)) | (mi << 1) | (th << 2)) << (3 * tap);
                }
            }
            inbits[mt] = ib;
        }
        for (int s = 0; s < 14; ++s) {
            const int kk = 2 * s + half;               // 27 real + 1 zero (bit 27 is never set)
            float w[NTW];
            load_w<NTW>(P.stem_w + ((size_t)s * 64 + lane) * NT + wc * NTW, w);
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ((inbits[mt] >> kk) & 1u) ? 1.0f : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = conv_mma(a[mt], w[nt], acc[mt][nt]);
        }
        // xres = the block input PLUS the bias of the block's second convolution, whose accumulators start from it
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const f32x16 b2 = channel_vec(P.conv_b + (size_t)P.conv_b_stride + col0 + nt * 32, half);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 o;
#pragma unroll
                for (int v = 0; v < 16; ++v) o[v] = relu(acc[mt][nt][v]);
                store_tile(act + ridx[mt] + col0 + nt * 32, o);
                xres[mt][nt] = o + b2;
            }
        }
    }
    __syncthreads();
    STAMP(2);
    if constexpr (TR) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + 56] = __builtin_amdgcn_s_memtime(); }   // core-clock counter

    // ---- residual blocks (ResBlock.forward resnet.py:21-27) -----------------------------------------
    f32x16 bv[NTW];                                // the bias of the next layer, fetched one layer ahead
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) bv[nt] = channel_vec(P.conv_b + col0 + nt * 32, half);
    // TILE16 state: lane l = cell (l & 15) of a 16-row tile / output channel (l & 15) of a 16-channel tile, quarter kq
    const int kq = lane >> 4, l15 = lane & 15;
    int ridx16[5];
    uint32_t vmask16[5];
    f32x4m acc16[5][2], xres16[5][2];
    float vacc[2], xresv[2];                       // the corner cell (80): this lane's partial sum, block input + bias
    float4 wb16[TILE16 ? QC16 : 1][2];
    constexpr int R80 = 80 * STRIDE;
    if constexpr (TILE16) {
        static_assert(TB == 1 && CELLS == 81, "one 9x9 board per workgroup");
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int pos = 16 * t + l15, y = pos / COLS, x = pos % COLS;
            uint32_t m = 0;
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                if (yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS) m |= 1u << tp;
            }
            vmask16[t] = m;
            ridx16[t] = pos * STRIDE + 4 * kq;
        }
        // block input (+ the bias of the block's second convolution) in the tower's register layout, from what the stem stored
        const float *b2 = P.conv_b + (size_t)P.conv_b_stride + col0;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float4 bq = *reinterpret_cast<const float4 *>(b2 + 16 * ct + 4 * kq);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const float4 x = *reinterpret_cast<const float4 *>(act + ridx16[t] + col0 + 16 * ct);
                xres16[t][ct] = (f32x4m){x.x + bq.x, x.y + bq.y, x.z + bq.z, x.w + bq.w};
            }
            xresv[ct] = act[R80 + col0 + 16 * ct + l15] + b2[16 * ct + l15];
        }
        const float *wq0 = P.conv_w4 + ((size_t)(2 * wc) * 64 + lane) * 4;
#pragma unroll
        for (int j = 0; j < QC16; ++j) {
            wb16[j][0] = *reinterpret_cast<const float4 *>(wq0 + (size_t)j * 2 * NT * 64 * 4);
            wb16[j][1] = *reinterpret_cast<const float4 *>(wq0 + (size_t)j * 2 * NT * 64 * 4 + 256);
        }
    }
    float4 wb[WBN];                                // STATIC_TOWER / ROLLED: the current chunk of weight quads (tower_layer, conv_rolled)
    if constexpr (STATIC_TOWER || ROLLED) {
        const float *wq0 = P.conv_w4 + ((size_t)wc * 64 + lane) * 4;
#pragma unroll
        for (int j = 0; j < WBN; ++j) wb[j] = *reinterpret_cast<const float4 *>(wq0 + (size_t)j * NT * 64 * 4);
    }
    if constexpr (TILE16C || TILE16M) {
        // ---- the whole tower on 16-row tiles, activations in slot order: ONE board = three tiles (cells 0..47, 42 real;
        //      t16_layer) or THREE boards = eight tiles in the tap-skipping order of P.perm (t16m_layer) ----
        constexpr int NRT = TILE16C ? 3 : 8, TROWS = 16 * NRT;
        constexpr int CP16 = CP / 4, ITEMS = TROWS * CP16, PER = (ITEMS + NTHREADS - 1) / NTHREADS;
        const int nl = 2 * P.blocks;
        if (nl > 0) {
            float4 mv[PER];
#pragma unroll
            for (int it = 0; it < PER; ++it) {          // natural -> slot order: slots 4 q .. 4 q + 3 of a 16-group <- channels t16_chan(q, 0..3)
                const int item = tid + it * NTHREADS, row = item / CP16, g4 = item % CP16, q = g4 & 3;
                const float *src = act + row * STRIDE + 16 * (g4 >> 2);
                if (item < ITEMS) mv[it] = make_float4(src[t16_chan(q, 0)], src[t16_chan(q, 1)], src[t16_chan(q, 2)], src[t16_chan(q, 3)]);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int item = tid + it * NTHREADS, row = item / CP16, g4 = item % CP16;
                if (item < ITEMS) *reinterpret_cast<float4 *>(act + row * STRIDE + 4 * g4) = mv[it];
            }
            __syncthreads();
            const int c16 = 16 * (tid >> 6);            // this wave's 16-channel output tile (slots c16 .. c16 + 15)
            int rix[NRT];
            uint32_t vmk[NRT];
            f32x4m a16[NRT], xr16[NRT];
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                const int r = TILE16C ? 16 * t + l15 : P.perm[16 * t + l15];        // board-cell row of this lane's cell
                const int b = r / CELLS, pos = r % CELLS, y = pos / COLS, x = pos % COLS;
                uint32_t m = 0;
                if (r < TB * CELLS && board0 + b < n_leaf)
                    for (int tp = 0; tp < 9; ++tp) {
                        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                        if (yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS) m |= 1u << tp;
                    }
                vmk[t] = m;
                rix[t] = r * STRIDE + 4 * kq;
            }
            {   // block input + the bias of the block's second convolution, from what the stem stored
                const float4 bq = *reinterpret_cast<const float4 *>(P.conv_b16 + (size_t)P.conv_b_stride + c16 + 4 * kq);
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    const float4 x = *reinterpret_cast<const float4 *>(act + rix[t] + c16);
                    xr16[t] = (f32x4m){x.x + bq.x, x.y + bq.y, x.z + bq.z, x.w + bq.w};
                }
            }
            constexpr int NWB = SPT16;                        // weight quads in registers: a whole tap
            float4 wbc[2][NWB];
            const float *wq0 = P.conv_w16 + ((size_t)(tid >> 6) * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < NWB; ++j) wbc[0][j] = *reinterpret_cast<const float4 *>(wq0 + (size_t)j * 2 * NT * 64 * 4);
            for (int layer = 0; layer < nl; ++layer) {
                const bool second = layer & 1;
                const float *blc = P.conv_b16 + (size_t)layer * P.conv_b_stride + c16;
                const bool fold = second && layer + 2 < nl;        // a next block exists: keep its input + bias
                if (second) {
#pragma unroll
                    for (int t = 0; t < NRT; ++t) a16[t] = xr16[t];
                } else {
                    const float4 bq = *reinterpret_cast<const float4 *>(blc + 4 * kq);
#pragma unroll
                    for (int t = 0; t < NRT; ++t) a16[t] = (f32x4m){bq.x, bq.y, bq.z, bq.w};
                }
                t16_layer<G, NRT, TILE16M, SPT16, NT, STRIDE>(a16, act, rix, vmk, ZERO_IDX + 4 * kq, wq0 + (size_t)layer * P.conv_w_stride, wbc);
                if (layer < 16) STAMP(3 + 3 * layer);
                float4 b2q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (fold) b2q = *reinterpret_cast<const float4 *>(blc + 2 * (size_t)P.conv_b_stride + 4 * kq);
                __syncthreads();                           // every wave is done reading the layer input
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    const float4 o = make_float4(relu(a16[t][0]), relu(a16[t][1]), relu(a16[t][2]), relu(a16[t][3]));
                    *reinterpret_cast<float4 *>(act + rix[t] + c16) = o;
                    if (fold) xr16[t] = (f32x4m){o.x + b2q.x, o.y + b2q.y, o.z + b2q.z, o.w + b2q.w};
                }
                __syncthreads();
                if (layer < 16) STAMP(5 + 3 * layer);
            }
#pragma unroll
            for (int it = 0; it < PER; ++it) {          // slot -> natural order for the heads
                const int item = tid + it * NTHREADS, row = item / CP16, g4 = item % CP16;
                if (item < ITEMS) mv[it] = *reinterpret_cast<const float4 *>(act + row * STRIDE + 4 * g4);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int item = tid + it * NTHREADS, row = item / CP16, g4 = item % CP16, q = g4 & 3;
                float *dst = act + row * STRIDE + 16 * (g4 >> 2);
                if (item < ITEMS) { dst[t16_chan(q, 0)] = mv[it].x; dst[t16_chan(q, 1)] = mv[it].y; dst[t16_chan(q, 2)] = mv[it].z; dst[t16_chan(q, 3)] = mv[it].w; }
            }
            __syncthreads();
        }
    } else
    for (int layer = 0; layer < 2 * P.blocks; ++layer) {
        const bool second = layer & 1;
        const float *wl = P.conv_w + (size_t)layer * P.conv_w_stride;
        const float *bl = P.conv_b + (size_t)layer * P.conv_b_stride;
        if constexpr (TILE16) {
            const float *blc = bl + col0;
            const bool fold = second && layer + 2 < 2 * P.blocks;      // a next block exists: keep its input + bias
            // the accumulators start from block input + bias (second convolution of a block) or from the bias; the
            // corner cell's sum is split over the four lanes (oc, kq): quarter 0 takes the start value
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if (second) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) acc16[t][ct] = xres16[t][ct];
                    vacc[ct] = kq == 0 ? xresv[ct] : 0.0f;
                } else {
                    const float4 bq = *reinterpret_cast<const float4 *>(blc + 16 * ct + 4 * kq);
#pragma unroll
                    for (int t = 0; t < 5; ++t) acc16[t][ct] = (f32x4m){bq.x, bq.y, bq.z, bq.w};
                    vacc[ct] = kq == 0 ? blc[16 * ct + l15] : 0.0f;
                }
            }
            const float *wq = P.conv_w4 + (size_t)layer * P.conv_w_stride + ((size_t)(2 * wc) * 64 + lane) * 4;
            conv_tile16<G, SPT16, NT, STRIDE>(acc16, vacc, act, ridx16, vmask16, R80 + 4 * kq, ZERO_IDX + 4 * kq, wq, wb16);
            if (layer < 16) STAMP(3 + 3 * layer);
            float4 b2q[2];
            float b2s[2];
            if (fold) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    b2q[ct] = *reinterpret_cast<const float4 *>(blc + 2 * (size_t)P.conv_b_stride + 16 * ct + 4 * kq);
                    b2s[ct] = blc[2 * (size_t)P.conv_b_stride + 16 * ct + l15];
                }
            }
            __syncthreads();                           // every wave is done reading the layer input
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const float4 o = make_float4(relu(acc16[t][ct][0]), relu(acc16[t][ct][1]), relu(acc16[t][ct][2]), relu(acc16[t][ct][3]));
                    *reinterpret_cast<float4 *>(act + ridx16[t] + col0 + 16 * ct) = o;
                    if (fold) xres16[t][ct] = (f32x4m){o.x + b2q[ct].x, o.y + b2q[ct].y, o.z + b2q[ct].z, o.w + b2q[ct].w};
                }
                float v = vacc[ct];                    // the corner cell: the four quarters of the input channels
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                v = relu(v);
                if (kq == 0) act[R80 + col0 + 16 * ct + l15] = v;
                if (fold) xresv[ct] = v + b2s[ct];
            }
            __syncthreads();
            if (layer < 16) STAMP(5 + 3 * layer);
            continue;
        }
        // Biases are fetched ahead where a wave owns ONE column tile (16 registers): the next layer's while this layer's
        // last MFMAs drain (bv: what a block's first convolution starts from), and -- except in the statically scheduled
        // kernels, which have no registers to spare across the barrier -- the bias after that, which is folded into the
        // block input kept in xres.
        constexpr bool AHEAD = NTW == 1;
        const int nl = 2 * P.blocks;
        if constexpr (!STATIC_TOWER) {
            if (second) {                              // x += residual: the accumulators start from block input + bias
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = xres[mt][nt];
            } else {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if constexpr (!AHEAD) bv[nt] = channel_vec(bl + col0 + nt * 32, half);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bv[nt];
                }
            }
        }
        if constexpr (STATIC_TOWER) {
            const float *wq = P.conv_w4 + (size_t)layer * P.conv_w_stride + ((size_t)wc * 64 + lane) * 4;
            if (wave == 0) tower_layer<G, 0, S / 4, NT, STRIDE>(acc, xres, bv[0], second, act, ridx, vmask, ZERO_IDX + 4 * half, wq, wb);
            else tower_layer<G, 1, S / 4, NT, STRIDE>(acc, xres, bv[0], second, act, ridx, vmask, ZERO_IDX + 4 * half, wq, wb);
        } else if constexpr (ROLLED) {
            const float *wq = P.conv_w4 + (size_t)layer * P.conv_w_stride + ((size_t)wc * 64 + lane) * 4;
            conv_rolled<G, MT, S / 4, NT, STRIDE>(acc, act, ridx, vmask, ZERO_IDX + 4 * half, wq, wb);
        } else {
            conv_mfma<G, MT, NTW, 9, S, PFE, NT, SK>(acc, wl + wc * NTW, act, ridx, vmask, STRIDE, ZERO_IDX + 4 * half, lane, need);
        }
        f32x16 b2[NTW];
        if constexpr (AHEAD) {
            bv[0] = channel_vec(bl + (size_t)(layer + 1 < nl ? P.conv_b_stride : 0) + col0, half);
            if constexpr (!STATIC_TOWER) { if (second && layer + 2 < nl) b2[0] = channel_vec(bl + 2 * (size_t)P.conv_b_stride + col0, half); }
        }
        if (layer < 16) STAMP(3 + 3 * layer);
        __syncthreads();                               // every wave is done reading the layer input
        if (layer < 16) STAMP(4 + 3 * layer);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[mt][nt][v] = relu(acc[mt][nt][v]);
                store_tile(act + ridx[mt] + col0 + nt * 32, acc[mt][nt]);
            }
        if (second && layer + 2 < nl) {                // the next block's input + the bias of its second convolution
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if constexpr (!AHEAD || STATIC_TOWER) b2[nt] = channel_vec(bl + 2 * (size_t)P.conv_b_stride + col0 + nt * 32, half);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xres[mt][nt] = acc[mt][nt] + b2[nt];
            }
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
        const f32x16 bp = channel_vec(P.pol_b, half);
#pragma unroll
        for (int i = 0; i < PMT; ++i) {
            accp[i][0] = bp;
            pridx[i] = ridx[0]; pvmask[i] = vmask[0]; pneed[i] = need[0];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) if (mt == pt0 + i) { pridx[i] = ridx[mt]; pvmask[i] = vmask[mt]; pneed[i] = need[mt]; }
        }
    }
    if (pol_here)
        conv_mfma<G, PMT, 1, 1, S, PFE, 1, false>(accp, P.pol_w, act, pridx, pvmask, STRIDE, ZERO_IDX + 4 * half, lane, pneed);
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
            // NINE independent accumulation chains, one per tap, summed at the end in a fixed order (the three taps of a kernel
            // row first, then the rows): a dependent 4x4x1 MFMA issues only every ~4 passes, and with three chains (one per
            // kernel row) the value conv still took 9.6 us of a three-board workgroup's 139 (tools/variant_trace.py)
            f32x4 ar[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) ar[t] = (f32x4){0, 0, 0, 0};
            ar[0] = (f32x4){bj, bj, bj, bj};
            const float *xr[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const bool ok = real && yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS;
                xr[t] = act + (ok ? (r + (t / 3 - 1) * COLS + (t % 3 - 1)) * STRIDE : ZERO_IDX);
            }
            const float *wt = P.val_w4 + j * 4;                                // [tap][c / 4][j][c % 4]
#pragma unroll 2
            for (int c = 0; c < CP; c += 4) {
                float4 w[9], x[9];             // rows are 16-byte aligned: four input channels per LDS read
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    w[t] = *reinterpret_cast<const float4 *>(wt + ((size_t)t * CP + c) * 4);
                    x[t] = *reinterpret_cast<const float4 *>(xr[t] + c);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) ar[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[t].x, w[t].x, ar[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) ar[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[t].y, w[t].y, ar[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) ar[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[t].z, w[t].z, ar[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) ar[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[t].w, w[t].w, ar[t], 0, 0, 0);
            }
            f32x4 av;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r0 = (ar[0][i] + ar[1][i]) + ar[2][i], r1 = (ar[3][i] + ar[4][i]) + ar[5][i], r2 = (ar[6][i] + ar[7][i]) + ar[8][i];
                av[i] = (r0 + r1) + r2;
            }
            const int rb = wv * 64 + (lane & ~3);
#pragma unroll
            for (int i = 0; i < 4; ++i) hval[(rb + i) * 4 + j] = fmaxf(av[i], 0.0f);   // hval is not read before the barriers below
        }
    } else if constexpr (std::is_same<G, Gomoku9>::value && (NTHREADS / (TB * CELLS) >= 2)) {
        // 9x9 boards, one per workgroup: the input channels of a cell are split over VS threads (one slice each, the same
        // three chains per kernel row inside a slice); the slices are added in order below, after the barrier
        constexpr int VS = NTHREADS / (TB * CELLS) >= 4 ? 4 : 2, CSL = CP / VS;
        static_assert(CSL % 4 == 0, "channel slices are whole 16-byte reads");
        if (tid < VS * TB * CELLS) {
            const int cell = tid % (TB * CELLS), vs = tid / (TB * CELLS);
            const int pos = cell % CELLS, y = pos / COLS, x = pos % COLS;
            const bool real = board0 + cell / CELLS < n_leaf;
            float hr[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
            for (int dx = 0; dx < 3; ++dx) {
                const float *xr[3], *wv[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1, xx = x + dx - 1;
                    const bool ok = real && yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS;
                    xr[dy] = act + (ok ? (cell + (dy - 1) * COLS + (dx - 1)) * STRIDE : ZERO_IDX) + vs * CSL;
                    wv[dy] = P.val_w + ((size_t)(dy * 3 + dx) * CP + vs * CSL) * 4;
                }
#pragma unroll 2
                for (int c4 = 0; c4 < CSL; c4 += 4) {
                    float4 x[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) x[dy] = *reinterpret_cast<const float4 *>(xr[dy] + c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const float a = j == 0 ? x[dy].x : (j == 1 ? x[dy].y : (j == 2 ? x[dy].z : x[dy].w));
                            const float4 w = *reinterpret_cast<const float4 *>(wv[dy] + (c4 + j) * 4);
                            hr[dy][0] = fmaf(a, w.x, hr[dy][0]);
                            hr[dy][1] = fmaf(a, w.y, hr[dy][1]);
                            hr[dy][2] = fmaf(a, w.z, hr[dy][2]);
                        }
                }
            }
            float *dst = hval + (size_t)(vs * TB * CELLS + cell) * 4;      // slice vs of the cell; summed after the barrier
            static_assert(VS * TB * CELLS * 4 <= L::PERM - L::HVAL, "the slices fit the hval region");
            dst[0] = (hr[0][0] + hr[1][0]) + hr[2][0];
            dst[1] = (hr[0][1] + hr[1][1]) + hr[2][1];
            dst[2] = (hr[0][2] + hr[1][2]) + hr[2][2];
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
#pragma unroll 1
            for (int c4 = 0; c4 < CP; c4 += 4) {
                float4 x[3];                   // rows are 16-byte aligned: four input channels per LDS read
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) x[dy] = *reinterpret_cast<const float4 *>(xr[dy] + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const float a = j == 0 ? x[dy].x : (j == 1 ? x[dy].y : (j == 2 ? x[dy].z : x[dy].w));
                        const int c = c4 + j;
                        hr[dy][0] = fmaf(a, wv[dy][c * 4 + 0], hr[dy][0]);
                        hr[dy][1] = fmaf(a, wv[dy][c * 4 + 1], hr[dy][1]);
                        hr[dy][2] = fmaf(a, wv[dy][c * 4 + 2], hr[dy][2]);
                    }
            }
        }
        hv0 = (hr[0][0] + hr[1][0]) + hr[2][0];
        hv1 = (hr[0][1] + hr[1][1]) + hr[2][1];
        hv2 = (hr[0][2] + hr[1][2]) + hr[2][2];
    }
    STAMP(52);
    __syncthreads();
    // policy-head activations over the (now dead) tower activations, value-head ones beside them
    if (pol_here) {
#pragma unroll
        for (int i = 0; i < PMT; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) accp[i][0][v] = relu(accp[i][0][v]);
            store_tile(act + pridx[i], accp[i][0]);
        }
    }
    if constexpr (!VMFMA && std::is_same<G, Gomoku9>::value && (NTHREADS / (TB * CELLS) >= 2)) {
        constexpr int VS = NTHREADS / (TB * CELLS) >= 4 ? 4 : 2;
        if (tid < TB * CELLS) {                    // bias + the channel slices in order (a thread overwrites only its own slice 0)
            hv0 = P.val_b[0]; hv1 = P.val_b[1]; hv2 = P.val_b[2];
#pragma unroll
            for (int vs = 0; vs < VS; ++vs) {
                const float *src = hval + (size_t)(vs * TB * CELLS + tid) * 4;
                hv0 += src[0]; hv1 += src[1]; hv2 += src[2];
            }
        }
    }
    if (!VMFMA && tid < TB * CELLS) {
        hval[tid * 4 + 0] = fmaxf(hv0, 0.0f);
        hval[tid * 4 + 1] = fmaxf(hv1, 0.0f);
        hval[tid * 4 + 2] = fmaxf(hv2, 0.0f);
    }
    __syncthreads();
    STAMP(53);

    // ---- Linear layers: Flatten is (c, y, x) order (resnet.py:60-61, :69-70) -------------------------
    float *vlogit = vl;                            // (for A <= 7 the value logit lives in logits[b][7])
    if constexpr (A <= 7) {
        static_assert(TB * CELLS <= NTHREADS, "one thread per board cell");
        // thread r (< TB*CELLS) = one board cell: partial dot products over channels, fixed order
        if (tid < TB * CELLS) {
            const int pos = tid % CELLS;
            float pa[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) pa[o] = 0.0f;
            const float *hrow = act + tid * STRIDE;
            for (int c4 = 0; c4 < POL_CH; c4 += 4) {
                const float4 h4 = *reinterpret_cast<const float4 *>(hrow + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c4 + j;
                    const float h = j == 0 ? h4.x : (j == 1 ? h4.y : (j == 2 ? h4.z : h4.w));
                    const float4 w0 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8);
                    const float4 w1 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8 + 4);
                    pa[0] = fmaf(h, w0.x, pa[0]); pa[1] = fmaf(h, w0.y, pa[1]); pa[2] = fmaf(h, w0.z, pa[2]); pa[3] = fmaf(h, w0.w, pa[3]);
                    pa[4] = fmaf(h, w1.x, pa[4]); pa[5] = fmaf(h, w1.y, pa[5]); pa[6] = fmaf(h, w1.z, pa[6]);
                }
            }
            float pv = 0.0f;
#pragma unroll
            for (int c = 0; c < VAL_CH; ++c) pv = fmaf(hval[tid * 4 + c], P.fcv_w[c * CELLS + pos], pv);
            pa[7] = pv;
#pragma unroll
            for (int o = 0; o < 8; ++o) part[tid * PSTR + o] = pa[o];
        }
        __syncthreads();
        if (tid < TB * 8) {
            const int b = tid >> 3, o = tid & 7;
            float sum = 0.0f;
            for (int pos = 0; pos < CELLS; ++pos) sum += part[(b * CELLS + pos) * PSTR + o];
            logits[tid] = sum + (o < 7 ? (o < A ? P.fcp_b[o] : 0.0f) : P.fcv_b[0]);
        }
        vlogit = nullptr;
    } else {
        // wide action spaces (Gomoku: 81): the 32 input channels of an output are cut into KS slices (whole channels, a
        // slice walks its CELLS inputs per channel in flatten order), one thread per (slice, board, output); weights are
        // stored [input][AP] so that the threads of a slice read them coalesced, nine loads in flight per thread.  The
        // slices are added in order after the barrier.
        constexpr int NOUT = TB * (A + 1);                    // per board: A policy logits + the value logit
        constexpr int KS = NTHREADS / NOUT >= 8 ? 8 : (NTHREADS / NOUT >= 4 ? 4 : (NTHREADS / NOUT >= 2 ? 2 : 1));
        constexpr int CS = POL_CH / KS;                       // channels per slice
        static_assert(KS * NOUT <= WG_ROWS * 8, "the partial sums [KS][NOUT] fit the FC partial-sum area");
        auto fslot = [&](int i) -> float & { return part[(i >> 3) * PSTR + (i & 7)]; };
#pragma unroll 1
        for (int t = tid; t < KS * NOUT; t += NTHREADS) {
            const int ks = t / NOUT, item = t % NOUT, b = item / (A + 1), o = item % (A + 1);
            float sum = 0.0f;
            if (o < A) {
#pragma unroll 1
                for (int c = ks * CS; c < (ks + 1) * CS; ++c) {
                    const float *ap = act + (size_t)b * CELLS * STRIDE + c;
                    const float *wp = P.fcp_w + (size_t)c * CELLS * AP + o;
#pragma unroll 9
                    for (int pos = 0; pos < CELLS; ++pos) sum = fmaf(ap[pos * STRIDE], wp[(size_t)pos * AP], sum);
                }
            } else if (ks == 0) {
#pragma unroll 1
                for (int c = 0; c < VAL_CH; ++c)
#pragma unroll 3
                    for (int pos = 0; pos < CELLS; ++pos)
                        sum = fmaf(hval[(b * CELLS + pos) * 4 + c], P.fcv_w[c * CELLS + pos], sum);
            }
            fslot(t) = sum;
        }
        __syncthreads();
        for (int item = tid; item < NOUT; item += NTHREADS) {
            const int b = item / (A + 1), o = item % (A + 1);
            float sum = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sum += fslot(ks * NOUT + item);
            if (o < A) logits[b * AP + o] = sum + P.fcp_b[o];
            else vl[b] = sum + P.fcv_b[0];
        }
        vlogit = vl;
    }
    __syncthreads();
    STAMP(54);
    // ---- masked softmax + tanh (ResNet.predict resnet.py:143-151; legal = the game's valid moves) --------
    if constexpr (A <= 7) {
        // eight lanes per board (one per action, the eighth holds the value logit): every lane exponentiates once, the
        // denominator is summed in action order exactly like the serial form (0 + e0 + e1 + ...), which the wide-action
        // form below still is
        if (tid < TB * 8) {
            const int b = tid >> 3, a = tid & 7;
            const bool liveb = board0 + b < n_leaf;
            const int slot = liveb ? leaf_slot[board0 + b] : 0;
            const typename G::B bmine = board_load_strided<G::W>(mine, slot, bstride), btheirs = board_load_strided<G::W>(theirs, slot, bstride);
            const bool leg = liveb && a < A && G::legal(bmine, btheirs, a);
            const float x = logits[b * AP + a];
            float mx = leg ? x : -INFINITY;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 8));
            const float ex = leg ? expf(x - mx) : 0.0f;
            float den = 0.0f;
#pragma unroll
            for (int k = 0; k < A; ++k) den += __shfl(ex, k, 8);
            const float pr = leg ? ex / den : 0.0f;
            const float val = tanhf(x);                          // meaningful in lane 7 (the value logit)
            if (liveb) {
                if constexpr (FUSED) {
                    // the results stay in LDS (probabilities over the logits, value in vl): this workgroup expands the leaves itself
                    logits[b * AP + a] = pr;
                    if (a == 7) vl[b] = val;
                } else {
                    out_p[(size_t)slot * p_stride + a] = pr;           // results by slot
                    if (a == 7) out_v[(size_t)slot * v_stride] = val;
                }
            }
        }
    } else {
        // wide action spaces: wave b takes board b, a lane the actions lane and lane + 64; maximum and denominator by
        // butterfly reductions (every lane adds the same pairs in the same order)
        static_assert(A <= 7 || !FUSED, "FUSED rounds exist for Connect-4");
        static_assert(A <= 128 && AP <= 128 && TB <= NTHREADS / 64, "one wave per board, two actions per lane");
        const int b = tid >> 6;
        if (b < TB && board0 + b < n_leaf) {
            const int slot = leaf_slot[board0 + b];
            const typename G::B bmine = board_load_strided<G::W>(mine, slot, bstride), btheirs = board_load_strided<G::W>(theirs, slot, bstride);
            const float *lg = logits + b * AP;
            float x[2], ex[2];
            bool leg[2];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a = lane + 64 * i;
                leg[i] = a < A && G::legal(bmine, btheirs, a);
                x[i] = a < A ? lg[a] : 0.0f;
                if (leg[i]) mx = fmaxf(mx, x[i]);
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            ex[0] = leg[0] ? expf(x[0] - mx) : 0.0f;
            ex[1] = leg[1] ? expf(x[1] - mx) : 0.0f;
            float den = ex[0] + ex[1];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) den += __shfl_xor(den, m, 64);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a = lane + 64 * i;
                if (a < AP) out_p[(size_t)slot * p_stride + a] = leg[i] ? ex[i] / den : 0.0f;     // results by slot
            }
            if (lane == 0) out_v[(size_t)slot * v_stride] = tanhf(vlogit ? vlogit[b] : lg[7]);
        }
    }
    STAMP(55);
    if constexpr (FUSED) {
        // ---- the games go on: expansion + backup, then descents until the next leaf that needs the network ----------
        // (advance_game, azsp_tree_dev.h; one wave, LPG lanes per game; scratch over the dead activation rows)
        static_assert(std::is_same<G, Connect4>::value && TB <= 64 / G::LPG, "one wave carries the workgroup's games");
        double (*s_pa)[G::AP] = reinterpret_cast<double (*)[G::AP]>(act);
        int *s_path = reinterpret_cast<int *>(act + 2 * (64 / G::LPG) * G::AP);
        double *s_tab = reinterpret_cast<double *>(s_path + (((64 / G::LPG) * TreeGeo<G>::D + 1) & ~1));
        const int ntab = F.v.tab_lds;
        for (int i = tid; i < ntab; i += NTHREADS) s_tab[i] = F.v.expl_tab[i];
        __syncthreads();
        if (tid < 64) {
            Tabs tb;
            tb.expl = s_tab; tb.sqrt_ = s_tab; tb.c = s_tab;
            const int ln = tid & (G::LPG - 1), grp = tid / G::LPG;
            const bool live = grp < TB && board0 + grp < n_leaf;
            const int g = live ? leaf_slot[board0 + grp] : F.v.lo;
            float p32[G::APL];
#pragma unroll
            for (int k = 0; k < G::APL; ++k) {
                const int a = ln + k * G::LPG;
                p32[k] = (live && a < A) ? logits[grp * AP + a] : 0.0f;
            }
            const float vleaf = live ? vl[grp] : 0.0f;
            advance_game<G, 0>(F.v, g, ln, grp, live, p32, vleaf, s_pa, s_path, tb, F.next_list, F.next_count);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    NetParams P;
    const u64 *mine, *theirs;               // leaf position of slot g: W words at mine + g * bstride / theirs + g * bstride
    int bstride;
    const int32_t *leaf_slot, *n_leaf;      // leaf list of the group (slot ids) and its length (device)
    int max_leaves;
    float *out_p, *out_v;                   // results of slot g at out_p + g * p_stride, out_v + g * v_stride (floats)
    int p_stride, v_stride;
    hipStream_t s;
    const FusedArgs *fused;                 // FUSED variants only, else nullptr
    int gate = 0;                           // (lo / 16) | (hi / 16) << 16: run only for list lengths in (lo, hi]; 0 = always
    // AZSP_FLAG_TIMING: events that take the launch's own start / end timestamps from its dispatch packet
    // (hipExtLaunchKernelGGL) -- no extra packets between dependent kernels, unlike hipEventRecord around the launch
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

// tracing state lives in azsp_resnet.hip (one copy for all kernel translation units)
bool resnet_trace_wanted(bool *print);                                         // env AZSP_RESNET_TRACE, or armed for one launch
void resnet_trace_report(const unsigned long long *stamps, int grid, bool print);

// Diagnostic: run the launch with per-workgroup phase timestamps and hand them to resnet_trace_report.
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
    resnet_trace_report(h.data(), grid, print);
}

// TRACEABLE: also instantiate the stamped diagnostic form of this variant (the headline kernel only)
template <class G, int NT, int MT, int PF, int CW = 1, int RW = 4, bool TRACEABLE = false, bool FUSED = false>
static void launch_forward(const FwdArgs &a) {
    using L = LdsLayout<G, NT, MT, CW, RW>;
    constexpr int TB = L::TB;
    static_assert(TB >= 1 && TB * G::CELLS <= 64 * RW * CW, "one thread per board cell in the head phases");
    const size_t lds = (size_t)L::TOTAL * sizeof(float);
    // the attribute is per device and idempotent: a racing second call sets the same value
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resnet_forward<G, NT, MT, PF, CW, RW, false, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if constexpr (TRACEABLE)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resnet_forward<G, NT, MT, PF, CW, RW, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int gate_hi = (int)((unsigned)a.gate >> 16) * 16;
    const int cap = gate_hi != 0 && gate_hi < a.max_leaves ? gate_hi : a.max_leaves;      // longest list this launch can meet
    const int grid = (cap + TB - 1) / TB;
    if constexpr (TRACEABLE) {
        bool print = false;
        if (resnet_trace_wanted(&print)) {
            trace_forward(a.P, grid, [&](const NetParams &Q) {
                hipLaunchKernelGGL((k_resnet_forward<G, NT, MT, PF, CW, RW, true>), dim3(grid), dim3(64 * RW * CW), lds, a.s, Q,
                                   a.mine, a.theirs, a.bstride, a.leaf_slot, a.n_leaf, a.out_p, a.out_v, a.p_stride, a.v_stride, a.gate); }, a.s, print);
            return;
        }
    }
    if constexpr (FUSED) {
        hipExtLaunchKernelGGL((k_resnet_forward<G, NT, MT, PF, CW, RW, false, true>), dim3(grid), dim3(64 * RW * CW), lds, a.s, a.ev_start,
                              a.ev_stop, 0, a.P, a.mine, a.theirs, a.bstride, a.leaf_slot, a.n_leaf, a.out_p, a.out_v, a.p_stride, a.v_stride,
                              *a.fused);
    } else {
        hipExtLaunchKernelGGL((k_resnet_forward<G, NT, MT, PF, CW, RW>), dim3(grid), dim3(64 * RW * CW), lds, a.s, a.ev_start, a.ev_stop, 0,
                              a.P, a.mine, a.theirs, a.bstride, a.leaf_slot, a.n_leaf, a.out_p, a.out_v, a.p_stride, a.v_stride, a.gate);
    }
}

// One function per compiled kernel variant (azsp_resnet_k*.hip); the table that picks among them is in azsp_resnet.hip.
void fwd_c4_64_big(const FwdArgs &);      // Connect-4, 64 ch, 6 boards / workgroup, 2 x 2 waves, static tower (+ traced form)
void fwd_c4_128_big(const FwdArgs &);     // Connect-4, 128 ch, 6 boards / workgroup, 2 x 4 waves, static tower
void fwd_c4_64_one_fused(const FwdArgs &);    // the one-board kernels of small batches with the tree step fused in
void fwd_c4_128_one_fused(const FwdArgs &);
void fwd_c4_32_small(const FwdArgs &);    // Connect-4, 32 ch, 3 boards / workgroup, double-buffered weight stream
void fwd_c4_32_big(const FwdArgs &);      // Connect-4, 32 ch, 6 boards / workgroup, tap skipping
void fwd_c4_64_one(const FwdArgs &);      // Connect-4, 64 ch, 1 board / workgroup (latency-bound batches)
void fwd_c4_64_mid(const FwdArgs &);      // Connect-4, 64 ch, 3 boards / workgroup
void fwd_c4_96(const FwdArgs &);          // Connect-4, 96 ch
void fwd_c4_128_one(const FwdArgs &);     // Connect-4, 128 ch, 1 board / workgroup
void fwd_c4_128_mid(const FwdArgs &);     // Connect-4, 128 ch, 3 boards / workgroup
void fwd_c4_192(const FwdArgs &);         // Connect-4, 192 ch (160 pads to it)
void fwd_c4_256(const FwdArgs &);         // Connect-4, 256 ch (224 pads to it)
void fwd_g9_32(const FwdArgs &);          // Gomoku 9x9, 3 boards / workgroup
void fwd_g9_64(const FwdArgs &);
void fwd_g9_96(const FwdArgs &);          // Gomoku 9x9, 1 board / workgroup, one wave per column tile
void fwd_g9_128(const FwdArgs &);
void fwd_g9_192(const FwdArgs &);
void fwd_g9_256(const FwdArgs &);

}  // namespace azsp
