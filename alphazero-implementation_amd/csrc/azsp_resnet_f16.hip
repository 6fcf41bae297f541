// Split-fp16 ("f16x2") variant of the fused ResNet inference kernel: the same network and the same
// fusion as azsp_resnet.hip, but every fp32 operand is carried as two fp16 numbers and the
// convolutions run on v_mfma_f32_32x32x16_f16 (16x the per-clock rate of the fp32 MFMA):
//
//     a = ah + al,  w = wh + wl   (ah = fp16(a), al = fp16(a - ah), same for w; power-of-two pre-scaling
//                                  keeps the residuals out of the fp16 subnormal range)
//     a*w  ~=  ah*wh + ah*wl + al*wh            (the dropped al*wl term is <= 2^-22 |a*w|)
//
// Products of fp16 numbers are exact in fp32 and the MFMA accumulates in fp32, so the result carries
// ~22 mantissa bits -- fp32-class accuracy (tests: <= 1e-5 on p and v against the reference vectors, same
// bound as the fp32 kernel) at 3 MFMAs of the 16x-rate instruction per 16-deep K slab = 5.3x fewer matrix
// cycles.  Reduction order is fixed per output, so the kernel is batch-invariant like the fp32 one.
//
// Mapping: workgroup = 256 GEMM rows (6 boards) x all channels; waves form a WR x CT grid, wave (wr, wc)
// owns MT = 8/WR row tiles x one 32-channel column tile (A fragments of a row tile are shared by the CT
// waves of a row group through LDS; each weight fragment is loaded by WR waves only).  Activations live
// in LDS as two fp16 planes [row][C+8] (row stride = odd number of 16-byte slots: conflict-free
// ds_read_b128), residual in fp32 registers, single fp32 accumulator per tile.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include <hip/hip_ext.h>

#include "azsp_internal.h"
#include "azsp_resnet_host.h"

namespace azsp {

// tracing state shared with the fp32 kernel (azsp_resnet.hip)
bool resnet_trace_wanted(bool *print);
void resnet_trace_report(const unsigned long long *stamps, int grid, bool print);

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F16_ROWS = 256;
constexpr int ACT_SHIFT = 4;            // activations are stored scaled by 2^4
constexpr float ACT_SCALE = 16.0f;
constexpr int HB_STRIDE = 37;           // floats per row of the head-activation buffer (odd: conflict-free column reads)
constexpr int TB_STRIDE = 33;           // floats per row of the value conv's per-tap partial sums

struct Net16Params {
    const h8 *stem_w, *conv_w, *pol_w, *val_w;
    const float *stem_b, *conv_b, *pol_b, *val_b, *fcp_w, *fcp_b, *fcv_w, *fcv_b;
    const float *conv_up, *conv_inv;   // per conv layer: 2^(ACT_SHIFT + k_l) and its inverse
    float stem_inv, pol_inv, val_inv;
    int conv_w_stride /* in h8 */, conv_b_stride, blocks;
    const int32_t *perm;               // [256] GEMM row -> board-cell row for this wave grid (tap skipping)
    unsigned long long *dbg;           // phase timestamps of the ABL = 8 diagnostic instantiation, else nullptr
    double *flop_acc;                  // device accumulator of executed matrix-pipe FLOP, or nullptr
    double wg_flops;                   // fp16 MFMA FLOP one workgroup executes (set per launch from the two figures below)
    double tower_row_flops, rest_board_flops;   // executed fp16 FLOP of the tower per GEMM row (no tap skipped), of stem + heads per board
    int gate;                          // launch gate as in the fp32 kernel: (lo / 16) | (hi / 16) << 16, 0 = none
};
// phase timestamp (100 MHz clock) of wave 0 of each workgroup; compiled only into the ABL & 8 diagnostic build
#define STAMP16(k) do { if constexpr ((ABL & 8) != 0) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + (k)] = wall_clock64(); } } while (0)

struct ResNet16Dev {
    int game = 0, CT = 0, C = 0, blocks = 0, cells = 0, A = 0, AS = 8;      // AS: floats per (channel, cell) row of the policy FC weights
    unsigned char *d_blob = nullptr;
    size_t bytes = 0;
    size_t o_stem_w, o_conv_w, o_pol_w, o_val_w;                                    // byte offsets of h8 arrays
    size_t o_stem_b, o_conv_b, o_pol_b, o_val_b, o_fcp_w, o_fcp_b, o_fcv_w, o_fcv_b, o_up, o_inv;   // float arrays
    size_t conv_w_stride = 0, conv_b_stride = 0;
    float stem_inv = 1, pol_inv = 1, val_inv = 1;
    int32_t *d_perm = nullptr;          // [4][256]: row orders for WR = 1, 2, 4, 8
};

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// C/D layout of the 32x32 MFMAs: register v of lane l holds row (v & 3) + 8 * (v >> 2) + 4 * (l >> 5), column l & 31.
// The convolutions put the WEIGHTS on the A side and the activations on the B side (both operands have the same lane
// layout), so the row is the output channel within the column tile and the column the board cell: a lane's 16 results
// are ONE cell x four runs of four consecutive channels, which the epilogue converts and stores four at a time
// (8-byte LDS writes per plane at the lane's own row) -- as in the fp32 kernel (conv_mma, azsp_resnet_kernel.h).
__device__ __forceinline__ int cd_ch16(int v, int lane) { return (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 mfma16(h8 a_act, h8 b_w, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(b_w, a_act, c, 0, 0, 0);
}
// the 16 per-channel constants of a lane for the column tile whose channel 0 is at b
__device__ __forceinline__ f32x16 channel_vec16(const float *b, int half) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = *reinterpret_cast<const float4 *>(b + 8 * i + 4 * half);
        r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    }
    return r;
}

// conv over the LDS-resident split activations for NTILE row tiles of this wave and ONE column tile.
//   w     : [tap][slab][ct][plane][64 lanes] h8, this wave's column tile `ct` selected by the caller via w + ...
//   abyte : per tile, byte offset of (row, 8*half) inside a plane; vmask: 9-bit tap validity
template <class G, int NTILE, int TAPS, int SL, int CT, int ABL = 0>
__device__ __forceinline__ void conv16(f32x16 (&acc)[NTILE], const h8 *__restrict__ w, const unsigned char *plane_hi,
                                       int plane_bytes, const int (&abyte)[NTILE], const uint32_t (&vmask)[NTILE],
                                       int row_bytes, int zero_byte, int lane) {
    // Written as a software pipeline over the flattened (tap, slab, row tile) sequence (A fragments of the next
    // tile and B fragments two slabs ahead are requested before the three MFMAs of the current tile).  hipcc's
    // scheduler re-orders this freely (measured: same speed as the naive loop; pinning the order with
    // sched_barrier makes its waits conservative: -6 % at 4096 boards, +13 % at 100) -- see DESIGN.md section 10.
    auto arow_of = [&](int tap, int i) -> const unsigned char * {
        const int t9 = (TAPS == 9) ? tap : 4;
        const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
        return plane_hi + (((vmask[i] >> t9) & 1u) ? abyte[i] + (dy * G::COLS + dx) * row_bytes : zero_byte);
    };
    auto wptr = [&](int q) -> const h8 * {          // q = tap * SL + slab (clamped at the end of the layer)
        const int qq = q < TAPS * SL ? q : TAPS * SL - 1;
        return w + ((size_t)((ABL & 1) ? 0 : qq) * CT) * 2 * 64 + lane;
    };
    const unsigned char *arow[NTILE], *arow_n[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) arow[i] = arow_of(0, i);
    h8 ah_n = *reinterpret_cast<const h8 *>(arow[0]);
    h8 al_n = *reinterpret_cast<const h8 *>(arow[0] + plane_bytes);
    h8 bh_1 = wptr(0)[0], bl_1 = wptr(0)[64];       // slab q
    h8 bh_2 = wptr(1)[0], bl_2 = wptr(1)[64];       // slab q + 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int tn = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
        for (int i = 0; i < NTILE; ++i) arow_n[i] = arow_of(tn, i);
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
            const h8 bh = bh_1, bl = bl_1;
            bh_1 = bh_2; bl_1 = bl_2;
            const h8 *wn = wptr(tap * SL + sl + 2);
            bh_2 = wn[0]; bl_2 = wn[64];
#pragma unroll
            for (int i = 0; i < NTILE; ++i) {
                const h8 ah = ah_n, al = al_n;
                const unsigned char *nx = (i + 1 < NTILE) ? arow[i + 1] + sl * 32
                                        : (sl + 1 < SL)   ? arow[0] + (sl + 1) * 32
                                                          : arow_n[0];
                ah_n = *reinterpret_cast<const h8 *>(nx);
                al_n = *reinterpret_cast<const h8 *>(nx + plane_bytes);
                acc[i] = mfma16(ah, bh, acc[i]);
                acc[i] = mfma16(ah, bl, acc[i]);
                acc[i] = mfma16(al, bh, acc[i]);
                if constexpr (CT > 4) __builtin_amdgcn_sched_barrier(0);      // wide nets' three-board form: pinned order (4 x 256, 3072 boards: 3.60 -> 3.15 ms); the six-board forms lose with it
            }
        }
#pragma unroll
        for (int i = 0; i < NTILE; ++i) arow[i] = arow_n[i];
    }
}

// conv16 with deeper rings, for the one-row-group forms with two or three row tiles per wave (the wide nets' one-board forms,
// Gomoku): 9-12 MFMAs per slab do not cover an L2 / MALL round trip of the weights two slabs ahead, nor three MFMAs an LDS
// read under eight waves' traffic.  Same order of sums.  (With four row tiles per wave the rotating form above is faster:
// 3072 boards at 4 x 192 2.10 against 2.49 ms.)
template <class G, int NTILE, int TAPS, int SL, int CT, int PD>
__device__ __forceinline__ void conv16r(f32x16 (&acc)[NTILE], const h8 *__restrict__ w, const unsigned char *plane_hi,
                                       int plane_bytes, const int (&abyte)[NTILE], const uint32_t (&vmask)[NTILE],
                                       int row_bytes, int zero_byte, int lane) {
    // Written as a software pipeline over the flattened (tap, slab, row tile) sequence: the A fragments of the next tile
    // and the B fragments (weights) PD slabs ahead are requested before the three MFMAs of the current tile; the weights
    // sit in a ring of PD register pairs indexed statically (PD divides SL).  PD = 2 is enough where the six-board form's
    // eight row tiles give a slab 24 MFMAs; the one-row-group forms of the wide nets have 9-12 MFMAs per slab and need the
    // weights of an L2 / MALL round trip in flight: PD = 4.  hipcc's scheduler re-orders this freely (measured: same speed
    // as the naive loop; pinning the order with sched_barrier makes its waits conservative: -6 % at 4096 boards, +13 % at
    // 100) -- see DESIGN.md section 10.
    static_assert(SL % PD == 0 || TAPS * SL < PD, "static ring indices");
    auto arow_of = [&](int tap, int i) -> const unsigned char * {
        const int t9 = (TAPS == 9) ? tap : 4;
        const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
        return plane_hi + (((vmask[i] >> t9) & 1u) ? abyte[i] + (dy * G::COLS + dx) * row_bytes : zero_byte);
    };
    auto wptr = [&](int q) -> const h8 * {          // q = tap * SL + slab (clamped at the end of the layer)
        const int qq = q < TAPS * SL ? q : TAPS * SL - 1;
        return w + ((size_t)qq * CT) * 2 * 64 + lane;
    };
    // A fragments: a ring of AR register pairs, the pair of step j + AD requested at step j (a step = one row tile of one slab)
    constexpr int STEPS = SL * NTILE, AR = STEPS % 4 == 0 ? 4 : 2, AD = AR - 1;
    const unsigned char *arow[NTILE], *arow_n[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) arow[i] = arow_of(0, i);
    h8 xh[AR], xl[AR];
#pragma unroll
    for (int j = 0; j < AD; ++j) {
        const unsigned char *a0 = arow[j % NTILE] + (j / NTILE) * 32;
        xh[j] = *reinterpret_cast<const h8 *>(a0);
        xl[j] = *reinterpret_cast<const h8 *>(a0 + plane_bytes);
    }
    h8 rh[PD], rl[PD];                              // slabs q .. q + PD - 1
#pragma unroll
    for (int k = 0; k < PD; ++k) { rh[k] = wptr(k)[0]; rl[k] = wptr(k)[64]; }
    for (int tap = 0; tap < TAPS; ++tap) {
        const int tn = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
        for (int i = 0; i < NTILE; ++i) arow_n[i] = arow_of(tn, i);
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
            const h8 bh = rh[sl % PD], bl = rl[sl % PD];
            const h8 *wn = wptr(tap * SL + sl + PD);
            rh[sl % PD] = wn[0]; rl[sl % PD] = wn[64];
#pragma unroll
            for (int i = 0; i < NTILE; ++i) {
                const int j = sl * NTILE + i, jn = j + AD;                 // (static after unrolling)
                const unsigned char *nx = jn < STEPS ? arow[jn % NTILE] + (jn / NTILE) * 32
                                                     : arow_n[(jn - STEPS) % NTILE] + ((jn - STEPS) / NTILE) * 32;
                const h8 ah = xh[j % AR], al = xl[j % AR];
                xh[jn % AR] = *reinterpret_cast<const h8 *>(nx);
                xl[jn % AR] = *reinterpret_cast<const h8 *>(nx + plane_bytes);
                acc[i] = mfma16(ah, bh, acc[i]);
                acc[i] = mfma16(ah, bl, acc[i]);
                acc[i] = mfma16(al, bh, acc[i]);
                __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of the MFMAs (the compiler moves them back to their uses)
            }
        }
#pragma unroll
        for (int i = 0; i < NTILE; ++i) arow[i] = arow_n[i];
    }
}

// The same convolution with tap skipping: `need[i]` (wave-uniform) = taps for which some row of tile i has an
// on-board input; the others would add exact zeros.  The kernel orders its rows (perm) so that whole row tiles
// lack the same taps -- see build_row_order16.  Loop order tap -> (weights of the tap in registers) -> tile -> slab.
template <class G, int NTILE, int TAPS, int SL, int CT>
__device__ __forceinline__ void conv16s(f32x16 (&acc)[NTILE], const h8 *__restrict__ w, const unsigned char *plane_hi,
                                        int plane_bytes, const int (&abyte)[NTILE], const uint32_t (&vmask)[NTILE],
                                        const uint32_t (&need)[NTILE], int row_bytes, int zero_byte, int lane) {
    // the weights of a tap live in registers; the next tap's are requested while this tap's MFMAs run
    h8 bh[SL], bl[SL], nh[SL], nl[SL];
    auto load_b = [&](int tap, h8 (&h)[SL], h8 (&l)[SL]) {
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
            const h8 *wp = w + ((size_t)(tap * SL + sl) * CT) * 2 * 64 + lane;
            h[sl] = wp[0];
            l[sl] = wp[64];
        }
    };
    constexpr bool DB = SL <= 4;            // double-buffer only while the registers last (<= 64 channels; with more slabs the second
                                            // buffer spills even in the one-board form: 9 x 128 launch 0.256 -> 0.367 ms)
    // (Without the second buffer the tap's last tile reloading each slab's pair with the next tap's right after its last use was
    //  measured too: slower, 9 x 128 one board 225 -> 268 us.)
    if constexpr (DB) load_b(0, nh, nl);
    for (int tap = 0; tap < TAPS; ++tap) {
        const int t9 = (TAPS == 9) ? tap : 4;
        if constexpr (DB) {
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) { bh[sl] = nh[sl]; bl[sl] = nl[sl]; }
            if (tap + 1 < TAPS) load_b(tap + 1, nh, nl);
        } else {
            load_b(tap, bh, bl);
        }
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < NTILE; ++i) any |= (need[i] >> t9) & 1u;
        if (!any) continue;
        const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
        const unsigned char *ar[NTILE];
#pragma unroll
        for (int i = 0; i < NTILE; ++i)
            ar[i] = plane_hi + (((vmask[i] >> t9) & 1u) ? abyte[i] + (dy * G::COLS + dx) * row_bytes : zero_byte);
        // (taking the tiles in pairs so that neighbouring MFMAs write different accumulators was measured: slower,
        //  0.156 vs 0.146 ms per 1536-board launch -- the extra live fragments cost more than the dependences)
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            if (!((need[i] >> t9) & 1u)) continue;
            // the tile's fragments two slabs ahead of their MFMAs, order pinned
            h8 xh[3], xl[3];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                xh[k] = *reinterpret_cast<const h8 *>(ar[i] + k * 32);
                xl[k] = *reinterpret_cast<const h8 *>(ar[i] + k * 32 + plane_bytes);
            }
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) {
                if (sl + 2 < SL) {
                    xh[(sl + 2) % 3] = *reinterpret_cast<const h8 *>(ar[i] + (sl + 2) * 32);
                    xl[(sl + 2) % 3] = *reinterpret_cast<const h8 *>(ar[i] + (sl + 2) * 32 + plane_bytes);
                }
                acc[i] = mfma16(xh[sl % 3], bh[sl], acc[i]);
                acc[i] = mfma16(xh[sl % 3], bl[sl], acc[i]);
                acc[i] = mfma16(xl[sl % 3], bh[sl], acc[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// Both head convolutions in ONE pass over the activations.  The policy conv is 1x1 (32 channels).  The value conv is 3x3
// with 3 output channels: as a 3x3 conv on a 32-wide column tile it would run 9 taps x 32 columns for 27 useful dot products
// per cell; it is computed as the 1x1 conv  T[cell][tap * 3 + c] = sum_k act[cell][k] * Wv[c][k][tap]  (27 of 32 columns) and
// the nine taps are added per cell from the neighbours' T afterwards (value_from_taps in the kernel): a ninth of the MFMAs.
//   wp, wt : [slab][plane][64 lanes] h8 (permute_split with one tap, one column tile)
template <class G, int NTILE, int SL>
__device__ __forceinline__ void head_pair16(f32x16 (&accp)[NTILE], f32x16 (&acct)[NTILE], const h8 *__restrict__ wp, const h8 *__restrict__ wt,
                                            const unsigned char *plane_hi, int plane_bytes, const int (&abyte)[NTILE],
                                            const uint32_t (&vmask)[NTILE], int zero_byte, int lane) {
    const unsigned char *ar[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) ar[i] = plane_hi + (((vmask[i] >> 4) & 1u) ? abyte[i] : zero_byte);      // the cell's own row
    h8 ph = wp[lane], pl = wp[64 + lane], th = wt[lane], tl = wt[64 + lane];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
        const h8 cph = ph, cpl = pl, cth = th, ctl = tl;
        if (sl + 1 < SL) {
            const h8 *a = wp + (size_t)(sl + 1) * 2 * 64 + lane, *b = wt + (size_t)(sl + 1) * 2 * 64 + lane;
            ph = a[0]; pl = a[64]; th = b[0]; tl = b[64];
        }
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            const h8 ah = *reinterpret_cast<const h8 *>(ar[i] + sl * 32);
            const h8 al = *reinterpret_cast<const h8 *>(ar[i] + sl * 32 + plane_bytes);
            accp[i] = mfma16(ah, cph, accp[i]);
            accp[i] = mfma16(ah, cpl, accp[i]);
            accp[i] = mfma16(al, cph, accp[i]);
            acct[i] = mfma16(ah, cth, acct[i]);
            acct[i] = mfma16(ah, ctl, acct[i]);
            acct[i] = mfma16(al, cth, acct[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// statically scheduled tower layer for the 2 x CT wave grid (4 row tiles x one column tile per wave)
// ------------------------------------------------------------------------------------------------
// The same idea as tower_layer of the fp32 kernel (azsp_resnet_kernel.h): which tile-taps can be skipped is a static
// property of the row order, so a layer is one straight-line sequence of steps (tile-tap x 16-deep K slab, three MFMAs
// each) per row group.  Step i first requests the two activation fragments (hi, lo plane) of step i + 2 (three rotating
// buffer pairs), then issues its MFMAs.  Weights move in chunks of two slabs (half a tap at 64 channels, 16 registers)
// through THREE rotating buffers: the first tile of chunk c loads chunk c + 2, so a load has at least a whole chunk of
// MFMAs to land (two full-tap buffers do not fit beside the accumulators and the residual: 254 spills).  A layer is
// 18 chunks, a multiple of 3, so every layer finds its first two chunks in buffers 0 and 1.  sched_barrier between steps
// keeps the compiler from moving the reads back to their uses (the compiler-scheduled form spends half its wave cycles
// in s_waitcnt: profiles/r02_a_mfma_busy.txt).
struct TStep16 { int chunk, tap, tile, sl, first_tile; };
__host__ __device__ constexpr bool tile_needs16(int group, int tile, int tap) {       // row order: build_row_order16(WR = 2)
    if (tile >= 2) return true;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if (group == 0) return tile == 0 ? dx != -1 : dy != -1;       // left column, top row
    return tile == 0 ? dx != 1 : dy != 1;                         // right column, bottom row
}
template <int GROUP, int SL>
struct TowerSched16 {
    static constexpr int NSTEP = 30 * SL, CPT = SL / 2;            // chunks of two slabs per tap
    TStep16 s[NSTEP + 2];
    constexpr TowerSched16() : s{} {
        int n = 0;
        for (int tap = 0; tap < 9; ++tap)
            for (int h = 0; h < CPT; ++h) {
                int first = 1;
                for (int tile = 3; tile >= 0; --tile) {
                    if (!tile_needs16(GROUP, tile, tap)) continue;
                    for (int k = 0; k < 2; ++k) { s[n].chunk = tap * CPT + h; s[n].tap = tap; s[n].tile = tile; s[n].sl = 2 * h + k; s[n].first_tile = first; ++n; }
                    first = 0;
                }
            }
        s[n] = s[n - 1]; s[n + 1] = s[n - 1];          // the reads past the end repeat the last one
    }
};
template <int GROUP, int SL>
struct TowerHolder16 { static constexpr TowerSched16<GROUP, SL> SC{}; };

template <class G, int ROW_BYTES>
__device__ __forceinline__ int abase16(int abyte_t, uint32_t vmask_t, int tap, int zero_byte) {      // byte offset into the hi plane
    return ((vmask_t >> tap) & 1u) ? abyte_t + ((tap / 3 - 1) * G::COLS + (tap % 3 - 1)) * ROW_BYTES : zero_byte;
}

template <class G, int GROUP, int SL, int CT, int ROW_BYTES, int PLANE_BYTES, int I>
__device__ __forceinline__ void tower16_step(f32x16 (&acc)[4], const unsigned char *lds, const int (&abyte)[4], const uint32_t (&vmask)[4],
                                             int zero_byte, const h8 *w, h8 (&wh)[3][2], h8 (&wl)[3][2], h8 (&xh)[3], h8 (&xl)[3], int &rbase) {
    constexpr TStep16 st = TowerHolder16<GROUP, SL>::SC.s[I], nx = TowerHolder16<GROUP, SL>::SC.s[I + 2];
    constexpr int buf = st.chunk % 3;
    if constexpr ((nx.sl & 1) == 0 && I + 2 < TowerSched16<GROUP, SL>::NSTEP)        // step i + 2 opens a new (chunk, tile) block
        rbase = abase16<G, ROW_BYTES>(abyte[nx.tile], vmask[nx.tile], nx.tap, zero_byte);
    xh[(I + 2) % 3] = *reinterpret_cast<const h8 *>(lds + rbase + nx.sl * 32);
    xl[(I + 2) % 3] = *reinterpret_cast<const h8 *>(lds + rbase + nx.sl * 32 + PLANE_BYTES);
    if constexpr (st.first_tile != 0) {            // slab of the chunk after next (past the layer's end: the next layer's)
        const h8 *wp = w + ((size_t)(st.sl + 4) * CT) * 2 * 64 + (size_t)st.tap * SL * CT * 2 * 64;
        wh[(st.chunk + 2) % 3][st.sl & 1] = wp[0];
        wl[(st.chunk + 2) % 3][st.sl & 1] = wp[64];
    }
    acc[st.tile] = mfma16(xh[I % 3], wh[buf][st.sl & 1], acc[st.tile]);
    acc[st.tile] = mfma16(xh[I % 3], wl[buf][st.sl & 1], acc[st.tile]);
    acc[st.tile] = mfma16(xl[I % 3], wh[buf][st.sl & 1], acc[st.tile]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < TowerSched16<GROUP, SL>::NSTEP)
        tower16_step<G, GROUP, SL, CT, ROW_BYTES, PLANE_BYTES, I + 1>(acc, lds, abyte, vmask, zero_byte, w, wh, wl, xh, xl, rbase);
}

// One 3x3 convolution of the tower.  w: this lane's pointer to (tap 0, slab 0) of the layer for the wave's column tile;
// buffers 0 and 1 hold the layer's first two chunks on entry and the next layer's on exit.
template <class G, int GROUP, int SL, int CT, int ROW_BYTES, int PLANE_BYTES>
__device__ __forceinline__ void tower16_layer(f32x16 (&acc)[4], const unsigned char *lds, const int (&abyte)[4], const uint32_t (&vmask)[4],
                                              int zero_byte, const h8 *w, h8 (&wh)[3][2], h8 (&wl)[3][2]) {
    constexpr TStep16 s0 = TowerHolder16<GROUP, SL>::SC.s[0];
    static_assert((9 * (SL / 2)) % 3 == 0 && SL % 2 == 0, "chunks per layer must be a multiple of the three weight buffers");
    int rbase = abase16<G, ROW_BYTES>(abyte[s0.tile], vmask[s0.tile], s0.tap, zero_byte);
    h8 xh[3], xl[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xh[i] = *reinterpret_cast<const h8 *>(lds + rbase + i * 32);
        xl[i] = *reinterpret_cast<const h8 *>(lds + rbase + i * 32 + PLANE_BYTES);
    }
    __builtin_amdgcn_sched_barrier(0);
    tower16_step<G, GROUP, SL, CT, ROW_BYTES, PLANE_BYTES, 0>(acc, lds, abyte, vmask, zero_byte, w, wh, wl, xh, xl, rbase);
}

// NR = GEMM rows of a workgroup: 256 (six boards, tap-skipping row order; large batches), 128 (three boards) or 64 (ONE
// board: small batches, where a simulation round is this workgroup's latency).  The smaller forms keep the boards' cells in
// board order (identity row order); the sums of every output are taken in the same order in all of them (taps ascending,
// 16-deep slabs ascending, hi*hi + hi*lo + lo*hi), so the variants return the same bits.
template <class G, int CT, int WR, int ABL = 0, int NR = 256>
__global__ __launch_bounds__(64 * WR * CT, 2)
void k_resnet_forward_f16(Net16Params P, const u64 *__restrict__ mine, const u64 *__restrict__ theirs, int bstride,
                          const int32_t *__restrict__ leaf_slot, const int32_t *__restrict__ n_leaf_p,
                          float *__restrict__ out_p, float *__restrict__ out_v, int p_stride, int v_stride) {
    constexpr int CELLS = G::CELLS, COLS = G::COLS, ROWS = G::ROWS, A = G::A;
    constexpr int F16_ROWS = NR;                                 // (shadows the file-level constant: this workgroup's rows)
    constexpr int MT = NR / 32 / WR, WAVES = WR * CT, NTHREADS = 64 * WAVES;
    static_assert(MT >= 1 && MT * WR * 32 == NR, "the wave grid's row groups must tile the workgroup's rows");
    constexpr int TB = F16_ROWS / CELLS;
    constexpr int CP = 32 * CT, SL = CP / 16, RS = CP + 8;       // row stride in halves
    constexpr int ROW_BYTES = RS * 2, PLANE_BYTES = (F16_ROWS + 1) * ROW_BYTES, ZERO_BYTE = F16_ROWS * ROW_BYTES;
    constexpr int PT = (MT + CT - 1) / CT;                       // head-conv row tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char *plane_hi = smem16;                            // [257][RS] halves, then the lo plane
    _Float16 *act_hi = reinterpret_cast<_Float16 *>(smem16);
    _Float16 *act_lo = reinterpret_cast<_Float16 *>(smem16 + PLANE_BYTES);
    float *hb = reinterpret_cast<float *>(smem16);               // after the tower: [256][HB_STRIDE] head activations
    float *part = hb + F16_ROWS * HB_STRIDE;                     // [256][8], then the logits: [TB][8], or [TB][AP] + [TB] value logits
    float *tbuf = part;                                          // before that: [256][TB_STRIDE] per-tap value partials (head_pair16)
    constexpr int LOGIT_FLOATS = A <= 7 ? TB * 8 : TB * (G::AP + 8);
    constexpr int HEAD_BYTES = (F16_ROWS * HB_STRIDE + F16_ROWS * TB_STRIDE + LOGIT_FLOATS) * 4;      // (the Linear layer's partial sums reuse the tap buffer)
    constexpr int PERM_OFF = 2 * PLANE_BYTES > HEAD_BYTES ? 2 * PLANE_BYTES : HEAD_BYTES;     // behind both uses of the buffer
    int *s_perm = reinterpret_cast<int *>(smem16 + PERM_OFF);    // [256] GEMM row -> board-cell row
    u64 *s_board = reinterpret_cast<u64 *>(smem16 + PERM_OFF + F16_ROWS * 4);      // [TB][2][W]: the workgroup's positions (zero beyond the list)
    int *s_slot = reinterpret_cast<int *>(s_board + TB * 2 * G::W);                // [TB]: their slots

    const int n_leaf = *n_leaf_p;
    const int board0 = blockIdx.x * TB;
    if (board0 >= n_leaf) return;
    {   // launch gate: run only for list lengths in (lo, hi] (see the fp32 kernel)
        const int lo = (P.gate & 0xffff) * 16, hi = (int)((unsigned)P.gate >> 16) * 16;
        if (n_leaf <= lo || (hi != 0 && n_leaf > hi)) return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.flop_acc != nullptr)
        atomicAdd(P.flop_acc, (double)((n_leaf + TB - 1) / TB) * P.wg_flops);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r32 = lane & 31;
    const int wr = wave / CT, wc = wave % CT;
    STAMP16(0);

    for (int i = tid; i < RS; i += NTHREADS) { act_hi[F16_ROWS * RS + i] = (_Float16)0; act_lo[F16_ROWS * RS + i] = (_Float16)0; }
    for (int i = tid; i < F16_ROWS; i += NTHREADS) s_perm[i] = NR == 256 ? P.perm[i] : i;
    // the positions come in beside the row order (one dependent chain list entry -> bitboards instead of one per stem tile
    // behind the row order's)
    if (tid < TB * 2 * G::W) {
        const int b = tid / (2 * G::W), k = tid % (2 * G::W);
        u64 word = 0;
        if (board0 + b < n_leaf) {
            const int slot = leaf_slot[board0 + b];
            word = (k < G::W ? mine : theirs)[(size_t)slot * bstride + k % G::W];
            if (k == 0) s_slot[b] = slot;
        }
        s_board[tid] = word;
    }
    __syncthreads();
    STAMP16(1);

    auto row_geom = [&](int g, uint32_t &vm, int &ab, int &r) {     // g = GEMM row; r = its board-cell row in LDS
        r = s_perm[g];
        const int b = r / CELLS, pos = r % CELLS, y = pos / COLS, x = pos % COLS;
        const bool real = r < TB * CELLS && board0 + b < n_leaf;
        uint32_t m = 0;
        if (real)
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS) m |= 1u << t;
            }
        vm = m;
        ab = r * ROW_BYTES + half * 16;
    };
    auto need_of = [&](uint32_t vm) -> uint32_t {            // wave-uniform OR of the rows' tap masks
        uint32_t nd = 0;
        for (int t = 0; t < 9; ++t) if (__ballot((vm >> t) & 1u)) nd |= 1u << t;
        return nd;
    };
    int abyte[MT], lrow[MT];
    uint32_t vmask[MT], need[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        row_geom(wr * (MT * 32) + mt * 32 + r32, vmask[mt], abyte[mt], lrow[mt]);
        need[mt] = need_of(vmask[mt]);
    }

    f32x16 acc[MT], xres[MT];
    const int ch0 = wc * 32;                                      // first output channel of this wave's column tile

    // writes relu(acc*inv) as split fp16 planes (scaled by 2^ACT_SHIFT); optionally keeps the fp32 copy, in the same scale
    // (inv is a power of two, so scaling rounds nowhere; the block's second convolution starts from residual * up / 2^ACT_SHIFT).
    auto store_act = [&](const f32x16 (&a)[MT], float inv, bool keep) {
        const float inv_s = inv * ACT_SCALE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            unsigned char *dst = smem16 + lrow[mt] * ROW_BYTES + (ch0 + 4 * half) * 2;     // + 16 bytes per run of four channels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4v as;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = __builtin_amdgcn_fmed3f(a[mt][4 * i + j] * inv_s, 0.0f, 65504.0f);      // ReLU and the fp16 range in one
                    if (keep) xres[mt][4 * i + j] = t;               // (the residual stays in the planes' scale: see `up` below)
                    as[j] = t;
                }
                const h4 hi = __builtin_convertvector(as, h4);
                const h4 lo = __builtin_convertvector(as - __builtin_convertvector(hi, f32x4v), h4);
                if ((ABL & 4) && i > 0) continue;
                *reinterpret_cast<h4 *>(dst + 16 * i) = hi;
                *reinterpret_cast<h4 *>(dst + 16 * i + PLANE_BYTES) = lo;
            }
        }
    };

    // ---- the tower's forms, and its first weights (requested before the stem: an L2 round trip off the tower's start) ----
    // 64 channels on the 2 x 2 wave grid: the statically scheduled tower (nine taps per layer, so the two weight
    // buffers swap roles from layer to layer: the layer's parity is a template parameter)
    constexpr bool STATIC16 = (ABL & 7) == 0 && NR == 256 && WR == 2 && (CT == 2 || CT == 4);
    // above 128 channels a tap's weights for one column tile (SL x 8 registers) no longer fit beside the accumulators and the
    // residual: those nets stream them two slabs ahead (conv16: same order of sums per output -- taps, then slabs, ascending)
    constexpr bool WIDE = CT > 4;
    // weight slabs in flight in the ring form (a divisor of SL)
    constexpr int WIDE_PD = SL % 4 == 0 ? 4 : (SL % 5 == 0 ? 5 : (SL % 7 == 0 ? 7 : 2));
    h8 wh[3][2], wl[3][2];
    if constexpr (STATIC16) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {           // the first two chunks of layer 0
            const h8 *wp = P.conv_w + (size_t)wc * 2 * 64 + lane + ((size_t)sl * CT) * 2 * 64;
            wh[sl / 2][sl & 1] = wp[0];
            wl[sl / 2][sl & 1] = wp[64];
        }
    }

    // ---- stem: A fragments straight from the bitboards (values 0 / 2^ACT_SHIFT: lo plane is zero) ------
    {
        const f32x16 bias = channel_vec16(P.stem_b + ch0, half);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias;
        // the 27 binary inputs of a row (K index kk = tap * 3 + plane; planes: empty, mine, theirs) as one bit mask, taps
        // static; a lane then needs bits 16 sl + 8 half + j of it
        uint32_t inb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = lrow[mt];
            const int b = r / CELLS, pos = r % CELLS, y = pos / COLS, x = pos % COLS;
            typename G::B bm = board_zero<G::W>(), bt = board_zero<G::W>();
            if (r < TB * CELLS) {
#pragma unroll
                for (int k = 0; k < G::W; ++k) { bm.w[k] = s_board[b * 2 * G::W + k]; bt.w[k] = s_board[b * 2 * G::W + G::W + k]; }
            }
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int bit = G::bit_of(y + t / 3 - 1, x + t % 3 - 1);
                const uint32_t mi = board_bit(bm, bit) ? 1u : 0u, th = board_bit(bt, bit) ? 1u : 0u;
                const uint32_t three = ((mi | th) ^ 1u) | (mi << 1) | (th << 2);
                m |= ((vmask[mt] >> t) & 1u) ? three << (3 * t) : 0u;
            }
            inb[mt] = m >> (8 * half);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const h8 *wp = P.stem_w + ((size_t)sl * CT + wc) * 2 * 64 + lane;
            const h8 bh = wp[0], bl = wp[64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                h8 ah;
#pragma unroll
                for (int j = 0; j < 8; ++j) ah[j] = ((inb[mt] >> (16 * sl + j)) & 1u) ? (_Float16)ACT_SCALE : (_Float16)0;
                acc[mt] = mfma16(ah, bh, acc[mt]);
                acc[mt] = mfma16(ah, bl, acc[mt]);
            }
        }
        store_act(acc, P.stem_inv, true);
    }
    __syncthreads();
    STAMP16(2);
    if constexpr ((ABL & 8) != 0) { if (tid == 0) P.dbg[(size_t)blockIdx.x * 64 + 56] = __builtin_amdgcn_s_memtime(); }   // core-clock counter

    // ---- residual blocks ---------------------------------------------------------------------------------
    for (int layer = 0; layer < 2 * P.blocks; ++layer) {
        const bool second = layer & 1;
        const f32x16 bias = channel_vec16(P.conv_b + (size_t)layer * P.conv_b_stride + ch0, half);
        const float up = P.conv_up[layer] * (1.0f / ACT_SCALE);      // the residual is kept scaled by 2^ACT_SHIFT
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mt][v] = second ? fmaf(xres[mt][v], up, bias[v]) : bias[v];
        if constexpr (STATIC16) {
            const h8 *wq = P.conv_w + (size_t)layer * P.conv_w_stride + (size_t)wc * 2 * 64 + lane;
            const int zb = ZERO_BYTE + half * 16;
            if (wr == 0) tower16_layer<G, 0, SL, CT, ROW_BYTES, PLANE_BYTES>(acc, smem16, abyte, vmask, zb, wq, wh, wl);
            else tower16_layer<G, 1, SL, CT, ROW_BYTES, PLANE_BYTES>(acc, smem16, abyte, vmask, zb, wq, wh, wl);
        } else if constexpr ((ABL & 7) == 0 && !WIDE)
            conv16s<G, MT, 9, SL, CT>(acc, P.conv_w + (size_t)layer * P.conv_w_stride + (size_t)wc * 2 * 64, plane_hi, PLANE_BYTES,
                                      abyte, vmask, need, ROW_BYTES, ZERO_BYTE + half * 16, lane);
        else
            if constexpr (WIDE && MT < 4 && (ABL & 7) == 0)
                conv16r<G, MT, 9, SL, CT, WIDE_PD>(acc, P.conv_w + (size_t)layer * P.conv_w_stride + (size_t)wc * 2 * 64, plane_hi, PLANE_BYTES,
                                                   abyte, vmask, ROW_BYTES, ZERO_BYTE + half * 16, lane);
            else
                conv16<G, MT, 9, SL, CT, ABL>(acc, P.conv_w + (size_t)layer * P.conv_w_stride + (size_t)wc * 2 * 64, plane_hi, PLANE_BYTES,
                                              abyte, vmask, ROW_BYTES, ZERO_BYTE + half * 16, lane);
        if (layer < 16) STAMP16(3 + 3 * layer);
        __syncthreads();
        if (layer < 16) STAMP16(4 + 3 * layer);
        store_act(acc, P.conv_inv[layer], second);
        __syncthreads();
        if (layer < 16) STAMP16(5 + 3 * layer);
    }

    if constexpr ((ABL & 8) != 0) {
        if (tid == 0) { P.dbg[(size_t)blockIdx.x * 64 + 57] = __builtin_amdgcn_s_memtime(); P.dbg[(size_t)blockIdx.x * 64 + 58] = wall_clock64(); }
    }
    // ---- heads: policy 1x1 conv and the value 3x3 conv as a 27-column 1x1 conv (head_pair16), one 32-wide column tile
    //      each; the row tiles of a row group are dealt round-robin to its CT waves --------------------------------
    f32x16 accp[PT], acct[PT];
    int hbyte[PT], hrow0[PT], hrow[PT];
    uint32_t hmask[PT];
    {
        const f32x16 bp = channel_vec16(P.pol_b, half);
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int mt = wc + i * CT;
            hrow0[i] = mt < MT ? wr * (MT * 32) + mt * 32 : -1;
            row_geom(hrow0[i] < 0 ? F16_ROWS - 1 : hrow0[i] + r32, hmask[i], hbyte[i], hrow[i]);
            if (hrow0[i] < 0) hmask[i] = 0;
            accp[i] = bp;
#pragma unroll
            for (int v = 0; v < 16; ++v) acct[i][v] = 0.0f;
        }
    }
    head_pair16<G, PT, SL>(accp, acct, P.pol_w, P.val_w, plane_hi, PLANE_BYTES, hbyte, hmask, ZERO_BYTE + half * 16, lane);
    STAMP16(52);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT; ++i)
        if (hrow0[i] >= 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ch = cd_ch16(v, lane);                          // the lane's cell is row hrow[i]
                hb[hrow[i] * HB_STRIDE + ch] = fmaxf(accp[i][v] * P.pol_inv, 0.0f);
                tbuf[hrow[i] * TB_STRIDE + ch] = acct[i][v];
            }
        }
    __syncthreads();
    // value_from_taps: value[cell][c] = relu((bias[c] + sum over the on-board taps, ascending, of T[neighbour][tap * 3 + c]) * inv)
    for (int item = tid; item < TB * CELLS * VAL_CH; item += NTHREADS) {
        const int r = item / VAL_CH, c = item % VAL_CH, pos = r % CELLS, y = pos / COLS, x = pos % COLS;
        float sum = P.val_b[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < ROWS && xx >= 0 && xx < COLS) sum += tbuf[(r + (t / 3 - 1) * COLS + (t % 3 - 1)) * TB_STRIDE + t * VAL_CH + c];
        }
        hb[r * HB_STRIDE + POL_CH + c] = fmaxf(sum * P.val_inv, 0.0f);
    }
    __syncthreads();
    STAMP16(53);

    // ---- Linear layers, masked softmax, tanh: identical to the fp32 kernel --------------------------------
    float *logits = part + F16_ROWS * TB_STRIDE;
    if constexpr (A <= 7) {
        // Linear layers: thread (channel slice ks of KS, cell pos) takes its 32 / KS policy channels for ALL the workgroup's boards
        // -- a weight row is loaded once per workgroup, not once per board --, slice 0 the three value channels too; partial sums
        // [ks][board][pos][8] over the dead tap buffer; then thread (board, output, ks) adds its 42 cells and the KS slices of an
        // output meet in neighbouring lanes (fixed order).
        constexpr int KS = 4, CS = POL_CH / KS;
        static_assert(KS * TB * CELLS * 8 <= F16_ROWS * TB_STRIDE && KS * TB * 8 <= NTHREADS, "partial sums fit the tap buffer");
        for (int item = tid; item < KS * CELLS; item += NTHREADS) {
            const int ks = item / CELLS, pos = item % CELLS;
            float pa[TB][8];
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int o = 0; o < 8; ++o) pa[b][o] = 0.0f;
#pragma unroll
            for (int cc = 0; cc < CS; ++cc) {
                const int c = ks * CS + cc;
                const float4 w0 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8);
                const float4 w1 = *reinterpret_cast<const float4 *>(P.fcp_w + ((size_t)c * CELLS + pos) * 8 + 4);
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    const float h = hb[(b * CELLS + pos) * HB_STRIDE + c];
                    pa[b][0] = fmaf(h, w0.x, pa[b][0]); pa[b][1] = fmaf(h, w0.y, pa[b][1]); pa[b][2] = fmaf(h, w0.z, pa[b][2]); pa[b][3] = fmaf(h, w0.w, pa[b][3]);
                    pa[b][4] = fmaf(h, w1.x, pa[b][4]); pa[b][5] = fmaf(h, w1.y, pa[b][5]); pa[b][6] = fmaf(h, w1.z, pa[b][6]);
                }
            }
            if (ks == 0) {
#pragma unroll
                for (int c = 0; c < VAL_CH; ++c) {
                    const float wv = P.fcv_w[c * CELLS + pos];
#pragma unroll
                    for (int b = 0; b < TB; ++b) pa[b][7] = fmaf(hb[(b * CELLS + pos) * HB_STRIDE + POL_CH + c], wv, pa[b][7]);
                }
            }
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int o = 0; o < 8; ++o) part[((ks * TB + b) * CELLS + pos) * 8 + o] = pa[b][o];
        }
        __syncthreads();
        if (tid < TB * 8 * KS) {
            const int ks = tid % KS, o = (tid / KS) & 7, b = tid / (8 * KS);
            float sum = 0.0f;
            for (int pos = 0; pos < CELLS; ++pos) sum += part[((ks * TB + b) * CELLS + pos) * 8 + o];
            sum += __shfl_xor(sum, 1, KS);
            sum += __shfl_xor(sum, 2, KS);
            if (ks == 0) logits[b * 8 + o] = sum + (o < 7 ? (o < A ? P.fcp_b[o] : 0.0f) : P.fcv_b[0]);
        }
        __syncthreads();
        // masked softmax + tanh on eight lanes per board, as in the fp32 kernel (one lane per action, the eighth holds the value
        // logit; the denominator is summed in action order)
        if (tid < TB * 8) {
            const int b = tid >> 3, a = tid & 7;
            const bool liveb = board0 + b < n_leaf;
            const int slot = liveb ? s_slot[b] : 0;
            typename G::B bmine, btheirs;
#pragma unroll
            for (int k = 0; k < G::W; ++k) { bmine.w[k] = s_board[b * 2 * G::W + k]; btheirs.w[k] = s_board[b * 2 * G::W + G::W + k]; }
            const bool leg = liveb && a < A && G::legal(bmine, btheirs, a);
            const float x = logits[b * 8 + a];
            float mx = leg ? x : -INFINITY;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 8));
            const float ex = leg ? expf(x - mx) : 0.0f;
            float den = 0.0f;
#pragma unroll
            for (int k = 0; k < A; ++k) den += __shfl(ex, k, 8);
            if (liveb) {
                out_p[(size_t)slot * p_stride + a] = leg ? ex / den : 0.0f;
                if (a == 7) out_v[(size_t)slot * v_stride] = tanhf(x);
            }
        }
    } else {
        // wide action spaces (Gomoku: 81), as in the fp32 kernel: the 32 input channels of an output are cut into KS slices (a
        // slice walks its CELLS inputs per channel in flatten order), one thread per (slice, board, output), weights stored
        // [input][AP]; the slices are added in order after the barrier; then one wave per board, two actions per lane.
        constexpr int AP = G::AP, NOUT = TB * (A + 1);        // per board: A policy logits + the value logit
        constexpr int KSMAX = NTHREADS / NOUT;
        constexpr int KS = KSMAX >= 8 ? 8 : (KSMAX >= 4 ? 4 : (KSMAX >= 2 ? 2 : 1));
        constexpr int CS = POL_CH / KS;
        static_assert(KS * NOUT <= F16_ROWS * 8 && A <= 128 && AP <= 128 && TB <= WAVES, "partial sums fit; one wave per board");
        float *vl = logits + TB * AP;
#pragma unroll 1
        for (int t = tid; t < KS * NOUT; t += NTHREADS) {
            const int ks = t / NOUT, item = t % NOUT, b = item / (A + 1), o = item % (A + 1);
            float sum = 0.0f;
            if (o < A) {
#pragma unroll 1
                for (int c = ks * CS; c < (ks + 1) * CS; ++c) {
                    const float *ap = hb + (size_t)b * CELLS * HB_STRIDE + c;
                    const float *wp = P.fcp_w + (size_t)c * CELLS * AP + o;
#pragma unroll 9
                    for (int pos = 0; pos < CELLS; ++pos) sum = fmaf(ap[pos * HB_STRIDE], wp[(size_t)pos * AP], sum);
                }
            } else if (ks == 0) {
#pragma unroll 1
                for (int c = 0; c < VAL_CH; ++c)
#pragma unroll 3
                    for (int pos = 0; pos < CELLS; ++pos)
                        sum = fmaf(hb[(b * CELLS + pos) * HB_STRIDE + POL_CH + c], P.fcv_w[c * CELLS + pos], sum);
            }
            part[t] = sum;
        }
        __syncthreads();
        for (int item = tid; item < NOUT; item += NTHREADS) {
            const int b = item / (A + 1), o = item % (A + 1);
            float sum = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sum += part[ks * NOUT + item];
            if (o < A) logits[b * AP + o] = sum + P.fcp_b[o];
            else vl[b] = sum + P.fcv_b[0];
        }
        __syncthreads();
        const int b = tid >> 6;
        if (b < TB && board0 + b < n_leaf) {
            const int slot = s_slot[b];
            typename G::B bmine, btheirs;
#pragma unroll
            for (int k = 0; k < G::W; ++k) { bmine.w[k] = s_board[b * 2 * G::W + k]; btheirs.w[k] = s_board[b * 2 * G::W + G::W + k]; }
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
                if (a < AP) out_p[(size_t)slot * p_stride + a] = leg[i] ? ex[i] / den : 0.0f;
            }
            if (lane == 0) out_v[(size_t)slot * v_stride] = tanhf(vl[b]);
        }
    }
    STAMP16(55);
}

// ------------------------------------------------------------------------------------------------
// host: scaling, splitting, permutation
// ------------------------------------------------------------------------------------------------
static int pow2_scale_for(const std::vector<float> &w) {       // k with max|w| * 2^k in [2^13, 2^14)
    float mx = 0.0f;
    for (float x : w) mx = fmaxf(mx, fabsf(x));
    if (!(mx > 0.0f) || !isfinite(mx)) return 0;
    int e;
    frexpf(mx, &e);                  // mx = m * 2^e, m in [0.5, 1)
    return 14 - e;
}

// dst[((((tap*SL + sl)*CTW + ct)*2 + plane)*64 + lane)*8 + j] = split(W[oc = ct*32 + (lane&31)][ic = 16 sl + 8 (lane>>5) + j][tap] * 2^k)
static void permute_split(const std::vector<float> &w, int oc, int ic, int taps, int SL, int CTW, int k, _Float16 *dst) {
    const float sc = ldexpf(1.0f, k);
    for (int tap = 0; tap < taps; ++tap)
        for (int sl = 0; sl < SL; ++sl)
            for (int ct = 0; ct < CTW; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = ct * 32 + (lane & 31), i = 16 * sl + 8 * (lane >> 5) + j;
                        float val = 0.0f;
                        if (o < oc && i < ic) val = w[((size_t)o * ic + i) * taps + tap] * sc;
                        const _Float16 hi = (_Float16)val;
                        const _Float16 lo = (_Float16)(val - (float)hi);
                        const size_t base = ((((size_t)tap * SL + sl) * CTW + ct) * 2) * 64 * 8;
                        dst[base + (size_t)lane * 8 + j] = hi;
                        dst[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
                    }
}

// Row orders for tap skipping (see build_row_order in azsp_resnet.hip for the idea): the 252 cells of 6 boards + 4
// padding rows as 8 row tiles of 32 -- left column, right column, top rows, bottom rows (each lacking 3 of the 9 taps)
// and 4 tiles of interior cells.  Row group wr of a WR x CT wave grid owns tiles [wr*MT, (wr+1)*MT): the special
// tiles are dealt round-robin to the row groups so that every wave skips the same number of tile-taps.
static void build_row_order16(int WR, int32_t *perm) {
    constexpr int R = Connect4::ROWS, C = Connect4::COLS, CE = Connect4::CELLS, TBB = 6;
    std::vector<int> cls[4], interior, rest;       // 0 left, 1 right, 2 top, 3 bottom
    for (int b = 0; b < TBB; ++b)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x) {
                const int r = b * CE + y * C + x;
                const bool corner_l = x == 0 && (y == 0 || y == R - 1) && b < 2;
                if (x == 0 && !corner_l) cls[0].push_back(r);
                else if (x == 0) cls[y == 0 ? 2 : 3].push_back(r);
                else if (x == C - 1) cls[1].push_back(r);
                else if (y == 0) cls[2].push_back(r);
                else if (y == R - 1) cls[3].push_back(r);
                else interior.push_back(r);
            }
    while (cls[1].size() > 32) { rest.push_back(cls[1].back()); cls[1].pop_back(); }
    for (int r = TBB * CE; r < F16_ROWS; ++r) rest.push_back(r);
    interior.insert(interior.end(), rest.begin(), rest.end());      // 128 rows
    const int MT = 8 / WR;
    int next_int = 0, g = 0;
    for (int wr = 0; wr < WR; ++wr) {
        int filled = 0;
        for (int sp = 0; sp < 4; ++sp)
            if (sp % WR == wr && filled < MT) { for (int i = 0; i < 32; ++i) perm[g++] = cls[sp][i]; ++filled; }
        for (; filled < MT; ++filled) for (int i = 0; i < 32; ++i) perm[g++] = interior[next_int++];
    }
}

ResNet16Dev *resnet16_create(int game, int blocks, int channels) {
    if ((game != 0 && game != 2) || blocks < 0 || channels <= 0) return nullptr;      // Connect-4, Gomoku 9x9
    const int CT = (channels + 31) / 32;
    if (CT < 1 || CT > 8) return nullptr;      // up to 256 channels (two fp16 planes of 128 rows x 264 halves: 136 KB of LDS)
    ResNet16Dev *n = new ResNet16Dev();
    n->game = game; n->CT = CT; n->C = channels; n->blocks = blocks;
    n->cells = game == 0 ? Connect4::CELLS : Gomoku9::CELLS;
    n->A = game == 0 ? Connect4::A : Gomoku9::A;
    n->AS = game == 0 ? 8 : Gomoku9::AP;
    const int SL = 2 * CT, CP = 32 * CT;
    size_t off = 0;
    auto reserve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    n->o_stem_w = reserve((size_t)2 * CT * 2 * 64 * 16);
    n->conv_w_stride = (size_t)9 * SL * CT * 2 * 64;             // in h8 units
    n->conv_b_stride = CP;
    n->o_conv_w = reserve(n->conv_w_stride * 16 * 2 * blocks + 16);
    n->o_pol_w = reserve((size_t)SL * 2 * 64 * 16);
    n->o_val_w = reserve((size_t)9 * SL * 2 * 64 * 16);
    n->o_stem_b = reserve(CP * 4);
    n->o_conv_b = reserve((size_t)CP * 4 * 2 * blocks + 4);
    n->o_pol_b = reserve(32 * 4);
    n->o_val_b = reserve(32 * 4);
    n->o_fcp_w = reserve((size_t)POL_CH * n->cells * n->AS * 4);
    n->o_fcp_b = reserve((size_t)n->AS * 4);
    n->o_fcv_w = reserve((size_t)VAL_CH * n->cells * 4);
    n->o_fcv_b = reserve(4 * 4);
    n->o_up = reserve((size_t)(2 * blocks + 1) * 4);
    n->o_inv = reserve((size_t)(2 * blocks + 1) * 4);
    n->bytes = off;
    if (hipMalloc((void **)&n->d_blob, off) != hipSuccess) { delete n; return nullptr; }
    std::vector<int32_t> perm(4 * F16_ROWS);
    for (int k = 0; k < 4; ++k) build_row_order16(1 << k, perm.data() + (size_t)k * F16_ROWS);
    if (hipMalloc((void **)&n->d_perm, perm.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(n->d_perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(n->d_blob);
        delete n;
        return nullptr;
    }
    return n;
}

void resnet16_destroy(ResNet16Dev *n) {
    if (!n) return;
    if (n->d_blob) (void)hipFree(n->d_blob);
    if (n->d_perm) (void)hipFree(n->d_perm);
    delete n;
}

int resnet16_load(ResNet16Dev *n, const float *blob, size_t n_floats, hipStream_t s) {
    (void)n_floats;
    const int CT = n->CT, SL = 2 * CT, C = n->C;
    std::vector<unsigned char> host(n->bytes, 0);
    std::vector<float> w, b;
    auto H = [&](size_t o) { return reinterpret_cast<_Float16 *>(host.data() + o); };
    auto F = [&](size_t o) { return reinterpret_cast<float *>(host.data() + o); };
    const float *p = blob;
    {   // stem: K index kk = tap*3 + plane (27 real, padded to 32 = 2 slabs); input planes are 0 / 2^ACT_SHIFT
        ConvBN c = take_conv_bn(p, C, 3, 3);
        fold(c, w, b);
        const int k = pow2_scale_for(w);
        const float sc = ldexpf(1.0f, k);
        _Float16 *dst = H(n->o_stem_w);
        for (int sl = 0; sl < 2; ++sl)
            for (int ct = 0; ct < CT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 16 * sl + 8 * (lane >> 5) + j, o = ct * 32 + (lane & 31);
                        float val = 0.0f;
                        if (kk < 27 && o < C) val = w[((size_t)o * 3 + kk % 3) * 9 + kk / 3] * sc;
                        const _Float16 hi = (_Float16)val, lo = (_Float16)(val - (float)hi);
                        const size_t base = (((size_t)sl * CT + ct) * 2) * 64 * 8;
                        dst[base + (size_t)lane * 8 + j] = hi;
                        dst[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
                    }
        for (int o = 0; o < C; ++o) F(n->o_stem_b)[o] = b[o] * ldexpf(1.0f, ACT_SHIFT + k);
        n->stem_inv = ldexpf(1.0f, -(ACT_SHIFT + k));
    }
    for (int l = 0; l < 2 * n->blocks; ++l) {
        ConvBN c = take_conv_bn(p, C, C, 3);
        fold(c, w, b);
        const int k = pow2_scale_for(w);
        permute_split(w, C, C, 9, SL, CT, k, H(n->o_conv_w) + (size_t)l * n->conv_w_stride * 8);
        for (int o = 0; o < C; ++o) F(n->o_conv_b)[(size_t)l * n->conv_b_stride + o] = b[o] * ldexpf(1.0f, ACT_SHIFT + k);
        F(n->o_up)[l] = ldexpf(1.0f, ACT_SHIFT + k);
        F(n->o_inv)[l] = ldexpf(1.0f, -(ACT_SHIFT + k));
    }
    {
        ConvBN c = take_conv_bn(p, POL_CH, C, 1);
        fold(c, w, b);
        const int k = pow2_scale_for(w);
        permute_split(w, POL_CH, C, 1, SL, 1, k, H(n->o_pol_w));
        for (int o = 0; o < POL_CH; ++o) F(n->o_pol_b)[o] = b[o] * ldexpf(1.0f, ACT_SHIFT + k);
        n->pol_inv = ldexpf(1.0f, -(ACT_SHIFT + k));
        const float *fw = take(p, (size_t)n->A * POL_CH * n->cells), *fb = take(p, n->A);
        for (int c2 = 0; c2 < POL_CH; ++c2)
            for (int pos = 0; pos < n->cells; ++pos)
                for (int o = 0; o < n->A; ++o)
                    F(n->o_fcp_w)[((size_t)c2 * n->cells + pos) * n->AS + o] = fw[(size_t)o * POL_CH * n->cells + (size_t)c2 * n->cells + pos];
        for (int o = 0; o < n->A; ++o) F(n->o_fcp_b)[o] = fb[o];
    }
    {
        ConvBN c = take_conv_bn(p, VAL_CH, C, 3);
        fold(c, w, b);
        const int k = pow2_scale_for(w);
        std::vector<float> w1((size_t)9 * VAL_CH * C);          // [o2 = tap * 3 + c][input channel]: the 3x3 conv as a 27-column 1x1 conv
        for (int t = 0; t < 9; ++t)
            for (int c2 = 0; c2 < VAL_CH; ++c2)
                for (int i = 0; i < C; ++i) w1[((size_t)(t * VAL_CH + c2)) * C + i] = w[((size_t)c2 * C + i) * 9 + t];
        permute_split(w1, 9 * VAL_CH, C, 1, SL, 1, k, H(n->o_val_w));
        for (int o = 0; o < VAL_CH; ++o) F(n->o_val_b)[o] = b[o] * ldexpf(1.0f, ACT_SHIFT + k);
        n->val_inv = ldexpf(1.0f, -(ACT_SHIFT + k));
        const float *fw = take(p, (size_t)VAL_CH * n->cells), *fb = take(p, 1);
        for (int i = 0; i < VAL_CH * n->cells; ++i) F(n->o_fcv_w)[i] = fw[i];
        F(n->o_fcv_b)[0] = fb[0];
    }
    if (hipMemcpyAsync(n->d_blob, host.data(), host.size(), hipMemcpyHostToDevice, s) != hipSuccess) return -3;
    if (hipStreamSynchronize(s) != hipSuccess) return -4;
    return 0;
}

template <class G, int CT, int WR, int ABL = 0, int NR = 256>
static void launch16(const Net16Params &P, const u64 *mine, const u64 *theirs, int bstride, const int32_t *leaf_slot, const int32_t *n_leaf,
                     int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s,
                     hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    constexpr int F16_ROWS = NR;
    constexpr int WAVES = WR * CT, RS = 32 * CT + 8, TB = F16_ROWS / G::CELLS;
    constexpr size_t planes = (size_t)2 * (F16_ROWS + 1) * RS * 2;
    constexpr size_t logit_floats = G::A <= 7 ? TB * 8 : TB * (G::AP + 8);
    constexpr size_t heads = ((size_t)F16_ROWS * HB_STRIDE + (size_t)F16_ROWS * TB_STRIDE + logit_floats) * 4;
    static_assert((planes > heads ? planes : heads) + F16_ROWS * sizeof(int32_t) + TB * 2 * G::W * 8 + TB * 4 <= 160 * 1024, "LDS of one workgroup");
    const size_t lds = (planes > heads ? planes : heads) + F16_ROWS * sizeof(int32_t) + TB * 2 * G::W * 8 + TB * 4;   // + the row-order table, the positions, their slots
    // the attribute is per device and idempotent: a racing second call sets the same value
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resnet_forward_f16<G, CT, WR, ABL, NR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int gate_hi = (int)((unsigned)P.gate >> 16) * 16;
    const int cap = gate_hi != 0 && gate_hi < max_leaves ? gate_hi : max_leaves;
    const int grid = (cap + TB - 1) / TB;
    Net16Params Q = P;
    Q.perm = P.perm + (size_t)(WR == 1 ? 0 : WR == 2 ? 1 : WR == 4 ? 2 : 3) * 256;           // row order of this wave grid (NR = 256 only)
    Q.wg_flops = P.tower_row_flops * NR * (NR == 256 ? 5.0 / 6.0 : 1.0) + P.rest_board_flops * TB;      // (only the six-board row order skips taps)
    hipExtLaunchKernelGGL((k_resnet_forward_f16<G, CT, WR, ABL, NR>), dim3(grid), dim3(64 * WAVES), lds, s, ev_start, ev_stop, 0, Q, mine, theirs,
                          bstride, leaf_slot, n_leaf, out_p, out_v, p_stride, v_stride);
}

// Executed fp16 MFMA FLOP: three MFMAs (hi*hi, hi*lo, lo*hi) per fp32 multiply-add, over the GEMM rows and padded channels a
// workgroup really computes.  tower_row: the tower per GEMM row; rest: stem + heads per board.
static void flops16(const ResNet16Dev *n, double flops_per_board, double *tower_row, double *rest_board) {
    const double C = n->C, cells = n->cells, CP = 32.0 * n->CT;
    const double tower = 2.0 * cells * 2.0 * n->blocks * C * C * 9;
    *tower_row = 3.0 * tower / cells * (CP / C) * (CP / C);
    *rest_board = 3.0 * (flops_per_board - tower);
}
// per board, for the form a launch of this capacity takes (Connect-4 <= 128 channels, large batches: six boards on 256 rows
// in the tap-skipping row order; three boards on 128 rows; one board on 64; Gomoku: one board on 96 rows)
double resnet16_exec_flops_per_board(const ResNet16Dev *n, double flops_per_board, int max_leaves) {
    double tr, rb;
    flops16(n, flops_per_board, &tr, &rb);
    if (n->game != 0) return tr * 96.0 + rb;
    if (n->CT > 4) return max_leaves <= 256 ? tr * 64.0 + rb : tr * 128.0 / 3.0 + rb;
    return tr * 256.0 * (5.0 / 6.0) / 6.0 + rb;
}

// what a launch needs besides the network parameters
struct Fwd16 {
    const u64 *mine, *theirs;
    int bstride;
    const int32_t *leaf_slot, *n_leaf;
    int max_leaves;
    float *out_p, *out_v;
    int p_stride, v_stride;
    hipStream_t s;
};
template <int CT, int WR, int NR = 256, int ABL = 0, class G = Connect4>
static void go16(Net16Params &P, const Fwd16 &a, int gate_lo, int gate_hi, hipEvent_t e0, hipEvent_t e1) {
    P.gate = (gate_lo / 16) | ((gate_hi / 16) << 16);              // run only for list lengths in (lo, hi]; hi = 0: no upper end
    launch16<G, CT, WR, ABL, NR>(P, a.mine, a.theirs, a.bstride, a.leaf_slot, a.n_leaf, a.max_leaves, a.out_p, a.p_stride, a.out_v, a.v_stride,
                              a.s, e0, e1);
    P.gate = 0;
}

// 64 / 128 channels: ONE board per workgroup (64 rows) up to g_one boards, three (128 rows) up to g_mid, six beyond -- by the
// launch's capacity, or, with the evaluation cache (by_count), by the list's length on the device: the forms whose range
// starts below the capacity are all enqueued, each gated to its range (the timing events span the group of launches).
template <int CT>
static void forward_sized(Net16Params &P, const Fwd16 &a, bool by_count, hipEvent_t e0, hipEvent_t e1) {
    // tuning knob AZSP_F16_GATES="a,b" (multiples of 16; the forms return the same bits)
    static const int gate_env[2] = {[] { const char *x = getenv("AZSP_F16_GATES"); return x ? atoi(x) / 16 * 16 : -1; }(),
                                    [] { const char *x = getenv("AZSP_F16_GATES"); const char *c = x ? strchr(x, ',') : nullptr; return c ? atoi(c + 1) / 16 * 16 : -1; }()};
    // measured, stand-alone launches (round 3): 4 x 64: one board per workgroup 42-49 us up to 256 boards, 75 at 512, 105 at
    // 768; three 72-81; six 78-86 up to 768 -> one up to 512, six beyond.  9 x 128: one 256-287 us up to 256, 441 at 512;
    // three 372-398 up to 512, 475 at 768; six 465-504 up to 768, 550 at 1024 -> one up to 256, three up to 768, six beyond.
    const int g_one = gate_env[0] >= 0 ? gate_env[0] : (CT >= 4 ? 256 : 512);
    const int g_mid = gate_env[1] >= 0 ? (gate_env[1] > g_one ? gate_env[1] : g_one) : (CT >= 4 ? 768 : g_one);
    const int cap = a.max_leaves;
    // row groups of the small forms: every row group re-reads the column tile's weights, and a one-board workgroup has only
    // its own 42 cells to amortise them over -- at 128 channels two row groups are 1.2 MB per layer from L2 (measured 443 us
    // per 9 x 128 launch, as slow as the six-board form); one row group halves that.  (AZSP_F16_SMALL_WR overrides.)
    static const int swr_env = [] { const char *x = getenv("AZSP_F16_SMALL_WR"); return x ? atoi(x) : 0; }();
    const int swr = swr_env ? swr_env : (CT >= 4 ? 1 : 2);
    auto one_form = [&](int lo, int hi, hipEvent_t x0, hipEvent_t x1) { if (swr == 1) go16<CT, 1, 64>(P, a, lo, hi, x0, x1); else go16<CT, 2, 64>(P, a, lo, hi, x0, x1); };
    auto mid_form = [&](int lo, int hi, hipEvent_t x0, hipEvent_t x1) { go16<CT, 2, 128>(P, a, lo, hi, x0, x1); };      // (one row group measured no faster here)
    if (!by_count) {
        if (cap <= g_one) one_form(0, 0, e0, e1);
        else if (cap <= g_mid) mid_form(0, 0, e0, e1);
        else go16<CT, 2, 256>(P, a, 0, 0, e0, e1);
        return;
    }
    const bool one = g_one > 0, mid = g_mid > g_one && cap > g_one, big = cap > g_mid;
    const int last = big ? 2 : (mid ? 1 : 0);
    if (one) one_form(0, cap > g_one ? g_one : 0, e0, last == 0 ? e1 : nullptr);
    if (mid) mid_form(g_one, big ? g_mid : 0, one ? nullptr : e0, last == 1 ? e1 : nullptr);
    if (big) go16<CT, 2, 256>(P, a, g_mid, 0, (one || mid) ? nullptr : e0, e1);
}

// 160 .. 256 channels: the planes of 256 rows do not fit in LDS; ONE board per workgroup (64 rows) up to 256 boards, three
// (128 rows) beyond; one row group of CT waves (a second would read every weight twice).
template <int CT>
static void forward_wide(Net16Params &P, const Fwd16 &a, bool by_count, hipEvent_t e0, hipEvent_t e1) {
    constexpr int g_one = 256;
    if (!by_count) {
        if (a.max_leaves <= g_one) go16<CT, 1, 64>(P, a, 0, 0, e0, e1);
        else go16<CT, 1, 128>(P, a, 0, 0, e0, e1);
        return;
    }
    const bool mid = a.max_leaves > g_one;
    go16<CT, 1, 64>(P, a, 0, mid ? g_one : 0, e0, mid ? nullptr : e1);
    if (mid) go16<CT, 1, 128>(P, a, g_one, 0, nullptr, e1);
}

void resnet16_forward(ResNet16Dev *n, const u64 *leaf0, int bstride, const int32_t *leaf_slot, const int32_t *n_leaf,
                      int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s, bool by_count,
                      double *flop_acc, double flops_per_board, hipEvent_t ev_start, hipEvent_t ev_stop) {
    Fwd16 a{leaf0, leaf0 + (n->game == 0 ? Connect4::W : Gomoku9::W), bstride, leaf_slot, n_leaf, max_leaves, out_p, out_v, p_stride, v_stride, s};
    Net16Params P;
    unsigned char *d = n->d_blob;
    auto H = [&](size_t o) { return reinterpret_cast<const h8 *>(d + o); };
    auto F = [&](size_t o) { return reinterpret_cast<const float *>(d + o); };
    P.stem_w = H(n->o_stem_w); P.conv_w = H(n->o_conv_w); P.pol_w = H(n->o_pol_w); P.val_w = H(n->o_val_w);
    P.stem_b = F(n->o_stem_b); P.conv_b = F(n->o_conv_b); P.pol_b = F(n->o_pol_b); P.val_b = F(n->o_val_b);
    P.fcp_w = F(n->o_fcp_w); P.fcp_b = F(n->o_fcp_b); P.fcv_w = F(n->o_fcv_w); P.fcv_b = F(n->o_fcv_b);
    P.conv_up = F(n->o_up); P.conv_inv = F(n->o_inv);
    P.stem_inv = n->stem_inv; P.pol_inv = n->pol_inv; P.val_inv = n->val_inv;
    P.conv_w_stride = (int)n->conv_w_stride; P.conv_b_stride = (int)n->conv_b_stride; P.blocks = n->blocks;
    P.perm = n->d_perm;
    P.dbg = nullptr;
    P.flop_acc = flop_acc;
    P.wg_flops = 0.0;
    flops16(n, flops_per_board, &P.tower_row_flops, &P.rest_board_flops);
    P.gate = 0;
    if (n->game == 2) {      // Gomoku 9x9: one board (81 cells) on 96 GEMM rows, one row group of CT waves; any list length
        switch (n->CT) {
            case 1: go16<1, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 2: go16<2, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 3: go16<3, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 4: go16<4, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 5: go16<5, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 6: go16<6, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            case 7: go16<7, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
            default: go16<8, 1, 96, 0, Gomoku9>(P, a, 0, 0, ev_start, ev_stop); break;
        }
        return;
    }
    // wave grid WR x CT over the 256-row x C tile (tuning knob AZSP_F16_WR: forces the six-board form on that grid)
    static const int wr_env = [] { const char *e = getenv("AZSP_F16_WR"); return e ? atoi(e) : 0; }();
    switch (n->CT) {
        case 1:
            if (wr_env == 8 || (!wr_env && max_leaves <= 768)) go16<1, 8>(P, a, 0, 0, ev_start, ev_stop);
            else go16<1, 4>(P, a, 0, 0, ev_start, ev_stop);
            break;
        case 2: {
            bool print = false;
            if (resnet_trace_wanted(&print)) {     // diagnostic instantiation with the fp32 kernel's stamps: phase times, in-kernel clock
                constexpr int TBB = 256 / Connect4::CELLS;
                const int grid = (max_leaves + TBB - 1) / TBB;
                unsigned long long *dd = nullptr;
                const size_t cnt = (size_t)grid * 64;
                if (hipMalloc((void **)&dd, cnt * 8) == hipSuccess) {
                    (void)hipMemsetAsync(dd, 0, cnt * 8, s);
                    Net16Params Q = P;
                    Q.dbg = dd;
                    go16<2, 2, 256, 8>(Q, a, 0, 0, nullptr, nullptr);
                    (void)hipStreamSynchronize(s);
                    std::vector<unsigned long long> h(cnt);
                    (void)hipMemcpy(h.data(), dd, cnt * 8, hipMemcpyDeviceToHost);
                    (void)hipFree(dd);
                    resnet_trace_report(h.data(), grid, print);      // keeps clock and span for azsp_probe_clock
                    break;
                }
            }
#ifdef AZSP_DIAG_BUILD   /* build.py --diag only (csrc/libazsp_diag.so): never part of the product library */
            static const int abl = [] { const char *e = getenv("AZSP_F16_ABLATE"); return e ? atoi(e) : 0; }();
            if (abl) {   // timing-only diagnostic builds (wrong results): 1 = no weight streaming, 2 = no LDS fragment traffic, 4 = no activation stores
                if (abl == 1) go16<2, 2, 256, 1>(P, a, 0, 0, nullptr, nullptr);
                else if (abl == 2) go16<2, 2, 256, 2>(P, a, 0, 0, nullptr, nullptr);
                else if (abl == 3) go16<2, 2, 256, 3>(P, a, 0, 0, nullptr, nullptr);
                else go16<2, 2, 256, 4>(P, a, 0, 0, nullptr, nullptr);
                break;
            }
#endif
            if (wr_env == 4) go16<2, 4>(P, a, 0, 0, ev_start, ev_stop);
            else if (wr_env == 2) go16<2, 2>(P, a, 0, 0, ev_start, ev_stop);
            else forward_sized<2>(P, a, by_count, ev_start, ev_stop);
            break;
        }
        case 3: go16<3, 2>(P, a, 0, 0, ev_start, ev_stop); break;
        case 5: forward_wide<5>(P, a, by_count, ev_start, ev_stop); break;
        case 6: forward_wide<6>(P, a, by_count, ev_start, ev_stop); break;
        case 7: forward_wide<7>(P, a, by_count, ev_start, ev_stop); break;
        case 8: forward_wide<8>(P, a, by_count, ev_start, ev_stop); break;
        default:
            if (wr_env == 4) go16<4, 4>(P, a, 0, 0, ev_start, ev_stop);
            else if (wr_env == 2) go16<4, 2>(P, a, 0, 0, ev_start, ev_stop);
            else forward_sized<4>(P, a, by_count, ev_start, ev_stop);
            break;
    }
}

}  // namespace azsp
