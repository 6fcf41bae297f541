// Host side of the fused ResNet inference kernel: geometry, BatchNorm folding, weight permutation, the table of
// compiled kernel variants and the choice among them.  The kernel itself is csrc/azsp_resnet_kernel.h; its
// instantiations are compiled in azsp_resnet_k1..k4.hip.
#include <limits.h>

#include "azsp_resnet_kernel.h"

namespace azsp {

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

// B-fragment order for a KxK conv: dst[((tap*S + s)*64 + lane)*NTW + nt] = W[oc = nt*32 + (lane&31)][ic = koff(s) + 4*(lane>>5)][tap]
static void permute_conv(const std::vector<float> &w, int oc, int ic, int k, int S, int NTW, float *dst) {
    const int taps = k * k;
    for (int tap = 0; tap < taps; ++tap)
        for (int s = 0; s < S; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int nt = 0; nt < NTW; ++nt) {
                    const int o = nt * 32 + (lane & 31), i = koff(s) + 4 * (lane >> 5);      // K order: see koff()
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
    // third table (at 512): THREE boards in eight 16-row tiles for t16m_layer: 0 = left column (12 non-corner cells + the
    // top-left and bottom-left corners of boards 0 and 1), 1 = 16 right-column cells, 2 = the 15 inner cells of the top rows
    // + board 2's top-left corner, 3 = the same at the bottom, 4..7 = the 60 interior cells, the two right-column cells left
    // over and the padding rows 126, 127
    {
        std::vector<int> l16, r16, t16, b16, in16;
        for (int b = 0; b < 3; ++b)
            for (int y = 0; y < R; ++y)
                for (int x = 0; x < C; ++x) {
                    const int r = b * CE + y * C + x;
                    if (x == 0) {
                        if ((y == 0 || y == R - 1) && b == 2) (y == 0 ? t16 : b16).push_back(r);
                        else l16.push_back(r);
                    } else if (x == C - 1) {
                        if (r16.size() < 16) r16.push_back(r); else in16.push_back(r);
                    } else if (y == 0) t16.push_back(r);
                    else if (y == R - 1) b16.push_back(r);
                    else in16.push_back(r);
                }
        in16.push_back(126);
        in16.push_back(127);
        int32_t *m = perm + 512;
        for (int i = 0; i < 16; ++i) { m[i] = l16[i]; m[16 + i] = r16[i]; m[32 + i] = t16[i]; m[48 + i] = b16[i]; }
        for (int i = 0; i < 64; ++i) m[64 + i] = in16[i];
    }
    // second table: 2 x 2 wave grid, a row group = 4 row tiles: [left, right, interior, interior | top, bottom, interior, interior]
    int32_t *q = perm + 256;
    for (int grp = 0; grp < 2; ++grp) {
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 32; ++i) q[grp * 128 + k * 32 + i] = (*special[grp * 2 + k])[i];
        for (int i = 0; i < 64; ++i) q[grp * 128 + 64 + i] = interior[grp * 64 + i];
    }
}

int resnet_forced_variant(int game, int nt);

ResNetDev *resnet_create(int game, int blocks, int channels, bool f16_split) {
    NetGeom g;
    if (!geom_for(game, blocks, channels, g)) return nullptr;
    ResNetDev *n = new ResNetDev();
    n->g = g;
    n->game = game;
    n->forced_variant = resnet_forced_variant(game, g.NT);
    n->g_words = game == 2 ? Gomoku9::W : Connect4::W;
    size_t off = 0;
    auto reserve = [&](size_t cnt) { size_t o = off; off += (cnt + 3) & ~(size_t)3; return o; };
    n->o_stem_w = reserve((size_t)14 * 64 * g.NT);
    n->o_stem_b = reserve(g.Cp);
    n->conv_w_stride = ((size_t)9 * g.S * 64 * g.NT + 3) & ~(size_t)3;
    n->conv_b_stride = g.Cp;
    n->o_conv_w = reserve(n->conv_w_stride * 2 * blocks);
    n->o_conv_b = reserve(n->conv_b_stride * 2 * blocks);
    if ((game == 0 && (g.NT == 2 || g.NT == 4)) || (game == 2 && g.NT >= 3)) { n->has_w4 = true; n->o_conv_w4 = reserve(n->conv_w_stride * 2 * blocks); }
    if (game == 0 && (g.NT == 2 || g.NT == 4)) {      // the one-board kernels on 16-row tiles: slot-ordered weights and biases
        n->has_w16 = true;
        n->o_conv_w16 = reserve(n->conv_w_stride * 2 * blocks);
        n->o_conv_b16 = reserve(n->conv_b_stride * 2 * blocks);
    }
    n->o_pol_w = reserve((size_t)g.S * 64);
    n->o_pol_b = reserve(32);
    n->o_val_w = reserve((size_t)9 * g.Cp * 4);
    n->o_val_w4 = reserve((size_t)9 * g.Cp * 4);
    n->o_val_b = reserve(32);
    n->o_fcp_w = reserve((size_t)POL_CH * g.cells * g.AP);
    n->o_fcp_b = reserve(g.AP);
    n->o_fcv_w = reserve((size_t)VAL_CH * g.cells);
    n->o_fcv_b = reserve(4);
    off += (size_t)2 * 8 * 8 * 64 * 4;      // slack: the weight streams prefetch one block / chunk past their array
    n->blob_floats = off;
    if (hipMalloc((void **)&n->d_blob, off * sizeof(float)) != hipSuccess) { delete n; return nullptr; }
    if (hipMalloc((void **)&n->d_flop_acc, sizeof(double)) != hipSuccess || hipMemset(n->d_flop_acc, 0, sizeof(double)) != hipSuccess) {
        (void)hipFree(n->d_blob); delete n; return nullptr;
    }
    if (game == 0) {
        int32_t perm[640];
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
    if (n->d_flop_acc) (void)hipFree(n->d_flop_acc);
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
        if (n->has_w4 && n->game == 2) {
            // 16-row-tile tower (conv_tile16): [tap][16-channel input group][16-channel output tile][lane][4] with
            // lane = (output channel & 15) + 16 * quarter, element j = input channel 16 * group + 4 * quarter + j
            float *dst = host.data() + n->o_conv_w4 + (size_t)l * n->conv_w_stride;
            const int G16 = g.Cp / 16;
            for (int tap = 0; tap < 9; ++tap)
                for (int sg = 0; sg < G16; ++sg)
                    for (int ct = 0; ct < G16; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j) {
                                const int o = ct * 16 + (lane & 15), i = 16 * sg + 4 * (lane >> 4) + j;
                                dst[((((size_t)tap * G16 + sg) * G16 + ct) * 64 + lane) * 4 + j] =
                                    (o < g.C && i < g.C) ? w[((size_t)o * g.C + i) * 9 + tap] : 0.0f;
                            }
        } else if (n->has_w4) {     // [tap][s / 4][nt][lane][s & 3] from [tap][s][lane][nt]
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
        if (n->has_w16) {
            // t16_layer: [tap][16-channel input group][16-channel output tile][lane][4]; lane = (A row m = output SLOT of the
            // tile) + 16 * quarter; element j = input channel 16 * group + t16_chan(quarter, j); slot m = channel t16_chan(m >> 2, m & 3)
            float *dst = host.data() + n->o_conv_w16 + (size_t)l * n->conv_w_stride;
            const int G16 = g.Cp / 16;
            for (int tap = 0; tap < 9; ++tap)
                for (int sg = 0; sg < G16; ++sg)
                    for (int ct = 0; ct < G16; ++ct)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j) {
                                const int m = lane & 15, q = lane >> 4;
                                const int o = ct * 16 + t16_chan(m >> 2, m & 3), i = 16 * sg + t16_chan(q, j);
                                dst[((((size_t)tap * G16 + sg) * G16 + ct) * 64 + lane) * 4 + j] =
                                    (o < g.C && i < g.C) ? w[((size_t)o * g.C + i) * 9 + tap] : 0.0f;
                            }
            float *bd = host.data() + n->o_conv_b16 + (size_t)l * n->conv_b_stride;
            for (int sl = 0; sl < g.Cp; ++sl) {
                const int o = (sl & ~15) + t16_chan((sl & 15) >> 2, sl & 3);
                bd[sl] = o < g.C ? b[o] : 0.0f;
            }
        }
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

// ------------------------------------------------------------------------------------------------
// tracing (diagnostic instantiation of the headline kernel)
// ------------------------------------------------------------------------------------------------
// Tracing is on for a thread while AZSP_RESNET_TRACE is set or resnet_trace_next() armed it for one launch
// (azsp_probe_clock); the last traced launch's in-kernel clock and span are kept for that probe.
static thread_local bool g_trace_armed = false;
static thread_local double g_trace_ghz = 0.0, g_trace_span_us = 0.0;
void resnet_trace_next() { g_trace_armed = true; g_trace_ghz = 0.0; g_trace_span_us = 0.0; }
void resnet_trace_result(double *ghz, double *span_us) { if (ghz) *ghz = g_trace_ghz; if (span_us) *span_us = g_trace_span_us; }

bool resnet_trace_wanted(bool *print) {
    static const bool trace_env = getenv("AZSP_RESNET_TRACE") != nullptr;
    const bool armed = g_trace_armed;
    g_trace_armed = false;
    *print = trace_env;
    return trace_env || armed;
}

// stamps: [grid][64] 100 MHz timestamps of wave 0 (k_resnet_forward STAMP ids), 56/57 = core-clock counter around the
// tower.  Keeps the in-kernel clock (mean over workgroups) and the launch span; prints the mean phase durations of the
// first three traced launches (AZSP_RESNET_TRACE) or of every armed launch (AZSP_PROBE_VERBOSE).
void resnet_trace_report(const unsigned long long *h, int grid, bool print) {
    {
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
    static std::atomic<int> calls{0};
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
    fprintf(stderr, " | tower clock %.3f GHz\n", g_trace_ghz);
}

// ------------------------------------------------------------------------------------------------
// variant table
// ------------------------------------------------------------------------------------------------
// Every variant returns the same bits for a position (tested: test_resnet_every_batch_regime_returns_the_same_bits);
// the choice is by game, column tiles (NT = padded channels / 32) and the launch's board capacity -- measured
// crossovers, DESIGN.md section 7: up to about one workgroup per CU a launch lasts as long as ONE workgroup, so small
// batches take one board per workgroup, mid-size ones three, large ones six (tap skipping).  Stand-alone launches, round 3
// (one-board kernels on 16-row tiles), 64 channels: one 82 us up to 256 boards, 134-139 us up to 512, 195 us at 768;
// mid 165-173 us up to 768; big 243-253 us up to 1536.  128 channels x 8 blocks: one 0.48 / 0.94 / 1.39 ms at 256 / 512 /
// 768, mid 1.14-1.18 ms up to 768, big 1.77-1.80 ms up to 1536.
// AZSP_RESNET_VARIANT=<name> (read once, when the network is created) forces a variant of matching game and width.
struct Variant {
    const char *name;
    int game, nt;
    int max_boards;                       // largest launch capacity this entry is chosen for
    void (*launch)(const FwdArgs &);
    int perm_offset;                      // which row-order table of ResNetDev::d_perm the kernel expects
    int mt, pf, cw, rw;                   // the instantiation's row tiles per wave, prefetch form and wave grid (for the FLOP accounting)
    int max_by_count = 0;                 // upper end of its range when the variant is chosen by list length (evaluation cache: two
                                          // groups' launches in flight); 0 = max_boards
};
static const Variant VARIANTS[] = {
    {"c4_32_small", 0, 1, 768, fwd_c4_32_small, 0, 1, 1, 1, 4},   {"c4_32_big", 0, 1, INT_MAX, fwd_c4_32_big, 0, 2, 0, 1, 4},
    {"c4_64_one", 0, 2, 512, fwd_c4_64_one, 0, 1, 2, 2, 2},
    {"c4_64_mid", 0, 2, 768, fwd_c4_64_mid, 512, 2, 3, 2, 2, 2048},   // (by length up to 1280: 4660 -> 4745 episodes/s with the cache; up to 2048, i.e. every list of a 2048-game group: 4903 -> 4947)
    {"c4_64_big", 0, 2, INT_MAX, fwd_c4_64_big, 256, 4, 0, 2, 2},
    {"c4_96", 0, 3, INT_MAX, fwd_c4_96, 0, 1, 1, 1, 4},
    {"c4_128_one", 0, 4, 512, fwd_c4_128_one, 0, 1, 2, 4, 2, 256},
    {"c4_128_mid", 0, 4, 768, fwd_c4_128_mid, 512, 2, 3, 4, 2, 1024}, // (by length 256 / 1024: 767 -> 796 episodes/s with the cache, 8 x 128)
    {"c4_128_big", 0, 4, INT_MAX, fwd_c4_128_big, 256, 4, 0, 4, 2},
    {"c4_192", 0, 6, INT_MAX, fwd_c4_192, 0, 2, 1, 3, 2},         {"c4_256", 0, 8, INT_MAX, fwd_c4_256, 0, 2, 1, 4, 2},
    {"g9_32", 2, 1, INT_MAX, fwd_g9_32, 0, 2, 0, 1, 4},           {"g9_64", 2, 2, INT_MAX, fwd_g9_64, 0, 2, 0, 1, 4},
    {"g9_96", 2, 3, INT_MAX, fwd_g9_96, 0, 3, 0, 3, 1},           {"g9_128", 2, 4, INT_MAX, fwd_g9_128, 0, 3, 0, 4, 1},
    {"g9_192", 2, 6, INT_MAX, fwd_g9_192, 0, 3, 0, 6, 1},         {"g9_256", 2, 8, INT_MAX, fwd_g9_256, 0, 3, 0, 8, 1},
};

static const Variant *pick_variant(const ResNetDev *n, int game, int max_leaves) {
    if (n->forced_variant >= 0) return &VARIANTS[n->forced_variant];
    for (const Variant &v : VARIANTS)
        if (v.game == game && v.nt == n->g.NT && max_leaves <= v.max_boards) return &v;
    return nullptr;
}

// Matrix-pipe FLOP the variant chosen for launches of capacity max_leaves executes per board slot of a full workgroup
// (bench.py's roofline numerator): the tower's GEMM rows include the padding rows of the workgroup's tile (6 boards =
// 252 of 256 rows, one board = 42 of 64) and the padded channels, the tap-skipping row order drops 3 of 18 tile-taps
// per 3x3 convolution (exact zeros), the 9x9 one-board kernels compute exactly their 81 cells (80 on MFMAs, the corner
// on the vector ALU); stem and heads are counted as in the algorithmic figure.  The split-fp16 kernel (always 6 boards,
// tap skipping) issues three fp16 MFMAs per fp32 multiply-add.
static double variant_wg_flops(const ResNetDev *n, const Variant *v, int *boards_per_wg) {
    const NetGeom &g = n->g;
    const double C = g.C, cells = g.cells;
    const double tower = 2.0 * cells * 2.0 * g.blocks * C * C * 9;
    const double rest = n->flops_per_board - tower;
    const double pad_ch = ((double)g.Cp / C) * ((double)g.Cp / C);
    const int wg_rows = v->rw * 32 * v->mt, tb = wg_rows / g.cells;
    const bool sk = v->game == 0 && ((v->pf == 0 && ((v->mt == 2 && v->rw == 4) || (v->mt == 4 && v->rw == 2 && v->cw == v->nt))) || v->pf == 3);
    const bool tile16 = v->game == 2 && v->pf == 0 && v->cw == v->nt && v->rw == 1 && v->mt == 3 && v->nt >= 3;
    const bool tile16c = v->game == 0 && v->pf == 2;                    // one Connect-4 board on three 16-row tiles
    const double rows = tile16 ? cells * tb : (tile16c ? 48.0 : (double)wg_rows);      // GEMM rows the workgroup computes
    if (boards_per_wg) *boards_per_wg = tb;
    return tower * (rows / cells) * (sk ? 5.0 / 6.0 : 1.0) * pad_ch + rest * tb;
}
double resnet_exec_flops_per_board(const ResNetDev *n, int game, int max_leaves) {
    if (!n) return 0.0;
    if (n->f16) return resnet16_exec_flops_per_board(n->f16, n->flops_per_board, max_leaves);
    const Variant *v = pick_variant(n, game, max_leaves);
    if (!v) return 0.0;
    int tb = 1;
    const double f = variant_wg_flops(n, v, &tb);
    return f / tb;
}
// executed matrix-pipe FLOP summed on the device by every launch since the last reset (exact per workgroup)
double resnet_exec_flops_total(const ResNetDev *n, bool reset) {
    if (!n || !n->d_flop_acc) return 0.0;
    double h = 0.0;
    (void)hipMemcpy(&h, n->d_flop_acc, sizeof h, hipMemcpyDeviceToHost);
    if (reset) (void)hipMemset(n->d_flop_acc, 0, sizeof(double));
    return h;
}

// AZSP_RESNET_VARIANT, resolved once per network
int resnet_forced_variant(int game, int nt) {
    const char *want = getenv("AZSP_RESNET_VARIANT");
    if (!want || !*want) return -1;
    for (size_t i = 0; i < sizeof VARIANTS / sizeof VARIANTS[0]; ++i)
        if (!strcmp(want, VARIANTS[i].name) && VARIANTS[i].game == game && VARIANTS[i].nt == nt) return (int)i;
    fprintf(stderr, "[azsp] AZSP_RESNET_VARIANT=%s names no variant for game %d with %d column tiles; ignored\n", want, game, nt);
    return -1;
}

// FUSED rounds exist for the one-board Connect-4 kernels of the exact-fp32 network at 64 / 128 channels (the large-batch
// forms measured slower than two launches per round and were retired after round 2)
static void (*fused_launcher(const ResNetDev *n, int game, int max_leaves))(const FwdArgs &) {
    if (n->f16 || game != 0 || n->forced_variant >= 0) return nullptr;
    const Variant *v = pick_variant(n, game, max_leaves);
    if (v && v->launch == fwd_c4_64_one) return fwd_c4_64_one_fused;
    if (v && v->launch == fwd_c4_128_one) return fwd_c4_128_one_fused;
    return nullptr;
}
// tab_lds: exploration-table entries the tree step stages in LDS over the one-board kernel's dead activation rows
bool resnet_has_fused(const ResNetDev *n, int game, int max_leaves, int tab_lds) {
    if (!n || !fused_launcher(n, game, max_leaves)) return false;
    if (tab_lds > 1024) return false;              // 2 * 1024 + 480 floats of tree scratch fit 65 x 68 activation floats
    // measured (one MI355X, 64 channels, 100 simulations): 106.6 -> 101.0 us per round with one game, 118.3 -> 114.2 with 8,
    // 117.9 -> 121.2 with 100 (the tree tail then delays other workgroups' launch slots): only for a few games
    return max_leaves <= 32;
}

static void forward_impl(ResNetDev *n, int game, const u64 *leaf0, int board_stride, const int32_t *leaf_slot, const int32_t *n_leaf,
                         int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s, const FusedArgs *fused,
                         bool by_count = false, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

void resnet_forward(ResNetDev *n, int game, const u64 *leaf0, int board_stride, const int32_t *leaf_slot, const int32_t *n_leaf,
                    int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s, bool by_count,
                    hipEvent_t ev_start, hipEvent_t ev_stop) {
    forward_impl(n, game, leaf0, board_stride, leaf_slot, n_leaf, max_leaves, out_p, p_stride, out_v, v_stride, s, nullptr, by_count,
                 ev_start, ev_stop);
}

// One FUSED round of a slot group: evaluates the leaves listed in leaf_slot[0 .. *n_leaf) and carries those games on
// (expansion, backup, next descent); their next leaves land in next_list / next_count, zero_count is cleared.
void resnet_forward_fused(ResNetDev *n, int game, const View &view, const u64 *leaf0, int board_stride, const int32_t *leaf_slot,
                          const int32_t *n_leaf, int32_t *next_list, int32_t *next_count, int32_t *zero_count, int max_leaves,
                          hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    FusedArgs f;
    f.v = view;
    f.next_list = next_list;
    f.next_count = next_count;
    f.zero_count = zero_count;
    forward_impl(n, game, leaf0, board_stride, leaf_slot, n_leaf, max_leaves, nullptr, 0, nullptr, 0, s, &f, false, ev_start, ev_stop);
}

static void forward_impl(ResNetDev *n, int game, const u64 *leaf0, int board_stride, const int32_t *leaf_slot, const int32_t *n_leaf,
                         int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s, const FusedArgs *fused,
                         bool by_count, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n->f16) {
        resnet16_forward(n->f16, leaf0, board_stride, leaf_slot, n_leaf, max_leaves, out_p, p_stride, out_v, v_stride, s, by_count,
                         n->d_flop_acc, n->flops_per_board, ev_start, ev_stop);
        return;
    }
    const Variant *v = pick_variant(n, game, max_leaves);
    if (!v) return;                        // resnet_create refused every geometry without a variant
    FwdArgs a;
    a.fused = fused;
    NetParams &P = a.P;
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
    P.perm = n->d_perm ? n->d_perm + v->perm_offset : nullptr;
    P.conv_w4 = n->has_w4 ? d + n->o_conv_w4 : nullptr;
    P.conv_w16 = n->has_w16 ? d + n->o_conv_w16 : nullptr;
    P.conv_b16 = n->has_w16 ? d + n->o_conv_b16 : nullptr;
    P.flop_acc = n->d_flop_acc;
    P.wg_flops = variant_wg_flops(n, v, nullptr);
    a.mine = leaf0; a.theirs = leaf0 + n->g_words; a.bstride = board_stride; a.leaf_slot = leaf_slot; a.n_leaf = n_leaf; a.max_leaves = max_leaves;
    a.out_p = out_p; a.out_v = out_v; a.p_stride = p_stride; a.v_stride = v_stride; a.s = s;
    a.ev_start = ev_start;
    a.ev_stop = ev_stop;
    if (fused) { fused_launcher(n, game, max_leaves)(a); return; }
    if (!by_count || n->forced_variant >= 0) { v->launch(a); return; }
    // by_count (evaluation cache): the list's length is known only on the device and varies from 1 to max_leaves.  Every
    // variant of this network whose range of lengths (prev, max_boards] starts below max_leaves is enqueued, gated to that
    // range: one of them does the work, the others return at once (a few microseconds of stream time each).
    // (tuning knob AZSP_RESNET_GATES="a,b": upper ends of the first two ranges, multiples of 16; the variants return the same bits)
    static const int gate_env[2] = {[] { const char *x = getenv("AZSP_RESNET_GATES"); return x ? atoi(x) / 16 * 16 : -1; }(),
                                    [] { const char *x = getenv("AZSP_RESNET_GATES"); const char *c = x ? strchr(x, ',') : nullptr; return c ? atoi(c + 1) / 16 * 16 : -1; }()};
    struct Pick { const Variant *w; int lo, hi; } picks[8];
    int prev = 0, k = 0, np = 0;
    for (const Variant &w : VARIANTS) {
        if (w.game != game || w.nt != n->g.NT) continue;
        if (prev >= max_leaves) break;
        int hi = w.max_by_count > 0 ? w.max_by_count : w.max_boards;
        if (hi != INT_MAX && k < 2 && gate_env[k] >= 0) hi = gate_env[k];
        ++k;
        if (hi <= prev) continue;                        // an empty range: this variant is never the one
        picks[np++] = {&w, prev, hi};
        prev = hi;
    }
    for (int i = 0; i < np; ++i) {                       // the timing events span the whole group of launches
        const Variant &w = *picks[i].w;
        P.perm = n->d_perm ? n->d_perm + w.perm_offset : nullptr;
        P.wg_flops = variant_wg_flops(n, &w, nullptr);
        a.gate = (picks[i].lo / 16) | ((picks[i].hi == INT_MAX ? 0 : picks[i].hi / 16) << 16);
        a.ev_start = i == 0 ? ev_start : nullptr;
        a.ev_stop = i == np - 1 ? ev_stop : nullptr;
        w.launch(a);
    }
}

}  // namespace azsp
