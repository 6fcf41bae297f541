// Device/host game traits: bitboard restatements of the BaseGame plugins
// (reference: src/alphazero_simple/base_game.py:6-62).  A trait is what a Python BaseGame subclass
// registers on the device side; every function below is callable from host and device.
//
// A position is two bit sets (side to move = "mine", opponent = "theirs") of W 64-bit words each.
//   Connect-4 / Connect2: cell (row r from the TOP, column c) -> bit c*(ROWS+1) + (ROWS-1-r)
//                         (column-major, one always-zero sentinel bit above each column)
//   Gomoku 9x9          : cell (r, c) -> bit r*(COLS+1) + c   (row-major, one sentinel column), 90 bits
// so shift-and line tests cannot wrap.  The same layouts are restated in oracle/az_oracle.py (`to_bits`)
// and oracle/az_oracle.c for the hash evaluator.
//
// Each trait also fixes the thread mapping of the tree kernels: LPG lanes cooperate on one game and every
// lane owns APL = ceil(A / LPG) actions (lane, lane + LPG, ...).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ __forceinline__
#else
#define AZ_HD inline
#endif

namespace azsp {

typedef uint64_t u64;

enum : int { TERM_NONE = 2 };   // terminal(): +1 / -1 / 0 reward for the side to move, or TERM_NONE

template <int W>
struct Board {
    u64 w[W];
};

template <int W>
AZ_HD Board<W> board_zero() {
    Board<W> b;
    for (int i = 0; i < W; ++i) b.w[i] = 0;
    return b;
}
template <int W>
AZ_HD Board<W> board_load(const u64 *arr, size_t idx) {
    Board<W> b;
    for (int i = 0; i < W; ++i) b.w[i] = arr[idx * W + i];
    return b;
}
template <int W>
AZ_HD Board<W> board_load_strided(const u64 *arr, size_t idx, size_t stride) {      // W words at arr + idx * stride
    Board<W> b;
    for (int i = 0; i < W; ++i) b.w[i] = arr[idx * stride + i];
    return b;
}
template <int W>
AZ_HD void board_store(u64 *arr, size_t idx, const Board<W> &b) {
    for (int i = 0; i < W; ++i) arr[idx * W + i] = b.w[i];
}
// (selects instead of runtime-indexed words: a dynamically indexed register array would go to scratch)
template <int W>
AZ_HD bool board_bit(const Board<W> &b, int bit) {
    u64 x = b.w[0];
    for (int k = 1; k < W; ++k) x = (bit >> 6) == k ? b.w[k] : x;
    return (x >> (bit & 63)) & 1ull;
}
template <int W>
AZ_HD void board_set(Board<W> &b, int bit) {
    for (int k = 0; k < W; ++k) b.w[k] |= (bit >> 6) == k ? 1ull << (bit & 63) : 0ull;
}

// --------------------------------------------------------------------------------------------
// Connect-4, 6x7, four in a row.  src/alphazero_simple/connect4_game.py:6-101
// --------------------------------------------------------------------------------------------
struct Connect4 {
    static constexpr int ROWS = 6, COLS = 7, A = 7, CELLS = 42, H1 = 7, MAX_PLIES = 42;
    static constexpr int W = 1, LPG = 8, APL = 1, AP = 8;      // words, lanes per game, actions per lane, padded A
    typedef Board<1> B;
    static constexpr u64 COL = 0x3Full;
    static constexpr u64 TOP = (1ull << 5) | (1ull << 12) | (1ull << 19) | (1ull << 26) | (1ull << 33) |
                               (1ull << 40) | (1ull << 47);

    // get_valid_moves, connect4_game.py:48-56: column c is legal iff its TOP cell is empty
    static AZ_HD uint32_t legal_mask(u64 mine, u64 theirs) {
        u64 free_top = ~(mine | theirs) & TOP;
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < COLS; ++c) m |= (uint32_t)((free_top >> (c * H1 + 5)) & 1ull) << c;
        return m;
    }
    static AZ_HD bool legal(const B &m, const B &t, int a) { return !(((m.w[0] | t.w[0]) >> (a * H1 + 5)) & 1ull); }
    // get_next_state(board, player=+1, a) followed by get_canonical_board(.., -1)
    // (connect4_game.py:28-42,100-101; called at monte_carlo_tree_search.py:140-142):
    // the piece goes to the lowest empty cell of the column (none -> board unchanged), then
    // the perspective flips, so the mover's stones become `theirs`.
    static AZ_HD void play(const B &m, const B &t, int a, B &nm, B &nt) {
        u64 col = ((m.w[0] | t.w[0]) >> (a * H1)) & COL;
        u64 low0 = (col + 1) & ~col & COL;          // lowest zero bit of the 6-cell column, 0 if full
        nm.w[0] = t.w[0];
        nt.w[0] = m.w[0] | (low0 << (a * H1));
    }
    static AZ_HD bool four(u64 b) {                  // is_win, connect4_game.py:58-88
        u64 v = b & (b >> 1);  if (v & (v >> 2)) return true;      // vertical
        u64 h = b & (b >> 7);  if (h & (h >> 14)) return true;     // horizontal
        u64 d = b & (b >> 6);  if (d & (d >> 12)) return true;     // diagonal
        u64 e = b & (b >> 8);  if (e & (e >> 16)) return true;     // anti-diagonal
        return false;
    }
    // get_reward_for_player(board, +1), connect4_game.py:90-98 (win tested before draw)
    static AZ_HD int terminal(const B &m, const B &t) {
        if (four(m.w[0])) return 1;
        if (four(t.w[0])) return -1;
        if (~(m.w[0] | t.w[0]) & TOP) return TERM_NONE;             // has_legal_moves :44-46
        return 0;
    }
    static AZ_HD int bit_of(int r, int c) { return c * H1 + (ROWS - 1 - r); }
    // Injective 49-bit key of a position in which every column is filled from the bottom without gaps (every position
    // a game can reach): mine + occupied is, per 7-bit column of height h, a number in [2^h - 1, 2^(h+1) - 2] -- the
    // ranges of different heights are disjoint, so h and with it `mine` can be read back.  Boards with a floating stone
    // (callers may hand MCTS.run_batch anything) have no key: false.
    static constexpr u64 BOTTOM = 1ull | (1ull << 7) | (1ull << 14) | (1ull << 21) | (1ull << 28) | (1ull << 35) | (1ull << 42);
    static AZ_HD bool pos_key(const B &m, const B &t, u64 &key) {
        const u64 occ = m.w[0] | t.w[0];
        key = m.w[0] + occ;
        return ((occ + BOTTOM) & occ) == 0 && (m.w[0] & t.w[0]) == 0;
    }
};

// --------------------------------------------------------------------------------------------
// Connect2, 1x4 strip, two in a row.  src/alphazero_simple/connect2_game.py:6-80
// --------------------------------------------------------------------------------------------
struct Connect2 {
    static constexpr int ROWS = 1, COLS = 4, A = 4, CELLS = 4, H1 = 2, MAX_PLIES = 4;
    static constexpr int W = 1, LPG = 8, APL = 1, AP = 8;
    typedef Board<1> B;
    static constexpr u64 ALL = 0x55ull;

    static AZ_HD bool legal(const B &m, const B &t, int a) { return !(((m.w[0] | t.w[0]) >> (a * H1)) & 1ull); }   // :44-52
    static AZ_HD void play(const B &m, const B &t, int a, B &nm, B &nt) {   // :28-36
        u64 bit = 1ull << (a * H1);                 // b[action] = player, unconditionally
        nm.w[0] = t.w[0] & ~bit;
        nt.w[0] = m.w[0] | bit;
    }
    static AZ_HD bool two(u64 b) { return (b & (b >> 2)) != 0; }    // :54-66
    static AZ_HD int terminal(const B &m, const B &t) {             // :68-78
        if (two(m.w[0])) return 1;
        if (two(t.w[0])) return -1;
        if (~(m.w[0] | t.w[0]) & ALL) return TERM_NONE;
        return 0;
    }
    static AZ_HD int bit_of(int r, int c) { (void)r; return c * H1; }
    static AZ_HD bool pos_key(const B &m, const B &t, u64 &key) { key = m.w[0] | (t.w[0] << 8); return true; }   // 4 cells: trivially injective
};

// --------------------------------------------------------------------------------------------
// Gomoku 9x9, five in a row (BASELINE.json configs[4]).  NOT in the reference: the device side of the NEW
// BaseGame plugin alphazero-implementation_amd/azsp/games.py `GomokuGame` (action = row*9 + col, an occupied
// cell leaves the board unchanged, win tested before draw -- the conventions of the reference's Connect4Game).
// --------------------------------------------------------------------------------------------
struct Gomoku9 {
    static constexpr int ROWS = 9, COLS = 9, A = 81, CELLS = 81, H1 = 10, MAX_PLIES = 81;
    static constexpr int W = 2, LPG = 64, APL = 2, AP = 88;
    typedef Board<2> B;

    static AZ_HD int bit_of(int r, int c) { return r * H1 + c; }
    static AZ_HD bool pos_key(const B &, const B &, u64 &key) { key = 0; return false; }     // 3^81 positions: no 64-bit key
    static AZ_HD bool legal(const B &m, const B &t, int a) {
        const int bit = bit_of(a / COLS, a % COLS);
        return !(board_bit(m, bit) || board_bit(t, bit));
    }
    static AZ_HD void play(const B &m, const B &t, int a, B &nm, B &nt) {
        nm = t;
        nt = m;
        const int bit = bit_of(a / COLS, a % COLS);
        if (!(board_bit(m, bit) || board_bit(t, bit))) board_set(nt, bit);
    }
    static AZ_HD B shr(const B &b, int s) {          // 128-bit logical shift right, 0 < s < 64
        B r;
        r.w[0] = (b.w[0] >> s) | (b.w[1] << (64 - s));
        r.w[1] = b.w[1] >> s;
        return r;
    }
    static AZ_HD bool five(const B &b) {
        const int dirs[4] = {1, H1, H1 + 1, H1 - 1};               // horizontal, vertical, diagonal, anti-diagonal
        for (int d = 0; d < 4; ++d) {
            const int s = dirs[d];
            B x = b;
            for (int i = 1; i < 5; ++i) {
                const B sh = shr(b, s * i);
                x.w[0] &= sh.w[0];
                x.w[1] &= sh.w[1];
            }
            if (x.w[0] | x.w[1]) return true;
        }
        return false;
    }
    static AZ_HD int terminal(const B &m, const B &t) {
        if (five(m)) return 1;
        if (five(t)) return -1;
        // any empty cell left?  occupied = 81 real cells (sentinel bits are never set)
        const u64 o0 = m.w[0] | t.w[0], o1 = m.w[1] | t.w[1];
        u64 full0 = 0, full1 = 0;
        for (int r = 0; r < ROWS; ++r)
            for (int c = 0; c < COLS; ++c) {
                const int bit = bit_of(r, c);
                if (bit < 64) full0 |= 1ull << bit; else full1 |= 1ull << (bit - 64);
            }
        if ((~o0 & full0) | (~o1 & full1)) return TERM_NONE;
        return 0;
    }
};

// int8 cells (row-major, row 0 = top, +1 mine / -1 theirs) <-> bit sets
template <class G>
AZ_HD void cells_to_board(const int8_t *cells, typename G::B &mine, typename G::B &theirs) {
    mine = board_zero<G::W>();
    theirs = board_zero<G::W>();
    for (int r = 0; r < G::ROWS; ++r)
        for (int c = 0; c < G::COLS; ++c) {
            int8_t v = cells[r * G::COLS + c];
            if (v > 0) board_set(mine, G::bit_of(r, c));
            else if (v < 0) board_set(theirs, G::bit_of(r, c));
        }
}
template <class G>
AZ_HD void board_to_cells(const typename G::B &mine, const typename G::B &theirs, int8_t *cells) {
    for (int r = 0; r < G::ROWS; ++r)
        for (int c = 0; c < G::COLS; ++c) {
            int b = G::bit_of(r, c);
            cells[r * G::COLS + c] = (int8_t)(board_bit(mine, b) ? 1 : (board_bit(theirs, b) ? -1 : 0));
        }
}

// --------------------------------------------------------------------------------------------
// Hash evaluator (definition: oracle/az_oracle.py `hash_eval_bits`)
// --------------------------------------------------------------------------------------------
AZ_HD u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <int W>
AZ_HD u64 board_fold(const Board<W> &b) {          // boards of several words are folded; identity for one word
    u64 h = b.w[0];
    for (int k = 1; k < W; ++k) h = mix64(h) ^ b.w[k];
    return h;
}
template <class G>
AZ_HD uint32_t hash_weight(u64 h0, const typename G::B &m, const typename G::B &t, int a) {
    if (!G::legal(m, t, a)) return 0u;
    const u64 r = mix64(h0 + (u64)a + 1ull) >> 52;
    return 1u + (uint32_t)((r * r * r) >> 24);
}
// p_out[a * stride] for a < A (float32, exact zeros on illegal actions), returns V
template <class G>
AZ_HD float hash_eval(const typename G::B &m, const typename G::B &t, u64 salt, float *p_out) {
    const u64 h0 = mix64(board_fold(m) ^ mix64(board_fold(t)) ^ salt);
    uint32_t s = 0;
    for (int a = 0; a < G::A; ++a) s += hash_weight<G>(h0, m, t, a);
    const float fs = (float)s;
    for (int a = 0; a < G::A; ++a) p_out[a] = (float)hash_weight<G>(h0, m, t, a) / fs;   // IEEE correctly-rounded f32 divide
    const int vi = (int)(mix64(h0 + 0x1000ull) >> 48) - 32768;
    return (float)vi / 32768.0f;                                   // exact
}

}  // namespace azsp
