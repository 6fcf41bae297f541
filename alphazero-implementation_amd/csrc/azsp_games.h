// Device/host game traits: bitboard restatements of the BaseGame plugins
// (reference: src/alphazero_simple/base_game.py:6-62).  A trait is what a Python BaseGame subclass
// registers on the device side; every function below is callable from host and device.
//
// Bit layout (shared with oracle/az_oracle.py `to_bits` and oracle/az_oracle.c):
//   cell (row r counted from the TOP, column c)  ->  bit  c*(ROWS+1) + (ROWS-1-r)
// i.e. column-major with one always-zero sentinel bit above each column, so shift-and
// line tests cannot wrap between columns.
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

// --------------------------------------------------------------------------------------------
// Connect-4, 6x7, four in a row.  src/alphazero_simple/connect4_game.py:6-101
// --------------------------------------------------------------------------------------------
struct Connect4 {
    static constexpr int ROWS = 6, COLS = 7, A = 7, CELLS = 42, H1 = 7, MAX_PLIES = 42;
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
    // get_next_state(board, player=+1, a) followed by get_canonical_board(.., -1)
    // (connect4_game.py:28-42,100-101; called at monte_carlo_tree_search.py:140-142):
    // the piece goes to the lowest empty cell of the column (none -> board unchanged), then
    // the perspective flips, so the mover's stones become `theirs`.
    static AZ_HD void play(u64 mine, u64 theirs, int a, u64 &nmine, u64 &ntheirs) {
        u64 col = ((mine | theirs) >> (a * H1)) & COL;
        u64 low0 = (col + 1) & ~col & COL;          // lowest zero bit of the 6-cell column, 0 if full
        nmine = theirs;
        ntheirs = mine | (low0 << (a * H1));
    }
    static AZ_HD bool four(u64 b) {                  // is_win, connect4_game.py:58-88
        u64 v = b & (b >> 1);  if (v & (v >> 2)) return true;      // vertical
        u64 h = b & (b >> 7);  if (h & (h >> 14)) return true;     // horizontal
        u64 d = b & (b >> 6);  if (d & (d >> 12)) return true;     // diagonal
        u64 e = b & (b >> 8);  if (e & (e >> 16)) return true;     // anti-diagonal
        return false;
    }
    // get_reward_for_player(board, +1), connect4_game.py:90-98 (win tested before draw)
    static AZ_HD int terminal(u64 mine, u64 theirs) {
        if (four(mine)) return 1;
        if (four(theirs)) return -1;
        if (~(mine | theirs) & TOP) return TERM_NONE;               // has_legal_moves :44-46
        return 0;
    }
    static AZ_HD int bit_of(int r, int c) { return c * H1 + (ROWS - 1 - r); }
};

// --------------------------------------------------------------------------------------------
// Connect2, 1x4 strip, two in a row.  src/alphazero_simple/connect2_game.py:6-80
// --------------------------------------------------------------------------------------------
struct Connect2 {
    static constexpr int ROWS = 1, COLS = 4, A = 4, CELLS = 4, H1 = 2, MAX_PLIES = 4;
    static constexpr u64 ALL = 0x55ull;

    static AZ_HD uint32_t legal_mask(u64 mine, u64 theirs) {        // :44-52
        u64 e = ~(mine | theirs) & ALL;
        return (uint32_t)((e & 1) | ((e >> 1) & 2) | ((e >> 2) & 4) | ((e >> 3) & 8));
    }
    static AZ_HD void play(u64 mine, u64 theirs, int a, u64 &nmine, u64 &ntheirs) {   // :28-36
        u64 bit = 1ull << (a * H1);                 // b[action] = player, unconditionally
        nmine = theirs & ~bit;
        ntheirs = mine | bit;
    }
    static AZ_HD bool two(u64 b) { return (b & (b >> 2)) != 0; }    // :54-66
    static AZ_HD int terminal(u64 mine, u64 theirs) {               // :68-78
        if (two(mine)) return 1;
        if (two(theirs)) return -1;
        if (~(mine | theirs) & ALL) return TERM_NONE;
        return 0;
    }
    static AZ_HD int bit_of(int r, int c) { (void)r; return c * H1; }
};

// int8 cells (row-major, row 0 = top, +1 mine / -1 theirs) <-> bitboards
template <class G>
AZ_HD void cells_to_bits(const int8_t *cells, u64 &mine, u64 &theirs) {
    mine = theirs = 0;
    for (int r = 0; r < G::ROWS; ++r)
        for (int c = 0; c < G::COLS; ++c) {
            int8_t v = cells[r * G::COLS + c];
            if (v > 0) mine |= 1ull << G::bit_of(r, c);
            else if (v < 0) theirs |= 1ull << G::bit_of(r, c);
        }
}
template <class G>
AZ_HD void bits_to_cells(u64 mine, u64 theirs, int8_t *cells) {
    for (int r = 0; r < G::ROWS; ++r)
        for (int c = 0; c < G::COLS; ++c) {
            int b = G::bit_of(r, c);
            cells[r * G::COLS + c] = (int8_t)(((mine >> b) & 1) ? 1 : (((theirs >> b) & 1) ? -1 : 0));
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
// p_out[A] (float32, exact zeros on illegal actions), returns V
template <int A>
AZ_HD float hash_eval(u64 mine, u64 theirs, uint32_t legal, u64 salt, float *p_out) {
    u64 h0 = mix64(mine ^ mix64(theirs) ^ salt);
    uint32_t w[A];
    uint32_t s = 0;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        uint32_t wa = 0;
        if ((legal >> a) & 1u) {
            u64 r = mix64(h0 + (u64)a + 1ull) >> 52;
            wa = 1u + (uint32_t)((r * r * r) >> 24);
        }
        w[a] = wa;
        s += wa;
    }
    float fs = (float)s;
#pragma unroll
    for (int a = 0; a < A; ++a) p_out[a] = (float)w[a] / fs;      // IEEE correctly-rounded f32 divide
    int vi = (int)(mix64(h0 + 0x1000ull) >> 48) - 32768;
    return (float)vi / 32768.0f;                                   // exact
}

}  // namespace azsp
