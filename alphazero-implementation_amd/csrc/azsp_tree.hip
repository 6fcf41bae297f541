// Tree kernels of the self-play engine (gfx950).  Compiled with -ffp-contract=off: every float64 /
// float32 operation below must round exactly like the reference's numpy/Python arithmetic.
//
// Thread mapping (from the game trait): LPG lanes cooperate on one game, each lane owning the actions
// lane, lane + LPG, ... (Connect-4: 8 lanes x 1 action, 8 games per 64-lane wavefront; Gomoku 9x9:
// 64 lanes x 2 actions, one game per wavefront); one wavefront per workgroup.  Lanes of a game cooperate
// with width-LPG shuffles; the descent path of each game is staged in LDS and flushed to HBM for the
// expand/backup kernel.
//
// Reference functions restated (src/alphazero_simple/monte_carlo_tree_search.py unless noted):
//   k_select        ucb_score :11-22, Node.select_child :63-78, descent + leaf step :129-152,
//                   terminal backup :150-152 -> MCTS.backpropagate :214-223
//   k_expand        mask/renormalise :119-121,:165-167 (np.sum order!), Node.expand :80-90, backpropagate :214-223
//   k_root_prepare  run_batch root section :112-123 (root value discarded)
//   k_finish        src/alphazero_less_simple/core/training/episode_generator.py:38-59 and
//                   Node.select_action(temperature=1) :43-61 (np.random.choice = cumsum, normalise,
//                   searchsorted side='right' on one uniform double)
//   k_emit_*        episode_generator.py:61-87 (z sign rule, slot restart, first-N-completions stop)
//
// Rule set 1 (template parameter R = 1) is the search of the reference's first-generation package,
// src/alphazero_implementation/core/search/mcts/{search.py,node.py}: all-float64 statistics, Q used without
// negation (search.py:33-40), the root's own expansion is the first simulation (:72-91), children exist for
// every legal action (:87-90), no sign flip above a terminal node (:48-58).  Same kernels, compile-time branches.
// R = 2 is rule set 0 with the float64 value sums numpy 1.x would produce (see include/azsp.h).
#include <type_traits>

#include "azsp_internal.h"

namespace azsp {

#define GAME_SWITCH(game, ...)                              \
    switch (game) {                                         \
        case 0: { using G = Connect4; __VA_ARGS__; } break; \
        case 1: { using G = Connect2; __VA_ARGS__; } break; \
        case 2: { using G = Gomoku9; __VA_ARGS__; } break;  \
        default: break;                                     \
    }
// the same with the rule set as a second compile-time parameter (R)
#define RULES_SWITCH(game, rules, ...)                                                 \
    if ((rules) == 0) { constexpr int R = 0; GAME_SWITCH(game, __VA_ARGS__) }          \
    else if ((rules) == 1) { constexpr int R = 1; GAME_SWITCH(game, __VA_ARGS__) }     \
    else { constexpr int R = 2; GAME_SWITCH(game, __VA_ARGS__) }

// a child exists: rule set 0 where prob != 0 (monte_carlo_tree_search.py:87); rule set 1 for every legal
// action, marked by prior >= 0 (-1 = no child)
template <int R>
__device__ __forceinline__ bool child_exists(double prior) { return R != 1 ? prior != 0.0 : prior >= 0.0; }

template <class G>
static inline int games_grid(int n_games) { return (n_games * G::LPG + 63) / 64; }

// the position every game starts from (all zero unless azsp_set_initial_board changed it)
template <class G>
__device__ __forceinline__ typename G::B initial_board(const u64 *words) {
    typename G::B b;
#pragma unroll
    for (int i = 0; i < G::W; ++i) b.w[i] = words[i];
    return b;
}

template <int LPG>
__device__ __forceinline__ int red_add(int x) {
#pragma unroll
    for (int m = 1; m < LPG; m <<= 1) x += __shfl_xor(x, m, LPG);
    return x;
}

// ------------------------------------------------------------------------------------------------
// root preparation: fresh tree per slot, root queued for evaluation (no backup: path_len = 0)
// ------------------------------------------------------------------------------------------------
template <class G, int R>
__global__ void k_root_prepare(View v) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) v.nl_other[0] = 0;             // the first fused expand+select round appends to the other counter
    if (g >= v.n_act) return;
    g += v.lo;
    size_t base = (size_t)g * v.C;
    int root = 0;
    if (v.reuse) {
        // tree reuse (not in the reference, which rebuilds its trees every ply): the child chosen by the last
        // k_finish becomes the root with everything below it; blocks keep being bump-allocated after the old ones
        const int nr = v.next_root[g];
        v.next_root[g] = -1;
        if (nr > 0) {
            v.root[g] = nr;
            if (v.kids[base + nr] != 0) { v.pending[g] = 0; return; }      // expanded: nothing to evaluate
            root = nr;                                                      // never visited: expand it as a root
        } else {
            v.n_blk[g] = 0;
            v.root[g] = 0;
        }
    } else {
        v.n_blk[g] = 0;
    }
    v.cnt[base + root] = 0;
    if constexpr (R == 0) v.wsum[base + root] = 0.0f; else v.wsum_d[base + root] = 0.0;
    v.kids[base + root] = 0;
    if constexpr (R != 1) v.path_len[g] = 0;       // run_batch :112-123: the root value is discarded
    else { v.path_len[g] = 1; v.path[(size_t)g * v.D] = root; }   // search.py:72-91: expanding the root IS simulation 1
    v.leaf_node[g] = root;
    board_store<G::W>(v.leaf_mine, g, board_load<G::W>(v.cur_mine, g));
    board_store<G::W>(v.leaf_theirs, g, board_load<G::W>(v.cur_theirs, g));
    v.pending[g] = 1;
}

// ------------------------------------------------------------------------------------------------
// selection: PUCT descent to the first unexpanded node, leaf step, terminal test
// ------------------------------------------------------------------------------------------------
// APPEND: the non-terminal leaves are appended to the group's leaf list here (one atomic per wavefront; the order of
// the list does not matter to the device evaluators, whose outputs do not depend on a position's place in the batch)
// instead of by the stable k_compact pass that the host-callback evaluator needs.
template <class G, int R, bool APPEND>
__device__ __forceinline__ void select_body(const View &v, int *s_path) {
    using WT = typename std::conditional<R == 0, float, double>::type;
    const WT *wsum_r = R == 0 ? (const WT *)v.wsum : (const WT *)v.wsum_d;
    WT *wsum_w = R == 0 ? (WT *)v.wsum : (WT *)v.wsum_d;
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL, GPW = 64 / LPG;
    static_assert(APL * LPG >= A, "every action needs a lane slot");
    const int lane = threadIdx.x & (LPG - 1);
    const int grp = threadIdx.x / LPG;
    const int gl = blockIdx.x * GPW + grp;
    const bool live = gl < v.n_act;
    const int g = live ? v.lo + gl : 0;
    const size_t base = (size_t)g * v.C;
    int *my_path = s_path + grp * v.D;

    // One dependent round trip per tree level: the lanes fetch N, W, P AND the first-child index of their
    // children together; the chosen child's first-child index then comes from a shuffle, and below the root
    // N_parent is not loaded but rebuilt from the children (an expanded node is visited once at expansion and
    // afterwards only on the way to one of its children: N = 1 + sum N_child).  The root's own count is read
    // up front, next to its first-child index (no dependency): whether it was counted at its expansion differs
    // between the rule sets and for a reused subtree.
    const int root = live ? (v.reuse ? v.root[g] : 0) : 0;
    int node = root, depth = 0, parent_fc = 0, action = 0;
    int fc = live ? v.kids[base + root] : 0;
    const int n_root = live ? v.cnt[base + root] : 0;
    bool walking = live;
    while (true) {
        walking = walking && fc != 0;               // Node.expanded() :35-36
        if (!__any(walking)) break;
        double p[APL];
        int n[APL], kc[APL];
        WT w[APL];
        int nsum = 0;
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            const int a = lane + k * LPG;
            const bool in = walking && a < A;
            const int child = fc + a;
            p[k] = in ? v.prior[base + child] : (R != 1 ? 0.0 : -1.0);     // lanes without an action: no child
            n[k] = in ? v.cnt[base + child] : 0;
            w[k] = in ? wsum_r[base + child] : (WT)0;
            kc[k] = in ? v.kids[base + child] : 0;
            nsum += n[k];
        }
        const int nkids = red_add<LPG>(nsum);
        const int np_ = node == root ? n_root : nkids + 1;
        double score = -INFINITY;
        int best = lane;
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            if (child_exists<R>(p[k])) {
                double q = 0.0, u;
                if constexpr (R == 0) {
                    // ucb_score :11-22.  (prior * sqrt(N_parent)) / (n + 1) in float64 [expl_tab = c * sqrt, c = 1
                    // in the reference]; -(W / n) in float32 (numpy 2.x), promoted for the sum.
                    u = (p[k] * v.expl_tab[np_]) / (double)(n[k] + 1);
                    if (n[k] > 0) {
                        const float qf = w[k] / (float)n[k];
                        q = (double)(-qf);
                    }
                } else if constexpr (R == 2) {
                    // the same expression as numpy 1.x evaluates it: W is float64 there (legacy scalar promotion
                    // of `0.0 + np.float32`), so -(W / n) is a float64 division
                    u = (p[k] * v.expl_tab[np_]) / (double)(n[k] + 1);
                    if (n[k] > 0) q = -(w[k] / (double)n[k]);
                } else {
                    // search.py:33-40: child.value + c * prior * sqrt(N) / (1 + n), left to right, all float64
                    u = ((v.c_tab[np_] * p[k]) * v.sqrt_tab[np_]) / (double)(1 + n[k]);
                    if (n[k] > 0) q = w[k] / (double)n[k];
                }
                double sc = q + u;
                if (!(sc == sc)) sc = -INFINITY;    // NaN is never '>' the running best
                if (sc > score) { score = sc; best = lane + k * LPG; }   // ascending actions within the lane
            }
        }
        // argmax, strict '>' in ascending action order == maximum score, lowest action on ties
#pragma unroll
        for (int m = 1; m < LPG; m <<= 1) {
            const double so = __shfl_xor(score, m, LPG);
            const int bo = __shfl_xor(best, m, LPG);
            if (so > score || (so == score && bo < best)) { score = so; best = bo; }
        }
        int kc_sel = kc[0];
#pragma unroll
        for (int k = 1; k < APL; ++k) if (best / LPG == k) kc_sel = kc[k];
        const int next_fc = __shfl(kc_sel, best & (LPG - 1), LPG);
        if (walking) {
            if (lane == 0) my_path[depth] = node;
            parent_fc = fc;
            action = best;
            node = fc + best;
            fc = next_fc;
            ++depth;
        }
    }
    if (live && lane == 0) my_path[depth] = node;
    const int L = depth + 1;                        // nodes on the path, root .. leaf
    __syncthreads();                                // one wave per block: orders the LDS path writes

    // leaf step (:135-145): parent's position + action, flip perspective, terminal test
    int term = TERM_NONE;
    typename G::B nm = board_zero<G::W>(), nt = board_zero<G::W>();
    if (live) {
        const int e = (parent_fc - 1) / A;
        const typename G::B pm = board_load<G::W>(v.st_mine, (size_t)g * v.E + e);
        const typename G::B pt = board_load<G::W>(v.st_theirs, (size_t)g * v.E + e);
        G::play(pm, pt, action, nm, nt);
        term = G::terminal(nm, nt);
    }
    if (live && term != TERM_NONE) {
        // terminal leaf: never expanded.  Rule set 0: value in {+1,-1,0} for the side to move at the leaf,
        // backpropagate :214-223 -- the sign alternates with distance from the leaf (to_play alternates level by
        // level).  Rule set 1 (search.py:78-80,:48-58): the utility of the player who MOVED into the leaf, added
        // unchanged to the leaf and to its parent (no flip above a terminal node), alternating from there on.
        const WT val = R != 1 ? (WT)term : (term == 0 ? (WT)0 : (WT)(-term));
        for (int k = lane; k < L; k += LPG) {
            const int nd = my_path[L - 1 - k];
            const bool neg = R != 1 ? (k & 1) : (k > 0 && ((k - 1) & 1));
            const WT dv = neg ? -val : val;
            wsum_w[base + nd] = wsum_r[base + nd] + dv;
            v.cnt[base + nd] = v.cnt[base + nd] + 1;
        }
        if (lane == 0) {
            v.pending[g] = 0;
            atomicAdd(&v.counters[1], 1ull);
            atomicAdd(&v.counters[2], (unsigned long long)L);
        }
    } else if (live) {
        for (int k = lane; k < L; k += LPG) v.path[(size_t)g * v.D + k] = my_path[k];
        if (lane == 0) {
            v.pending[g] = 1;
            v.path_len[g] = L;
            v.leaf_node[g] = node;
            board_store<G::W>(v.leaf_mine, g, nm);
            board_store<G::W>(v.leaf_theirs, g, nt);
            atomicAdd(&v.counters[2], (unsigned long long)L);
        }
    }
    if constexpr (APPEND) {
        const bool mine_to_add = live && term == TERM_NONE && lane == 0;
        const unsigned long long bal = __ballot(mine_to_add);
        const int cnt = __popcll(bal);
        int base_j = 0;
        if (cnt > 0) {
            const int leader = __ffsll((long long)bal) - 1;
            if ((int)threadIdx.x == leader) {
                base_j = atomicAdd(v.nl, cnt);
                atomicAdd(&v.counters[0], (unsigned long long)cnt);
            }
            base_j = __shfl(base_j, leader, 64);
        }
        if (mine_to_add) {
            const int j = v.lo + base_j + __popcll(bal & ((1ull << threadIdx.x) - 1ull));
            v.leaf_slot[j] = g;
            v.leaf_index[g] = j;
        }
    }
}

template <class G, int R>
__global__ __launch_bounds__(64) void k_select(View v) {
    extern __shared__ int s_path[];                 // [GPW games][D]
    select_body<G, R, false>(v, s_path);
}

// ------------------------------------------------------------------------------------------------
// stable (slot-order) compaction of the pending leaves; one workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_compact(View v) {
    __shared__ int s_sum[1024];
    const int t = threadIdx.x;
    const int per = (v.n_act + 1023) / 1024;
    const int lo = v.lo + t * per, hi = min(lo + per, v.lo + v.n_act);
    int c = 0;
    for (int g = lo; g < hi; ++g) c += v.pending[g];
    s_sum[t] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
        int x = (t >= off) ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += x;
        __syncthreads();
    }
    int pos = v.lo + s_sum[t] - c;
    for (int g = lo; g < hi; ++g)
        if (v.pending[g]) { v.leaf_index[g] = pos; v.leaf_slot[pos++] = g; }
    if (t == 1023) {
        v.nl[0] = s_sum[1023];
        atomicAdd(&v.counters[0], (unsigned long long)s_sum[1023]);
    }
}

// ------------------------------------------------------------------------------------------------
// hash evaluator
// ------------------------------------------------------------------------------------------------
template <class G>
__global__ void k_eval_hash(View v, uint64_t salt) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.nl[0]) return;
    j += v.lo;
    const int g = v.leaf_slot[j];
    const typename G::B m = board_load<G::W>(v.leaf_mine, g), t = board_load<G::W>(v.leaf_theirs, g);
    float *p = v.eval_p + (size_t)j * G::AP;
    v.eval_v[j] = hash_eval<G>(m, t, salt, p);
    for (int a = G::A; a < G::AP; ++a) p[a] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// expansion + backup of the evaluated leaves
// ------------------------------------------------------------------------------------------------
// BY_SLOT = false: lane groups are mapped to the compacted leaves j (stand-alone k_expand); true: to the slots of
// the group, a slot taking part if its last descent left a leaf awaiting expansion (fused expand + select kernel).
template <class G, int R, bool BY_SLOT>
__device__ __forceinline__ void expand_body(const View &v, double (*s_pa)[G::AP]) {
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL, GPW = 64 / LPG;
    using WT = typename std::conditional<R == 0, float, double>::type;
    WT *wsum = R == 0 ? (WT *)v.wsum : (WT *)v.wsum_d;
    const int lane = threadIdx.x & (LPG - 1);
    const int grp = threadIdx.x / LPG;
    const int jl = blockIdx.x * GPW + grp;
    bool live;
    int j, g;
    if constexpr (BY_SLOT) {
        g = jl < v.n_act ? v.lo + jl : v.lo;
        live = jl < v.n_act && v.pending[g] != 0;
        j = live ? v.leaf_index[g] : v.lo;
    } else {
        live = jl < v.nl[0];
        j = v.lo + (live ? jl : 0);
        g = v.leaf_slot[j];
    }
    const size_t base = (size_t)g * v.C;
    const typename G::B mine = board_load<G::W>(v.leaf_mine, g), theirs = board_load<G::W>(v.leaf_theirs, g);

    // action_probs * valid_moves -> float64 ; /= np.sum(action_probs)
    double pa[APL];
    bool ok[APL];
#pragma unroll
    for (int k = 0; k < APL; ++k) {
        const int a = lane + k * LPG;
        pa[k] = 0.0;
        ok[k] = live && a < A && G::legal(mine, theirs, a);
        if (live && a < A) pa[k] = (double)v.eval_p[(size_t)j * G::AP + a] * (double)(ok[k] ? 1 : 0);
        if (a < G::AP) s_pa[grp][a] = pa[k];
    }
    __syncthreads();
    double s = 0.0;
    if constexpr (R == 1) {
        // search.py:85-90: the model's probabilities are used as they come (models/games/connect4/model.py:33-40
        // already took the softmax over the legal moves); no renormalisation
        s = 1.0;
    } else if constexpr (A < 8) {                          // numpy: plain left-to-right loop for fewer than 8 elements
#pragma unroll
        for (int a = 0; a < A; ++a) s = s + s_pa[grp][a];
    } else {
        // numpy pairwise_sum (n <= 128): eight running sums over strides of 8, combined as
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail one element at a time
        constexpr int NB = A - (A % 8);
        double r = 0.0;
        if (lane < 8) {
            r = s_pa[grp][lane];
            for (int i = 8; i < NB; i += 8) r = r + s_pa[grp][i + lane];
        }
        const double r0 = __shfl(r, 0, LPG), r1 = __shfl(r, 1, LPG), r2 = __shfl(r, 2, LPG), r3 = __shfl(r, 3, LPG);
        const double r4 = __shfl(r, 4, LPG), r5 = __shfl(r, 5, LPG), r6 = __shfl(r, 6, LPG), r7 = __shfl(r, 7, LPG);
        s = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (int i = NB; i < A; ++i) s = s + s_pa[grp][i];
    }
    const int node = v.leaf_node[g];
    const int e = v.n_blk[g];                       // read by all lanes before lane 0 bumps it
    const int fc = 1 + e * A;
    const int L = v.path_len[g];
    if (live) {
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            const int a = lane + k * LPG;
            if (a >= A) continue;
            const double pr = R != 1 ? pa[k] / s : (ok[k] ? pa[k] : -1.0);
            double prior = pr;
            // optional root exploration noise (off by default; not in the reference's search, listed in its
            // src/deepmind_alphazero_pseudocode.py:285-290): prior*(1-eps) + noise*eps on the existing root children
            if (L == (R == 1 ? 1 : 0) && v.root_noise != nullptr && child_exists<R>(pr))      // a root expansion
                prior = pr * v.noise_keep + v.root_noise[(size_t)g * A + a] * v.noise_eps;
            v.cnt[base + fc + a] = 0;                   // Node.expand :80-90
            wsum[base + fc + a] = (WT)0;
            v.prior[base + fc + a] = prior;
            v.kids[base + fc + a] = 0;
        }
        // the value is the evaluator's float32 for the side to move at the leaf (rule set 1: value[state.player],
        // a float32 widened by .tolist(), search.py:91); signs alternate upwards in both rule sets
        const WT val = (WT)v.eval_v[j];
        for (int k = lane; k < L; k += LPG) {           // backpropagate :214-223 / search.py:48-58
            const int nd = v.path[(size_t)g * v.D + (L - 1 - k)];
            const WT dv = (k & 1) ? -val : val;
            wsum[base + nd] = wsum[base + nd] + dv;
            v.cnt[base + nd] = v.cnt[base + nd] + 1;
        }
    }
    __syncthreads();                                // all lanes have read n_blk
    if (live && lane == 0) {
        v.kids[base + node] = fc;
        v.blk_parent[(size_t)g * v.E + e] = node;
        board_store<G::W>(v.st_mine, (size_t)g * v.E + e, mine);
        board_store<G::W>(v.st_theirs, (size_t)g * v.E + e, theirs);
        v.n_blk[g] = e + 1;
        v.pending[g] = 0;
    }
}

template <class G, int R>
__global__ __launch_bounds__(64) void k_expand(View v) {
    __shared__ double s_pa[64 / G::LPG][G::AP];     // staged priors for numpy's pairwise summation
    expand_body<G, R, false>(v, s_pa);
}

// Expansion + backup of the leaves evaluated in round r-1 and the descent of round r in ONE launch: both are per-game
// work on the same lane mapping, and games do not interact.  The leaves found here are appended to the group's OTHER
// leaf counter (the one this kernel's expand phase read is still in use until the launch ends); the counter for the
// round after next is zeroed.  Two launches per simulation round (this one + the evaluator) instead of four.
template <class G, int R>
__global__ __launch_bounds__(64) void k_expand_select(View v) {
    __shared__ double s_pa[64 / G::LPG][G::AP];
    extern __shared__ int s_path[];
    if (blockIdx.x == 0 && threadIdx.x == 0) v.nl_other[0] = 0;      // the next fused launch appends there
    expand_body<G, R, true>(v, s_pa);
    __syncthreads();
    __threadfence();                                // this game's new children are visible to its own descent
    select_body<G, R, true>(v, s_path);
}

// compacted leaf positions as int8 cells, for the host evaluator
template <class G>
__global__ void k_leaf_cells(View v, int8_t *dst) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.nl[0]) return;
    const int g = v.leaf_slot[v.lo + j];
    board_to_cells<G>(board_load<G::W>(v.leaf_mine, g), board_load<G::W>(v.leaf_theirs, g), dst + (size_t)j * G::CELLS);
}

// ------------------------------------------------------------------------------------------------
// end of ply: pi, sample, apply, terminal test
// ------------------------------------------------------------------------------------------------
template <class G, int R>
__global__ __launch_bounds__(64) void k_finish(View v, const double *uniforms) {
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL, GPW = 64 / LPG;
    __shared__ int s_cnt[GPW][G::AP];
    __shared__ unsigned char s_ex[GPW][G::AP];
    const int lane = threadIdx.x & (LPG - 1);
    const int grp = threadIdx.x / LPG;
    const int gl = blockIdx.x * GPW + grp;
    const bool live = gl < v.n_act;
    const int g = live ? gl : 0;
    const size_t base = (size_t)g * v.C;
    const int len = v.cur_len[g];
    const size_t h = (size_t)g * v.max_plies + len;
    const int rfc = live ? v.kids[base + (v.reuse ? v.root[g] : 0)] : 1;     // the root's children
#pragma unroll
    for (int k = 0; k < APL; ++k) {
        const int a = lane + k * LPG;
        if (a >= G::AP) continue;
        int c = 0, ex = 0;
        if (live && a < A) {
            ex = child_exists<R>(v.prior[base + rfc + a]);      // root.children keys
            c = ex ? v.cnt[base + rfc + a] : 0;
        }
        s_cnt[grp][a] = c;
        s_ex[grp][a] = (unsigned char)ex;
        if (live) v.h_counts[h * G::AP + a] = c;    // episode_generator.py:45-52 (pi = counts / sum at emit)
    }
    __syncthreads();
    if (!live || lane != 0) return;

    const typename G::B mine = board_load<G::W>(v.cur_mine, g), theirs = board_load<G::W>(v.cur_theirs, g);
    board_store<G::W>(v.h_mine, h, mine);
    board_store<G::W>(v.h_theirs, h, theirs);

    // Node.select_action(temperature=1) :43-61, sequentially in child order like numpy
    double tot = 0.0;
    for (int a = 0; a < A; ++a) if (s_ex[grp][a]) tot = tot + (double)s_cnt[grp][a];      // builtin sum(): 0 + d0 + ...
    double last = 0.0;
    for (int a = 0; a < A; ++a) if (s_ex[grp][a]) last = last + (double)s_cnt[grp][a] / tot;   // np.cumsum -> cdf[-1]
    const double u = uniforms[g];
    int act = -1, last_ex = 0;
    double run = 0.0;
    for (int a = 0; a < A; ++a)
        if (s_ex[grp][a]) {
            last_ex = a;
            run = run + (double)s_cnt[grp][a] / tot;
            if (act < 0 && !(run / last <= u)) act = a;                       // searchsorted(side='right')
        }
    if (act < 0) act = last_ex;
    // optional: after `sampling_plies` plies of a game the move is the arg-max of the visit counts
    // (Node.select_action(temperature=0), :49-50: first maximum in child order); off (< 0) by default
    if (v.sampling_plies >= 0 && len >= v.sampling_plies) {
        int best = -1, bc = -1;
        for (int a = 0; a < A; ++a)
            if (s_ex[grp][a] && s_cnt[grp][a] > bc) { bc = s_cnt[grp][a]; best = a; }
        act = best;
    }

    typename G::B nm, nt;
    G::play(mine, theirs, act, nm, nt);             // episode_generator.py:55-58
    const int term = G::terminal(nm, nt);           // :59 reward for the player now to move
    board_store<G::W>(v.cur_mine, g, nm);
    board_store<G::W>(v.cur_theirs, g, nt);
    v.cur_len[g] = len + 1;
    v.chosen[g] = act;
    v.h_action[h] = (uint8_t)act;
    if (v.reuse) v.next_root[g] = term == TERM_NONE ? rfc + act : -1;
    v.done[g] = term != TERM_NONE;
    v.reward[g] = (int8_t)(term == TERM_NONE ? 0 : term);
}

// number of games that ended this ply (before any cap), for the two-phase play of the sharded exact mode
__global__ __launch_bounds__(1024) void k_count_done(View v, int32_t *out) {
    __shared__ int s_c[1024];
    const int t = threadIdx.x;
    int c = 0;
    for (int g = t; g < v.n_act; g += 1024) c += v.done[g];
    s_c[t] = c;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (t < off) s_c[t] += s_c[t + off];
        __syncthreads();
    }
    if (t == 0) *out = s_c[0];
}

// first-N-completions bookkeeping (episode_generator.py:78-87); one workgroup, slot order.
// `cap` >= 0: emit at most `cap` of this ply's completions (slot order) -- the caller owns the stop rule
// (sharded exact mode); cap < 0: the engine's own first-n_slots rule.
__global__ __launch_bounds__(1024) void k_emit_scan(View v, PlyResultDev *res, long long cap) {
    __shared__ int s_ep[1024];
    __shared__ int s_sm[1024];
    __shared__ int s_last;
    const int t = threadIdx.x;
    const int per = (v.n_act + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, v.n_act);
    const long long ep0 = v.totals[0], sm0 = v.totals[1];
    long long remaining = cap >= 0 ? cap : (long long)v.n_slots - ep0;
    if (remaining > (long long)v.cap_eps - ep0) remaining = (long long)v.cap_eps - ep0;   // never overrun the arena
    if (t == 0) s_last = -1;
    int ce = 0;
    for (int g = lo; g < hi; ++g) ce += v.done[g];
    s_ep[t] = ce;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? s_ep[t - off] : 0;
        __syncthreads();
        s_ep[t] += x;
        __syncthreads();
    }
    int rank = s_ep[t] - ce;                        // completions before this thread's first slot
    int cs = 0;
    for (int g = lo; g < hi; ++g)
        if (v.done[g]) { if (rank < remaining) cs += v.cur_len[g]; ++rank; }
    s_sm[t] = cs;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? s_sm[t - off] : 0;
        __syncthreads();
        s_sm[t] += x;
        __syncthreads();
    }
    rank = s_ep[t] - ce;
    long long soff = sm0 + s_sm[t] - cs;
    for (int g = lo; g < hi; ++g) {
        int off = -1;
        if (v.done[g]) {
            if (rank < remaining) {
                off = (int)soff;
                const long long e = ep0 + rank;
                v.ep_slot[e] = g;
                v.ep_ply[e] = (int)v.totals[2];
                v.ep_offset[e + 1] = soff + v.cur_len[g];
                soff += v.cur_len[g];
                if (rank == remaining - 1) s_last = g;      // the N-th completion: generation stops here
            }
            ++rank;
        }
        v.emit_off[g] = off;
    }
    __syncthreads();
    if (t == 1023) {
        const long long total_done = s_ep[1023];
        const long long counted = total_done < remaining ? total_done : remaining;
        const long long ep1 = ep0 + counted, sm1 = sm0 + s_sm[1023];
        res->ply = (int)v.totals[2];
        res->n_finished = (int)counted;
        res->episodes_total = ep1;
        res->samples_total = sm1;
        res->finished = cap >= 0 ? 0 : ep1 >= v.n_slots;
        res->uniforms_used = (cap < 0 && ep1 >= v.n_slots && s_last >= 0) ? s_last + 1 : v.n_act;
        res->last_counted_slot = (total_done >= remaining && remaining > 0) ? s_last : -1;   // slot of the cap-th completion
        v.totals[0] = ep1;
        v.totals[1] = sm1;
        v.totals[2] = v.totals[2] + 1;
    }
}

// copy finished episodes into the arena and restart their slots
template <class G>
__global__ __launch_bounds__(64) void k_emit_copy(View v) {
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL, GPW = 64 / LPG;
    const int lane = threadIdx.x & (LPG - 1);
    const int g = blockIdx.x * GPW + threadIdx.x / LPG;
    if (g >= v.n_act || !v.done[g]) return;
    const int len = v.cur_len[g];
    const int off = v.emit_off[g];
    const int rew = v.reward[g];
    if (off >= 0) {
        for (int q = 0; q < len; ++q) {
            const size_t h = (size_t)g * v.max_plies + q;
            int c[APL], part = 0;
#pragma unroll
            for (int k = 0; k < APL; ++k) {
                const int a = lane + k * LPG;
                c[k] = a < A ? v.h_counts[h * G::AP + a] : 0;
                part += c[k];
            }
            const int sum = red_add<LPG>(part);
            // pi = counts / np.sum(counts)  (episode_generator.py:45-48): int -> float64 divide
#pragma unroll
            for (int k = 0; k < APL; ++k) {
                const int a = lane + k * LPG;
                if (a < A) v.out_policies[(size_t)(off + q) * A + a] = (double)c[k] / (double)sum;
            }
            const typename G::B m = board_load<G::W>(v.h_mine, h), t = board_load<G::W>(v.h_theirs, h);
            for (int cell = lane; cell < G::CELLS; cell += LPG) {
                const int b = G::bit_of(cell / G::COLS, cell % G::COLS);
                v.out_states[(size_t)(off + q) * G::CELLS + cell] = (int8_t)(board_bit(m, b) ? 1 : (board_bit(t, b) ? -1 : 0));
            }
            // value = reward * (-1)**(hist_player != final_player)  (:73-74); players alternate
            if (lane == 0) {
                v.out_values[off + q] = ((len - q) & 1) ? -rew : rew;
                v.out_actions[off + q] = v.h_action[h];
            }
        }
    }
    if (lane == 0) {                                // :80-82 fresh game in this slot
        board_store<G::W>(v.cur_mine, g, initial_board<G>(v.init_mine));
        board_store<G::W>(v.cur_theirs, g, initial_board<G>(v.init_theirs));
        v.cur_len[g] = 0;
    }
}

template <class G>
__global__ void k_reset_slots(View v) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) { v.totals[0] = 0; v.totals[1] = 0; v.totals[2] = 0; v.ep_offset[0] = 0; }
    if (g >= v.n_slots) return;
    board_store<G::W>(v.cur_mine, g, initial_board<G>(v.init_mine));
    board_store<G::W>(v.cur_theirs, g, initial_board<G>(v.init_theirs));
    v.cur_len[g] = 0;
    v.done[g] = 0;
    v.pending[g] = 0;
    if (v.reuse) { v.next_root[g] = -1; v.root[g] = 0; }
}

template <class G>
__global__ void k_set_boards(View v, const u64 *mine, const u64 *theirs, int n) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    board_store<G::W>(v.cur_mine, g, board_load<G::W>(mine, g));
    board_store<G::W>(v.cur_theirs, g, board_load<G::W>(theirs, g));
    v.cur_len[g] = 0;
    if (v.reuse) v.next_root[g] = -1;
}

template <class G, int R>
__global__ void k_root_stats(View v, int32_t *counts, float *child_w, double *prior, float *root_w, int32_t *root_n,
                             double *child_wd, double *root_wd) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n_act * G::A) return;
    const int g = i / G::A, a = i % G::A;
    const size_t base = (size_t)g * v.C + (v.reuse ? v.root[g] : 0);      // the root node
    const int rfc = v.kids[base] - (v.reuse ? v.root[g] : 0);             // its children, relative to it
    const double p = v.prior[base + rfc + a];
    const bool ex = child_exists<R>(p);
    const double w = ex ? (R == 0 ? (double)v.wsum[base + rfc + a] : v.wsum_d[base + rfc + a]) : 0.0;
    if (counts) counts[i] = ex ? v.cnt[base + rfc + a] : 0;
    if (child_w) child_w[i] = (float)w;
    if (child_wd) child_wd[i] = w;
    if (prior) prior[i] = ex ? p : 0.0;
    if (a == 0) {
        const double rw = R == 0 ? (double)v.wsum[base] : v.wsum_d[base];
        if (root_w) root_w[g] = (float)rw;
        if (root_wd) root_wd[g] = rw;
        if (root_n) root_n[g] = v.cnt[base];
    }
}

// packed records for the multi-GPU all-gather: int8 state[cells] | int8 value | pad to 8 | double policy[A]
template <class G>
__global__ void k_export_records(View v, long long first, long long n, unsigned char *dst) {
    constexpr int HEAD = ((G::CELLS + 1 + 7) / 8) * 8;
    constexpr int REC = HEAD + 8 * G::A;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned char *r = dst + i * REC;
    const long long s = first + i;
    for (int c = 0; c < G::CELLS; ++c) r[c] = (unsigned char)v.out_states[s * G::CELLS + c];
    r[G::CELLS] = (unsigned char)(int8_t)v.out_values[s];
    for (int c = G::CELLS + 1; c < HEAD; ++c) r[c] = 0;
    double *pol = (double *)(r + HEAD);
    for (int a = 0; a < G::A; ++a) pol[a] = v.out_policies[s * G::A + a];
}

// training tensors straight from the arena: BaseModel.format_dataset (base_model.py:61-67) with
// ResNet._states_to_tensor (resnet.py:115-128): planes float32 [S][3][cells] = [empty, mine, theirs],
// policies float32 [S][A] (float64 -> float32 as torch.FloatTensor does), values float32 [S]
template <class G>
__global__ void k_export_dataset(View v, long long first, long long n, float *planes, float *pol, float *val) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int PER = G::CELLS > G::A ? G::CELLS : G::A;
    if (i >= n * PER) return;
    const long long smp = i / PER;
    const int cell = (int)(i % PER);
    const long long src = first + smp;
    if (cell < G::CELLS) {
        const int8_t c = v.out_states[src * G::CELLS + cell];
        float *pl = planes + smp * 3 * G::CELLS + cell;
        pl[0] = c == 0 ? 1.0f : 0.0f;
        pl[G::CELLS] = c == 1 ? 1.0f : 0.0f;
        pl[2 * G::CELLS] = c == -1 ? 1.0f : 0.0f;
    }
    if (cell < G::A) pol[smp * G::A + cell] = (float)v.out_policies[src * G::A + cell];
    if (cell == 0) val[smp] = (float)v.out_values[src];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_root_prepare(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_root_prepare<G, R>), dim3((v.n_act + 255) / 256), dim3(256), 0, s, v));
}
void launch_select(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64),
                                                    (size_t)(64 / G::LPG) * v.D * sizeof(int), s, v));
}
void launch_compact(const View &v, hipStream_t s) { hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, v); }
void launch_eval_hash(int game, const View &v, uint64_t salt, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_eval_hash<G>, dim3((v.n_act + 255) / 256), dim3(256), 0, s, v, salt));
}
void launch_expand(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expand<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64), 0, s, v));
}
void launch_expand_select(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expand_select<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64),
                                                    (size_t)(64 / G::LPG) * v.D * sizeof(int), s, v));
}
void launch_leaf_cells(int game, const View &v, int8_t *dst, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_leaf_cells<G>, dim3((v.n_act + 255) / 256), dim3(256), 0, s, v, dst));
}
void launch_finish(int game, const View &v, const double *uniforms, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_finish<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64), 0, s, v, uniforms));
}
void launch_count_done(const View &v, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_done, dim3(1), dim3(1024), 0, s, v, out);
}
void launch_emit(int game, const View &v, PlyResultDev *res, long long cap, hipStream_t s) {
    hipLaunchKernelGGL(k_emit_scan, dim3(1), dim3(1024), 0, s, v, res, cap);
    GAME_SWITCH(game, hipLaunchKernelGGL(k_emit_copy<G>, dim3(games_grid<G>(v.n_act)), dim3(64), 0, s, v));
}
void launch_reset_slots(int game, const View &v, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_reset_slots<G>, dim3((v.n_slots + 255) / 256), dim3(256), 0, s, v));
}
void launch_set_boards(int game, const View &v, const u64 *mine, const u64 *theirs, int n, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_set_boards<G>, dim3((n + 255) / 256), dim3(256), 0, s, v, mine, theirs, n));
}
void launch_root_stats(int game, const View &v, int32_t *counts, float *child_w, double *prior, float *root_w,
                       int32_t *root_n, double *child_wd, double *root_wd, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_root_stats<G, R>), dim3((v.n_act * G::A + 255) / 256), dim3(256), 0, s, v,
                                                    counts, child_w, prior, root_w, root_n, child_wd, root_wd));
}
void launch_export_records(int game, const View &v, int64_t first, int64_t n, void *dst, hipStream_t s) {
    if (n <= 0) return;
    GAME_SWITCH(game, hipLaunchKernelGGL(k_export_records<G>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v,
                                         (long long)first, (long long)n, (unsigned char *)dst));
}
void launch_export_dataset(int game, const View &v, int64_t first, int64_t n, float *planes, float *pol, float *val,
                           hipStream_t s) {
    if (n <= 0) return;
    GAME_SWITCH(game, {
        constexpr int PER = G::CELLS > G::A ? G::CELLS : G::A;
        hipLaunchKernelGGL(k_export_dataset<G>, dim3((unsigned)((n * PER + 255) / 256)), dim3(256), 0, s, v,
                           (long long)first, (long long)n, planes, pol, val);
    });
}

}  // namespace azsp
