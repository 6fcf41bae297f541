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

#include <hip/hip_ext.h>

#include "azsp_internal.h"
#include "azsp_tree_dev.h"

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

// ------------------------------------------------------------------------------------------------
// root preparation: fresh tree per slot, root queued for evaluation (no backup: path_len = 0)
// ------------------------------------------------------------------------------------------------
template <class G, int R>
__global__ void k_root_prepare(View v) {
    using T = TreeGeo<G>;
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) v.nl_other[0] = 0;             // the first fused expand+select round appends to the other counter
    if (g >= v.n_act) return;
    g += v.lo;
    SlotHead *h = reinterpret_cast<SlotHead *>(slot_ptr(v, g));
    int root = 0;
    if (v.reuse) {
        // tree reuse (not in the reference, which rebuilds its trees every ply): the child chosen by the last
        // k_finish becomes the root with everything below it; blocks keep being bump-allocated after the old ones
        const int nr = h->next_root;
        h->next_root = -1;
        if (nr > 0) {
            h->root = nr;
            h->sims_done = 0;
            if (node_chunk<G>(v, g, nr)->kid != 0) { h->pending = 0; return; }      // expanded: nothing to evaluate
            root = nr;                                                                // never visited: expand it as a root
        } else {
            h->n_blk = 0;
            h->root = 0;
        }
    } else {
        h->n_blk = 0;
        h->root = 0;
    }
    if (root == 0) {
        h->root_kid = 0;
        h->root_n = 0;
        h->root_w = 0.0f;
    } else {
        NodeChunk *c = node_chunk<G>(v, g, root);
        c->n = 0;
        c->w = 0.0f;
        c->kid = 0;
    }
    if constexpr (R != 0) v.wsum_d[(size_t)g * v.C + root] = 0.0;
    int32_t *path = reinterpret_cast<int32_t *>(slot_ptr(v, g) + T::PATH_CHUNK * 16);
    if constexpr (R != 1) h->path_len = 0;         // run_batch :112-123: the root value is discarded
    else { h->path_len = 1; path[0] = root; }      // search.py:72-91: expanding the root IS simulation 1
    h->leaf_node = root;
    h->sims_done = 0;
    const typename G::B rm = board_load<G::W>(v.cur_mine, g), rt = board_load<G::W>(v.cur_theirs, g);
    slot_store_leaf<G>(v, g, rm, rt);
    h->pending = 1;
    if constexpr (G::W == 1) {
        // evaluation cache: the root joins the group's leaf list here (as the entry to evaluate, if this game claimed
        // one) instead of through k_compact; the host zeroed the list's counter before this launch
        if (v.ent != nullptr) {
            bool owner;
            const int id = ent_claim<G>(v, g, rm, rt, owner);
            h->eval_ent = id;
            append_leaves(v, true, owner, id, v.leaf_slot, v.nl);
        }
    }
}

// dynamic LDS of the tree kernels: [GPW games][D] path staging, then the exploration table(s)
template <class G>
__device__ __forceinline__ double *tab_region(int *s_path) {
    return reinterpret_cast<double *>(s_path + (((64 / G::LPG) * TreeGeo<G>::D + 1) & ~1));
}
template <class G, int R>
static inline size_t tree_lds_bytes(const View &v) {
    return (size_t)((((64 / G::LPG) * TreeGeo<G>::D + 1) & ~1)) * sizeof(int) + (size_t)v.tab_lds * (R == 1 ? 2 : 1) * sizeof(double);
}

// the evaluator's output for game g as expand_game wants it: this lane's actions and the value
template <class G>
__device__ __forceinline__ float eval_rows(const View &v, int g, int lane, bool live, const SlotRegs<G> &st, float (&p32)[G::APL]) {
    if constexpr (G::W == 1 && G::APL == 1) {
        if (v.ent != nullptr) {                               // evaluation cache: the row sits in the entry the slot points at
            const float *row = v.ent[live ? st.eval_ent : 0].row;
            p32[0] = (live && lane < G::A) ? row[lane] : 0.0f;
            return live ? row[7] : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < G::APL; ++k) {
        const int a = lane + k * G::LPG;
        p32[k] = (live && a < G::A) ? v.eval_p[(size_t)g * v.eval_pstride + a] : 0.0f;
    }
    return live ? v.eval_v[(size_t)g * v.eval_vstride] : 0.0f;
}

template <class G, int R>
__global__ __launch_bounds__(64) void k_select(View v) {
    extern __shared__ __attribute__((aligned(16))) int s_path[];                 // [GPW games][D]
    constexpr int LPG = G::LPG, GPW = 64 / LPG;
    Tabs tb;
    stage_tabs<R>(v, tab_region<G>(s_path), tb);
    const int lane = threadIdx.x & (LPG - 1), grp = threadIdx.x / LPG;
    const int gl = blockIdx.x * GPW + grp;
    const bool live = gl < v.n_act;
    const int g = live ? v.lo + gl : v.lo;
    typename G::B cur_m = board_zero<G::W>(), cur_t = board_zero<G::W>();
    if (live) { cur_m = board_load<G::W>(v.cur_mine, g); cur_t = board_load<G::W>(v.cur_theirs, g); }
    SlotRegs<G> st;
    slot_load<G>(v, g, lane, live, st);
    __syncthreads();
    int leaf_id;
    bool leaf_add;                     // (the slot-ordered four-launch rounds run without the evaluation cache: v.ent == nullptr)
    select_game<G, R>(v, g, lane, grp, live, st, s_path, tb, cur_m, cur_t, leaf_id, leaf_add);
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
    for (int g = lo; g < hi; ++g) c += reinterpret_cast<const SlotHead *>(slot_ptr(v, g))->pending;
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
        if (reinterpret_cast<const SlotHead *>(slot_ptr(v, g))->pending) v.leaf_slot[pos++] = g;
    if (t == 1023) {
        v.nl[0] = s_sum[1023];
        atomicAdd(&v.counters[0], (unsigned long long)s_sum[1023]);
        atomicAdd(&v.counters[4], (unsigned long long)s_sum[1023]);
    }
}

// ------------------------------------------------------------------------------------------------
// hash evaluator
// ------------------------------------------------------------------------------------------------
template <class G>
__global__ void k_eval_hash(View v, uint64_t salt) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.nl[0]) return;
    const int g = v.leaf_slot[v.lo + j];
    typename G::B m, t;
    if constexpr (G::W == 1) {
        if (v.ent != nullptr) {            // evaluation cache: the list holds entries; position from and row into the entry
            EvalEnt *e = &v.ent[g];
            m.w[0] = e->mine;
            t.w[0] = e->theirs;
            float row[G::AP];
            const float val = hash_eval<G>(m, t, salt, row);
            for (int a = 0; a < G::AP; ++a) e->row[a] = a < G::A ? row[a] : 0.0f;
            e->row[7] = val;
            return;
        }
    }
    leaf_boards<G>(v, g, m, t);
    float *p = v.eval_p + (size_t)g * v.eval_pstride;
    const float val = hash_eval<G>(m, t, salt, p);
    for (int a = G::A; a < G::AP; ++a) p[a] = 0.0f;
    v.eval_v[(size_t)g * v.eval_vstride] = val;           // (last: inside a slot record the value is the row's element 7)
}

// host evaluator: results arrive in list order; the tree kernels read them by slot
template <class G>
__global__ void k_scatter_eval(View v, const float *p_compact, const float *v_compact) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.nl[0] * G::AP) return;
    const int j = i / G::AP, a = i % G::AP;
    const int g = v.leaf_slot[v.lo + j];
    if (a < G::A) v.eval_p[(size_t)g * v.eval_pstride + a] = p_compact[i];        // (the padding is never read)
    if (a == 0) v.eval_v[(size_t)g * v.eval_vstride] = v_compact[j];
}

template <class G, int R>
__global__ __launch_bounds__(64) void k_expand(View v) {
    __shared__ double s_pa[64 / G::LPG][G::AP];     // staged priors for numpy's pairwise summation
    constexpr int LPG = G::LPG, GPW = 64 / LPG;
    const int lane = threadIdx.x & (LPG - 1), grp = threadIdx.x / LPG;
    const int gl = blockIdx.x * GPW + grp;
    const bool in = gl < v.n_act;
    const int g = in ? v.lo + gl : v.lo;
    SlotRegs<G> st;
    slot_load<G>(v, g, lane, in, st);
    const bool live = in && st.pending != 0;
    float p32[G::APL];
    const float vleaf = eval_rows<G>(v, g, lane, live, st, p32);
    expand_game<G, R>(v, g, lane, grp, live, st, p32, vleaf, s_pa);
    if (live && lane == 0) slot_store_head<G>(v, g, st);
}

// Expansion + backup of the leaves evaluated in round r-1 and the descent of round r in ONE launch: both are per-game
// work on the same lane mapping, and games do not interact.  The game's slot record is loaded once, updated in
// registers by the expansion and used by the descent.  The leaves found here are appended to the group's OTHER
// leaf counter (the one this kernel's expand phase read is still in use until the launch ends); the counter for the
// round after next is zeroed.  Two launches per simulation round (this one + the evaluator) instead of four.
template <class G, int R>
__global__ __launch_bounds__(64) void k_expand_select(View v) {
    __shared__ double s_pa[64 / G::LPG][G::AP];
    extern __shared__ __attribute__((aligned(16))) int s_path[];
    constexpr int LPG = G::LPG, GPW = 64 / LPG;
    // These waves share their SIMDs with waves of other slot groups' inference launches, which keep the matrix pipe
    // (and with it the vector issue port) busy back to back and, being older, win every arbitration: measured 100 us per
    // launch under load against 22 us alone.  A game's round is a few thousand vector instructions: let them go first.
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && threadIdx.x == 0) v.nl_other[0] = 0;      // the next fused launch appends there
    Tabs tb;
    stage_tabs<R>(v, tab_region<G>(s_path), tb);
    const int lane = threadIdx.x & (LPG - 1), grp = threadIdx.x / LPG;
    const int gl = blockIdx.x * GPW + grp;
    const bool in = gl < v.n_act;
    const int g = in ? v.lo + gl : v.lo;
    // the game's current position is requested together with its slot record (the descent below starts from it)
    typename G::B cur_m = board_zero<G::W>(), cur_t = board_zero<G::W>();
    if (in) { cur_m = board_load<G::W>(v.cur_mine, g); cur_t = board_load<G::W>(v.cur_theirs, g); }
    SlotRegs<G> st;
    slot_load<G>(v, g, lane, in, st);
    const bool pend = in && st.pending != 0;
    float p32[G::APL];
    const float vleaf = eval_rows<G>(v, g, lane, pend, st, p32);
    expand_game<G, R>(v, g, lane, grp, pend, st, p32, vleaf, s_pa);
    if (pend && lane == 0) slot_store_head<G>(v, g, st);
    __syncthreads();
    // A game is expanded and descended by the same lanes of the same wave and no other workgroup touches its records
    // inside this launch, so the stores above only have to be ordered before this wave's own loads: a workgroup-scope
    // fence (s_waitcnt) -- an agent-scope fence would write back and invalidate the XCD's L2 from every wave.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int leaf_id;
    bool leaf_add;
    const int term = select_game<G, R>(v, g, lane, grp, in, st, s_path, tb, cur_m, cur_t, leaf_id, leaf_add);
    const bool need = in && term == TERM_NONE && lane == 0;
    append_leaves(v, need, need && leaf_add, leaf_id, v.leaf_slot, v.nl);
}

// listed leaf positions as int8 cells, for the host evaluator
template <class G>
__global__ void k_leaf_cells(View v, int8_t *dst) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.nl[0]) return;
    const int g = v.leaf_slot[v.lo + j];
    typename G::B m, t;
    leaf_boards<G>(v, g, m, t);
    board_to_cells<G>(m, t, dst + (size_t)j * G::CELLS);
}

// the block of the root's children: nullptr while the root is unexpanded
template <class G>
__device__ __forceinline__ const NodeChunk *root_children(const View &v, int g) {
    const SlotHead *h = reinterpret_cast<const SlotHead *>(slot_ptr(v, g));
    const int root = v.reuse ? h->root : 0;
    const int kid = root == 0 ? h->root_kid : node_chunk<G>(v, g, root)->kid;
    return kid ? reinterpret_cast<const NodeChunk *>(block_ptr(v, g, kid - 1)) : nullptr;
}
// float64 prior of child a of the block `blk` (what Node.prior holds in the reference)
template <class G, int R>
__device__ __forceinline__ double child_prior(const View &v, int g, const NodeChunk *blk, int a) {
    const BlockHead *hd = reinterpret_cast<const BlockHead *>(blk + G::A);
    if (hd->flags & 1) return v.root_prior[(size_t)g * G::A + a];
    if constexpr (R == 1) return (double)blk[a].p;
    return (double)blk[a].p / hd->psum;
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
    const int len = v.cur_len[g];
    const size_t h = (size_t)g * v.max_plies + len;
    const NodeChunk *rblk = live ? root_children<G>(v, g) : nullptr;         // the root's children
#pragma unroll
    for (int k = 0; k < APL; ++k) {
        const int a = lane + k * LPG;
        if (a >= G::AP) continue;
        int c = 0, ex = 0;
        if (rblk != nullptr && a < A) {
            ex = child_exists<R>(child_prior<G, R>(v, g, rblk, a));      // root.children keys
            c = ex ? rblk[a].n : 0;
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
    if (v.reuse) {                                  // node id of the chosen child: 1 + block * A + action
        const int rb = (int)((reinterpret_cast<const uint8_t *>(rblk) - block_ptr(v, g, 0)) / v.rec_block);
        reinterpret_cast<SlotHead *>(slot_ptr(v, g))->next_root = term == TERM_NONE ? 1 + rb * A + act : -1;
    }
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
    SlotHead *h = reinterpret_cast<SlotHead *>(slot_ptr(v, g));
    h->pending = 0;
    h->next_root = -1;
    h->root = 0;
}

template <class G>
__global__ void k_set_boards(View v, const u64 *mine, const u64 *theirs, int n) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    board_store<G::W>(v.cur_mine, g, board_load<G::W>(mine, g));
    board_store<G::W>(v.cur_theirs, g, board_load<G::W>(theirs, g));
    v.cur_len[g] = 0;
    reinterpret_cast<SlotHead *>(slot_ptr(v, g))->next_root = -1;
}

template <class G, int R>
__global__ void k_root_stats(View v, int32_t *counts, float *child_w, double *prior, float *root_w, int32_t *root_n,
                             double *child_wd, double *root_wd) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n_act * G::A) return;
    const int g = i / G::A, a = i % G::A;
    const SlotHead *h = reinterpret_cast<const SlotHead *>(slot_ptr(v, g));
    const int root = v.reuse ? h->root : 0;                             // the root node
    const NodeChunk *blk = root_children<G>(v, g);                      // its children (always expanded after a search)
    const int rb = blk ? (int)((reinterpret_cast<const uint8_t *>(blk) - block_ptr(v, g, 0)) / v.rec_block) : 0;
    const double p = blk ? child_prior<G, R>(v, g, blk, a) : 0.0;
    const bool ex = blk != nullptr && child_exists<R>(p);
    const double w = ex ? (R == 0 ? (double)blk[a].w : v.wsum_d[(size_t)g * v.C + 1 + rb * G::A + a]) : 0.0;
    if (counts) counts[i] = ex ? blk[a].n : 0;
    if (child_w) child_w[i] = (float)w;
    if (child_wd) child_wd[i] = w;
    if (prior) prior[i] = ex ? p : 0.0;
    if (a == 0) {
        const double rw = R != 0 ? v.wsum_d[(size_t)g * v.C + root] : (root == 0 ? (double)h->root_w : (double)node_chunk<G>(v, g, root)->w);
        if (root_w) root_w[g] = (float)rw;
        if (root_wd) root_wd[g] = rw;
        if (root_n) root_n[g] = root == 0 ? h->root_n : node_chunk<G>(v, g, root)->n;
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
// record sizes of the game's tree layout, for the engine's allocations
void tree_geometry(int game, int32_t *block_bytes, int32_t *slot_bytes, int32_t *path_cap, int32_t *eval_off) {
    GAME_SWITCH(game, { *block_bytes = TreeGeo<G>::BLOCK_BYTES; *slot_bytes = TreeGeo<G>::SLOT_BYTES; *path_cap = TreeGeo<G>::D;
                        *eval_off = TreeGeo<G>::EVAL_IN_SLOT ? TreeGeo<G>::EVAL_CHUNK * 16 : -1; });
}
void launch_root_prepare(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_root_prepare<G, R>), dim3((v.n_act + 255) / 256), dim3(256), 0, s, v));
}
void launch_select(int game, const View &v, hipStream_t s) {
    RULES_SWITCH(game, v.rules, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64),
                                                    (tree_lds_bytes<G, R>(v)), s, v));
}
void launch_compact(const View &v, hipStream_t s) { hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, v); }
void launch_eval_hash(int game, const View &v, uint64_t salt, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_eval_hash<G>, dim3((v.n_act + 255) / 256), dim3(256), 0, s, v, salt));
}
// (ea / eb: optional timing events that take the launch's own start / end timestamps from its dispatch packet)
void launch_expand(int game, const View &v, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
    RULES_SWITCH(game, v.rules, hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_expand<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64), 0, s, ea, eb, 0, v));
}
void launch_expand_select(int game, const View &v, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
    RULES_SWITCH(game, v.rules, hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_expand_select<G, R>), dim3(games_grid<G>(v.n_act)), dim3(64),
                                                       (uint32_t)(tree_lds_bytes<G, R>(v)), s, ea, eb, 0, v));
}
void launch_leaf_cells(int game, const View &v, int8_t *dst, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_leaf_cells<G>, dim3((v.n_act + 255) / 256), dim3(256), 0, s, v, dst));
}
void launch_scatter_eval(int game, const View &v, const float *p_compact, const float *v_compact, hipStream_t s) {
    GAME_SWITCH(game, hipLaunchKernelGGL(k_scatter_eval<G>, dim3((v.n_act * G::AP + 255) / 256), dim3(256), 0, s, v, p_compact, v_compact));
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
