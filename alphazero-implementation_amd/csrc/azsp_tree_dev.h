// Device functions of the tree search, shared by the tree kernels (azsp_tree.hip) and by the inference kernels that
// carry a game through expansion, backup and its next descent themselves (azsp_resnet_kernel.h, FUSED variants).
// Every float64 / float32 operation here must round exactly like the reference's numpy / Python arithmetic: functions
// that do such arithmetic switch floating-point contraction off for their body, whatever the translation unit's flags.
#pragma once
#include <type_traits>

#include "azsp_internal.h"

// (HIP's default is -ffp-contract=fast-honor-pragmas: the pragma governs every function body parsed below)
#pragma clang fp contract(off)

namespace azsp {

// a wave-wide rendezvous for LDS traffic between the lanes of ONE wave (the tree code runs as one-wave workgroups in
// the tree kernels and as one wave of a larger workgroup at the end of a FUSED inference kernel: no s_barrier there)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): LDS operations of a wave complete in order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a child exists: rule set 0 where prob != 0 (monte_carlo_tree_search.py:87); rule set 1 for every legal
// action, marked by prior >= 0 (-1 = no child)
template <int R>
__device__ __forceinline__ bool child_exists(double prior) { return R != 1 ? prior != 0.0 : prior >= 0.0; }


template <int LPG>
__device__ __forceinline__ int red_add(int x) {
#pragma unroll
    for (int m = 1; m < LPG; m <<= 1) x += __shfl_xor(x, m, LPG);
    return x;
}


// ------------------------------------------------------------------------------------------------
// tree storage access (layout: azsp_internal.h)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t *slot_ptr(const View &v, int g) { return v.slot + (size_t)g * v.slot_bytes; }
__device__ __forceinline__ uint8_t *block_ptr(const View &v, int g, int e) { return v.rec + (size_t)g * v.rec_slot + (size_t)e * v.rec_block; }
template <class G>
__device__ __forceinline__ NodeChunk *node_chunk(const View &v, int g, int node) {      // node >= 1
    const int e = (node - 1) / G::A, a = (node - 1) % G::A;
    return reinterpret_cast<NodeChunk *>(block_ptr(v, g, e)) + a;
}
__device__ __forceinline__ int comp4(const int4 &t, int k) { return k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w)); }
__device__ __forceinline__ double make_double(int lo, int hi) { return __hiloint2double(hi, lo); }

// The slot record of game g in registers: every lane of the game's group loads the 16-byte chunks lane, lane + LPG, ...;
// the scalar fields are then broadcast from the lanes that hold them.  `sc` keeps the raw chunks (the path lives there).
template <class G>
struct SlotRegs {
    int pending, path_len, leaf_node, n_blk, root, root_kid, root_n, next_root;
    int sims_done;                              // simulations of this ply finished so far (FUSED rounds)
    int eval_ent;                               // evaluation cache: entry holding the pending leaf's row
    float root_w;
    typename G::B lm, lt;                       // leaf position
    int4 sc[TreeGeo<G>::CPL];
};
template <class G>
__device__ __forceinline__ int4 slot_chunk_bcast(const SlotRegs<G> &st, int c) {         // c: compile-time chunk index
    constexpr int LPG = G::LPG;
    int4 r;
    const int4 &src = st.sc[c / LPG];
    r.x = __shfl(src.x, c % LPG, LPG); r.y = __shfl(src.y, c % LPG, LPG);
    r.z = __shfl(src.z, c % LPG, LPG); r.w = __shfl(src.w, c % LPG, LPG);
    return r;
}
template <class G>
__device__ __forceinline__ void slot_load(const View &v, int g, int lane, bool live, SlotRegs<G> &st) {
    using T = TreeGeo<G>;
    const int4 *src = reinterpret_cast<const int4 *>(slot_ptr(v, g));
    // the first LPG chunks (head, leaf position, first path entries) at once; the rest of the path only for a descent
    // deeper than they hold (Connect-4: 16 entries in the first 128 bytes, the other 128 bytes are rarely needed)
    st.sc[0] = (live && lane < T::SLOT_CHUNKS) ? src[lane] : make_int4(0, 0, 0, 0);
    const int4 c0 = slot_chunk_bcast<G>(st, 0), c1 = slot_chunk_bcast<G>(st, 1), c2 = slot_chunk_bcast<G>(st, 2);
#pragma unroll
    for (int k = 1; k < T::CPL; ++k) {
        const int c = lane + k * G::LPG;
        const bool need = c0.y > 4 * (k * G::LPG - T::PATH_CHUNK);            // path_len reaches into this round of chunks
        st.sc[k] = (live && need && c < T::SLOT_CHUNKS) ? src[c] : make_int4(0, 0, 0, 0);
    }
    st.pending = c0.x; st.path_len = c0.y; st.leaf_node = c0.z; st.n_blk = c0.w;
    st.root = c1.x; st.root_kid = c1.y; st.root_n = c1.z; st.root_w = __int_as_float(c1.w);
    st.next_root = c2.x;
    st.sims_done = c2.y;
    st.eval_ent = c2.z;
    if constexpr (G::W == 1) {
        const int4 b = slot_chunk_bcast<G>(st, T::LEAF_CHUNK);
        st.lm.w[0] = (u64)(uint32_t)b.x | ((u64)(uint32_t)b.y << 32);
        st.lt.w[0] = (u64)(uint32_t)b.z | ((u64)(uint32_t)b.w << 32);
    } else {
        const int4 bm = slot_chunk_bcast<G>(st, T::LEAF_CHUNK), bt = slot_chunk_bcast<G>(st, T::LEAF_CHUNK + 1);
        st.lm.w[0] = (u64)(uint32_t)bm.x | ((u64)(uint32_t)bm.y << 32); st.lm.w[1] = (u64)(uint32_t)bm.z | ((u64)(uint32_t)bm.w << 32);
        st.lt.w[0] = (u64)(uint32_t)bt.x | ((u64)(uint32_t)bt.y << 32); st.lt.w[1] = (u64)(uint32_t)bt.z | ((u64)(uint32_t)bt.w << 32);
    }
}
// path entry i (< D) of the loaded slot record, for the lane that holds it: chunk PATH_CHUNK + i / 4, component i % 4
template <class G>
__device__ __forceinline__ bool path_entry_here(int lane, int k, int j, int &i) {        // chunk lane + k * LPG, component j
    using T = TreeGeo<G>;
    const int c = lane + k * G::LPG;
    i = 4 * (c - T::PATH_CHUNK) + j;
    return c >= T::PATH_CHUNK && c < T::SLOT_CHUNKS && i < T::D;
}
// write the head chunks back (lane 0) -- the fields a kernel changed
template <class G>
__device__ __forceinline__ void slot_store_head(const View &v, int g, const SlotRegs<G> &st) {
    int4 *dst = reinterpret_cast<int4 *>(slot_ptr(v, g));
    dst[0] = make_int4(st.pending, st.path_len, st.leaf_node, st.n_blk);
    dst[1] = make_int4(st.root, st.root_kid, st.root_n, __float_as_int(st.root_w));
    dst[2] = make_int4(st.next_root, st.sims_done, st.eval_ent, 0);
}
template <class G>
__device__ __forceinline__ void slot_store_leaf(const View &v, int g, const typename G::B &m, const typename G::B &t) {
    u64 *dst = reinterpret_cast<u64 *>(slot_ptr(v, g) + TreeGeo<G>::LEAF_CHUNK * 16);
#pragma unroll
    for (int i = 0; i < G::W; ++i) { dst[i] = m.w[i]; dst[G::W + i] = t.w[i]; }
}

// exploration tables: the leading tab_lds entries sit in LDS (a table lookup would otherwise be one more dependent
// global round trip per tree level)
struct Tabs { const double *expl, *sqrt_, *c; };
template <int R>
__device__ __forceinline__ void stage_tabs(const View &v, double *s_tab, Tabs &t) {
    const int n = v.tab_lds;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if constexpr (R != 1) s_tab[i] = v.expl_tab[i];
        else { s_tab[i] = v.sqrt_tab[i]; s_tab[n + i] = v.c_tab[i]; }
    }
    t.expl = s_tab; t.sqrt_ = s_tab; t.c = s_tab + n;
}
__device__ __forceinline__ double tab_at(const double *lds, const double *glob, int n_lds, int i) { return i < n_lds ? lds[i] : glob[i]; }

template <class G>
__device__ __forceinline__ void leaf_boards(const View &v, int g, typename G::B &m, typename G::B &t) {
    const u64 *src = reinterpret_cast<const u64 *>(slot_ptr(v, g) + TreeGeo<G>::LEAF_CHUNK * 16);
#pragma unroll
    for (int i = 0; i < G::W; ++i) { m.w[i] = src[i]; t.w[i] = src[G::W + i]; }
}


// ------------------------------------------------------------------------------------------------
// evaluation cache (layout and protocol: azsp_internal.h)
// ------------------------------------------------------------------------------------------------
// Finds or claims the entry of position (m, t) for game g of the launch's slot group.  Returns the entry index and
// whether the caller must put it on the leaf list (it claimed a table entry, or fell back to its private one).
template <class G>
__device__ __forceinline__ int ent_claim(const View &v, int g, const typename G::B &m, const typename G::B &t, bool &owner) {
    u64 pk = 0;
    int id = g;                                                // the slot's private entry
    owner = true;
    if (G::pos_key(m, t, pk)) {
        const u64 word = pk | ((u64)v.grp << 49) | ((u64)v.ent_epoch << 52);
        uint32_t h = (uint32_t)((pk * 0x9E3779B97F4A7C15ull) >> v.ent_shift);
        for (int probe = 0; probe < EVAL_PROBES; ++probe, h = (h + 1) & v.ent_mask) {
            unsigned long long *kp = reinterpret_cast<unsigned long long *>(&v.ent[(size_t)v.n_slots + h].key);
            const u64 old = atomicCAS(kp, 0ull, (unsigned long long)word);
            if (old == 0) { id = v.n_slots + (int)h; break; }                   // claimed
            if ((old & EVAL_POS_MASK) == pk) {
                // the same position: usable if this group or a LOWER one claimed it (own stream order; the ring of tree steps
                // in search_enqueue puts the lower group's inference launch before our next read) or an earlier search did
                // (all streams joined since); a HIGHER group's claim of this search is not: evaluate privately
                if ((int)((old >> 49) & 7) <= v.grp || (int)(old >> 52) != v.ent_epoch) { id = v.n_slots + (int)h; owner = false; }
                break;
            }
        }
    }
    if (owner) {
        EvalEnt *e = &v.ent[id];
        e->mine = m.w[0];
        e->theirs = t.w[0];
    }
    return id;
}

// ------------------------------------------------------------------------------------------------
// expansion + backup of an evaluated leaf (Node.expand :80-90, backpropagate :214-223, mask / renormalise :119-121)
// ------------------------------------------------------------------------------------------------
// One game per LPG lanes; `st` is the game's slot record (registers, updated here, stored by the caller).
// p32_in / vleaf: the evaluator's output for the leaf (this lane's actions; the value for the side to move)
template <class G, int R>
__device__ __forceinline__ void expand_game(const View &v, int g, int lane, int grp, bool live, SlotRegs<G> &st,
                                            const float (&p32_in)[G::APL], float vleaf, double (*s_pa)[G::AP]) {
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL;
    using T = TreeGeo<G>;
    using WT = typename std::conditional<R == 0, float, double>::type;

    // action_probs * valid_moves -> float64 ; /= np.sum(action_probs)
    double pa[APL];
    float p32[APL];
    bool ok[APL];
#pragma unroll
    for (int k = 0; k < APL; ++k) {
        const int a = lane + k * LPG;
        pa[k] = 0.0;
        p32[k] = 0.0f;
        ok[k] = live && a < A && G::legal(st.lm, st.lt, a);
        if (live && a < A) {
            p32[k] = p32_in[k];
            pa[k] = (double)p32[k] * (double)(ok[k] ? 1 : 0);
        }
        if (a < G::AP) s_pa[grp][a] = pa[k];
    }
    wave_lds_sync();
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
    if (!live) return;
    const int node = st.leaf_node;
    const int e = st.n_blk;
    const int L = st.path_len;
    const bool root_exp = L == (R == 1 ? 1 : 0);           // the leaf is the tree's root
    const bool noisy = root_exp && v.root_noise != nullptr;
    // the new block: one 16-byte store per child (Node.expand :80-90) + the header
    int4 *blk = reinterpret_cast<int4 *>(block_ptr(v, g, e));
#pragma unroll
    for (int k = 0; k < APL; ++k) {
        const int a = lane + k * LPG;
        if (a < A) {
            // the float32 numerator of the prior: p32 * mask (a masked-out NaN / inf stays what the reference's product gives)
            const float pm = R != 1 ? (ok[k] ? p32[k] : p32[k] * 0.0f) : (ok[k] ? p32[k] : -1.0f);
            blk[a] = make_int4(0, __float_as_int(0.0f), 0, __float_as_int(pm));
            if constexpr (R != 0) v.wsum_d[(size_t)g * v.C + 1 + e * A + a] = 0.0;
            if (noisy) {
                // optional root exploration noise (off by default; not in the reference's search, listed in its
                // src/deepmind_alphazero_pseudocode.py:285-290): prior*(1-eps) + noise*eps on the existing root children
                const double pr = R != 1 ? pa[k] / s : (ok[k] ? pa[k] : -1.0);
                v.root_prior[(size_t)g * A + a] = child_exists<R>(pr) ? pr * v.noise_keep + v.root_noise[(size_t)g * A + a] * v.noise_eps : pr;
            }
        } else if (a == A) {
            const double ps = s;
            blk[A] = make_int4(__double2loint(ps), __double2hiint(ps), noisy ? 1 : 0, node);
        }
    }
    // the value is the evaluator's float32 for the side to move at the leaf (rule set 1: value[state.player],
    // a float32 widened by .tolist(), search.py:91); signs alternate upwards in both rule sets
    const WT val = (WT)vleaf;
#pragma unroll
    for (int k = 0; k < T::CPL; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // backpropagate :214-223 / search.py:48-58
            int i;
            if (!path_entry_here<G>(lane, k, j, i) || i >= L) continue;
            const int nd = comp4(st.sc[k], j);
            const int dist = L - 1 - i;
            const WT dv = (dist & 1) ? -val : val;
            if constexpr (R != 0) { double *w = v.wsum_d + (size_t)g * v.C + nd; *w = *w + dv; }
            if (nd != 0) {                              // (node 0, a fresh root, keeps its statistics in the slot record: below)
                NodeChunk *c = node_chunk<G>(v, g, nd);
                if constexpr (R == 0) {
                    int2 nw = *reinterpret_cast<int2 *>(c);
                    nw.x += 1;
                    nw.y = __float_as_int(__int_as_float(nw.y) + (float)dv);
                    *reinterpret_cast<int2 *>(c) = nw;
                } else {
                    c->n = c->n + 1;
                }
            }
        }
    if (L > 0 && st.root == 0) {                    // the path starts at the root: node 0 can only be entry 0, distance L - 1
        st.root_n += 1;
        if constexpr (R == 0) st.root_w = st.root_w + (float)(((L - 1) & 1) ? -val : val);
    }
    if (node == 0) st.root_kid = e + 1;
    else if (lane == 0) node_chunk<G>(v, g, node)->kid = e + 1;
    st.n_blk = e + 1;
    st.pending = 0;
}

// ------------------------------------------------------------------------------------------------
// selection: PUCT descent to the first unexpanded node, leaf step, terminal test
// ------------------------------------------------------------------------------------------------
// APPEND: the non-terminal leaves are appended to the group's leaf list here (one atomic per wavefront; the order of
// the list does not matter to the device evaluators, whose outputs do not depend on a position's place in the batch)
// instead of by the stable k_compact pass that the host-callback evaluator needs.
// leaf_id / leaf_add (meaningful on lane 0 of a game whose descent ended on a non-terminal leaf): what goes on the leaf
// list -- the slot itself, or with the evaluation cache the entry to evaluate, and only if this game claimed it
// cur_m / cur_t: the game's current position (the root's), loaded by the caller -- together with the slot record, so
// that the descent does not start with a round trip of its own
template <class G, int R>
__device__ __forceinline__ int select_game(const View &v, int g, int lane, int grp, bool live, SlotRegs<G> &st, int *s_path,
                                            const Tabs &tb, const typename G::B &cur_m, const typename G::B &cur_t,
                                            int &leaf_id, bool &leaf_add) {
    leaf_id = g;
    leaf_add = true;
    using WT = typename std::conditional<R == 0, float, double>::type;
    using T = TreeGeo<G>;
    constexpr int A = G::A, LPG = G::LPG, APL = G::APL;
    constexpr int HL = A % LPG, HK = A / LPG;          // lane and load slot that fetch a block's header chunk
    int *my_path = s_path + grp * T::D;

    // One dependent round trip per tree level: the lanes fetch the whole record of the node's children with one 16-byte
    // load each; the chosen child's block index then comes from a shuffle, and below the root N_parent is not loaded
    // but rebuilt from the children (an expanded node is visited once at expansion and afterwards only on the way to
    // one of its children: N = 1 + sum N_child).  The position is replayed move by move from the slot's current one.
    const int root = live ? st.root : 0;
    int node = root, depth = 0, action = 0;
    int kid = 0, n_root = 0;                           // 1 + block of the node's children (0 = unexpanded)
    typename G::B bm = board_zero<G::W>(), bt = board_zero<G::W>();
    if (live) {
        bm = cur_m;
        bt = cur_t;
        if (root == 0) { kid = st.root_kid; n_root = st.root_n; }
        else { const NodeChunk c = *node_chunk<G>(v, g, root); kid = c.kid; n_root = c.n; }
    }
    bool walking = live;
    while (true) {
        walking = walking && kid != 0;              // Node.expanded() :35-36
        if (!__any(walking)) break;
        const int e = kid - 1;
        const int4 *blk = reinterpret_cast<const int4 *>(block_ptr(v, g, walking ? e : 0));
        int4 ch[APL];
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            const int c = lane + k * LPG;
            ch[k] = (walking && c <= A) ? blk[c] : make_int4(0, 0, 0, 0);
        }
        WT w[APL];
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            const int a = lane + k * LPG;
            if constexpr (R == 0) w[k] = __int_as_float(ch[k].y);
            else w[k] = (walking && a < A) ? v.wsum_d[(size_t)g * v.C + 1 + e * A + a] : 0.0;
        }
        const double psum = make_double(__shfl(ch[HK].x, HL, LPG), __shfl(ch[HK].y, HL, LPG));
        const int flags = __shfl(ch[HK].z, HL, LPG);
        double p[APL];
        int n[APL];
        int nsum = 0;
#pragma unroll
        for (int k = 0; k < APL; ++k) {
            const int a = lane + k * LPG;
            const bool in = walking && a < A;
            n[k] = in ? ch[k].x : 0;
            const float pf = __int_as_float(ch[k].w);
            if (!in) p[k] = R != 1 ? 0.0 : -1.0;                           // lanes without an action: no child
            else if (flags & 1) p[k] = v.root_prior[(size_t)g * A + a];     // root priors with noise mixed in
            else if constexpr (R == 1) p[k] = (double)pf;
            else p[k] = (double)pf / psum;                                 // f64(p32 * mask) / np.sum, :119-121
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
                    u = (p[k] * tab_at(tb.expl, v.expl_tab, v.tab_lds, np_)) / (double)(n[k] + 1);
                    if (n[k] > 0) {
                        const float qf = w[k] / (float)n[k];
                        q = (double)(-qf);
                    }
                } else if constexpr (R == 2) {
                    // the same expression as numpy 1.x evaluates it: W is float64 there (legacy scalar promotion
                    // of `0.0 + np.float32`), so -(W / n) is a float64 division
                    u = (p[k] * tab_at(tb.expl, v.expl_tab, v.tab_lds, np_)) / (double)(n[k] + 1);
                    if (n[k] > 0) q = -(w[k] / (double)n[k]);
                } else {
                    // search.py:33-40: child.value + c * prior * sqrt(N) / (1 + n), left to right, all float64
                    u = ((tab_at(tb.c, v.c_tab, v.tab_lds, np_) * p[k]) * tab_at(tb.sqrt_, v.sqrt_tab, v.tab_lds, np_)) / (double)(1 + n[k]);
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
        int kc_sel = ch[0].z;
#pragma unroll
        for (int k = 1; k < APL; ++k) if (best / LPG == k) kc_sel = ch[k].z;
        const int next_kid = __shfl(kc_sel, best & (LPG - 1), LPG);
        if (walking) {
            if (lane == 0) my_path[depth] = node;
            // the position below: parent's position + action, seen by the opponent (:135-145)
            typename G::B nm, nt;
            G::play(bm, bt, best, nm, nt);
            bm = nm;
            bt = nt;
            action = best;
            node = 1 + e * A + best;
            kid = next_kid;
            ++depth;
        }
    }
    (void)action;
    if (live && lane == 0) my_path[depth] = node;
    const int L = depth + 1;                        // nodes on the path, root .. leaf
    wave_lds_sync();                                // orders the LDS path writes of lane 0 before the other lanes' reads

    // leaf step: the replayed position IS get_next_state + get_canonical_board of the parent's; terminal test :145
    int term = TERM_NONE;
    if (live) term = depth > 0 ? G::terminal(bm, bt) : TERM_NONE;      // (depth 0: an unexpanded root, handled by the caller)
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
            if constexpr (R != 0) { double *w = v.wsum_d + (size_t)g * v.C + nd; *w = *w + dv; }
            if (nd != 0) {                              // (a fresh root's statistics: slot record, lane 0 below)
                NodeChunk *c = node_chunk<G>(v, g, nd);
                if constexpr (R == 0) {
                    int2 nw = *reinterpret_cast<int2 *>(c);
                    nw.x += 1;
                    nw.y = __float_as_int(__int_as_float(nw.y) + (float)dv);
                    *reinterpret_cast<int2 *>(c) = nw;
                } else {
                    c->n = c->n + 1;
                }
            }
        }
        st.pending = 0;
        if (root == 0) {                                // node 0 is path entry 0: distance L - 1 from the leaf
            const int k0 = L - 1;
            const bool neg0 = R != 1 ? (k0 & 1) : (k0 > 0 && ((k0 - 1) & 1));
            st.root_n += 1;
            if constexpr (R == 0) st.root_w = st.root_w + (float)(neg0 ? -val : val);
        }
        if (lane == 0) {
            SlotHead *h = reinterpret_cast<SlotHead *>(slot_ptr(v, g));
            h->pending = 0;
            h->root_n = st.root_n;
            h->root_w = st.root_w;
        }
    } else if (live) {
        int32_t *path = reinterpret_cast<int32_t *>(slot_ptr(v, g) + T::PATH_CHUNK * 16);
        for (int k = lane; k < L; k += LPG) path[k] = my_path[k];
        st.pending = 1;
        st.path_len = L;
        st.leaf_node = node;
        st.lm = bm;
        st.lt = bt;
        if (lane == 0) {
            SlotHead *h = reinterpret_cast<SlotHead *>(slot_ptr(v, g));
            h->pending = 1;
            h->path_len = L;
            h->leaf_node = node;
            if constexpr (G::W == 1) {
                if (v.ent != nullptr) {
                    leaf_id = ent_claim<G>(v, g, bm, bt, leaf_add);
                    h->eval_ent = leaf_id;
                }
            }
            slot_store_leaf<G>(v, g, bm, bt);
        }
        if constexpr (G::W == 1) st.eval_ent = __shfl(leaf_id, 0, LPG);
    }
    // statistics: ONE atomic per wavefront and counter (one per game made 4096 same-address atomics per launch)
    {
        const bool l0 = live && lane == 0;
        const unsigned long long terms = __ballot(l0 && term != TERM_NONE);
        int lsum = l0 ? L : 0;
#pragma unroll
        for (int m = LPG; m < 64; m <<= 1) lsum += __shfl_xor(lsum, m, 64);
        if ((threadIdx.x & 63) == 0) {
            if (lsum) atomicAdd(&v.counters[2], (unsigned long long)lsum);
            if (terms) atomicAdd(&v.counters[1], (unsigned long long)__popcll(terms));
        }
    }
    return term;
}

// The non-terminal leaves found by a wave join a leaf list: one atomic per wavefront (the order of the list does not
// matter to the device evaluators, whose outputs do not depend on a position's place in the batch, and results come
// back by slot).  `list` is indexed from the group's first slot `lo`; `count` is the list's length counter.
// `need`: this lane's game found a position that needs the evaluator (counted as one of the reference's evaluations);
// `add` (implies need): its list entry `id` -- the slot, or with the evaluation cache the entry this game claimed --
// joins the list (counted as an evaluation executed).
__device__ __forceinline__ void append_leaves(const View &v, bool need, bool add, int id, int32_t *list, int32_t *count) {
    const unsigned long long bal_need = __ballot(need);
    const unsigned long long bal = __ballot(add);
    const int cnt = __popcll(bal);
    const int wl = threadIdx.x & 63;
    if (bal_need != 0 && wl == __ffsll((long long)bal_need) - 1) atomicAdd(&v.counters[0], (unsigned long long)__popcll(bal_need));
    int base_j = 0;
    if (cnt > 0) {
        const int leader = __ffsll((long long)bal) - 1;
        if (wl == leader) {
            base_j = atomicAdd(count, cnt);
            atomicAdd(&v.counters[4], (unsigned long long)cnt);
        }
        base_j = __shfl(base_j, leader, 64);
        if (add) list[v.lo + base_j + __popcll(bal & ((1ull << wl) - 1ull))] = id;
    }
}

// FUSED rounds: the workgroup that evaluated the leaf of game g carries the game on -- expansion + backup with the
// (p, v) it just computed, then descents until the next leaf that needs the network (a terminal leaf is backed up on
// the spot and the game descends again: its simulations are sequential and independent of every other game, so doing
// several of them in one launch changes no result) or until the ply's n_sims simulations are done.  The leaf joins the
// NEXT launch's list.  All lanes of the wave must call this (lane groups without a game pass live = false).
template <class G, int R>
__device__ __forceinline__ void advance_game(const View &v, int g, int lane, int grp, bool live, const float (&p32)[G::APL], float vleaf,
                                             double (*s_pa)[G::AP], int *s_path, const Tabs &tb, int32_t *next_list, int32_t *next_count) {
    SlotRegs<G> st;
    slot_load<G>(v, g, lane, live, st);
    const bool counted = live && st.path_len > (R == 1 ? 1 : 0);      // a root expansion is not a simulation (rule set 0)
    expand_game<G, R>(v, g, lane, grp, live, st, p32, vleaf, s_pa);
    if (counted) st.sims_done += 1;
    if (live && lane == 0) slot_store_head<G>(v, g, st);
    bool again = live && st.sims_done < v.n_sims;
    bool found = false;
    typename G::B cur_m = board_zero<G::W>(), cur_t = board_zero<G::W>();
    if (live) { cur_m = board_load<G::W>(v.cur_mine, g); cur_t = board_load<G::W>(v.cur_theirs, g); }
    while (__any(again)) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // this wave's stores before its own next loads
        int leaf_id;
        bool leaf_add;
        const int term = select_game<G, R>(v, g, lane, grp, again, st, s_path, tb, cur_m, cur_t, leaf_id, leaf_add);     // (FUSED rounds run without the evaluation cache)
        if (again) {
            if (term != TERM_NONE) {
                st.sims_done += 1;
                if (lane == 0) reinterpret_cast<SlotHead *>(slot_ptr(v, g))->sims_done = st.sims_done;
                again = st.sims_done < v.n_sims;
            } else {
                found = true;
                again = false;
            }
        }
        wave_lds_sync();                                             // the path staging area is reused by the next descent
    }
    append_leaves(v, found && lane == 0, found && lane == 0, g, next_list, next_count);
}
}  // namespace azsp

#pragma clang fp contract(fast)
