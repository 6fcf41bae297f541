// Host side of libazsp.so: engine object, HIP resource management, round/ply sequencing and the
// C ABI declared in include/azsp.h.  No torch, no Python: plain pointers and sizes only.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <chrono>
#include <vector>

#include "../../include/azsp.h"
#include "azsp_internal.h"

using namespace azsp;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(AZSP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// Every entry point works on the engine's own device and leaves the calling thread's current device as it found it
// (a training loop next to the generator keeps its torch.cuda.current_device()).
struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == dev || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1;                // nothing to restore
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ON_DEVICE(dev)                                                                    \
    DeviceScope _dev_scope(dev);                                                          \
    if (!_dev_scope.ok) return fail(AZSP_ERR_HIP, "hipSetDevice(%d) failed", (int)(dev))

struct GameInfo { int rows, cols, A, cells, max_plies, W, AP; };
static bool game_info(int game, GameInfo &gi) {
    switch (game) {
        case AZSP_GAME_CONNECT4: gi = {Connect4::ROWS, Connect4::COLS, Connect4::A, Connect4::CELLS, Connect4::MAX_PLIES, Connect4::W, Connect4::AP}; return true;
        case AZSP_GAME_CONNECT2: gi = {Connect2::ROWS, Connect2::COLS, Connect2::A, Connect2::CELLS, Connect2::MAX_PLIES, Connect2::W, Connect2::AP}; return true;
        case AZSP_GAME_GOMOKU9: gi = {Gomoku9::ROWS, Gomoku9::COLS, Gomoku9::A, Gomoku9::CELLS, Gomoku9::MAX_PLIES, Gomoku9::W, Gomoku9::AP}; return true;
    }
    return false;
}

template <class G>
static void host_cells_to_words(const int8_t *cells, u64 *m, u64 *t) {
    typename G::B bm, bt;
    cells_to_board<G>(cells, bm, bt);
    for (int i = 0; i < G::W; ++i) { m[i] = bm.w[i]; t[i] = bt.w[i]; }
}
static void host_cells_to_bits(int game, const int8_t *cells, u64 *m, u64 *t) {
    if (game == AZSP_GAME_CONNECT4) host_cells_to_words<Connect4>(cells, m, t);
    else if (game == AZSP_GAME_CONNECT2) host_cells_to_words<Connect2>(cells, m, t);
    else host_cells_to_words<Gomoku9>(cells, m, t);
}

enum { EV_INFER = 0, EV_SELECT = 1, EV_EXPAND = 2, EV_PLAY = 3, EV_KINDS = 4 };

struct EventPair { hipEvent_t a, b; int kind; };

struct azsp_engine {
    azsp_config cfg{};
    GameInfo gi{};
    hipStream_t stream = nullptr;                 // main stream (group 0 runs on it too)
    hipStream_t gstream[MAX_GROUPS] = {};         // gstream[0] == stream
    hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS] = {};
    hipEvent_t ev_tree[MAX_GROUPS] = {};          // evaluation cache with several slot groups: the ring of tree steps (search_enqueue)
    int n_groups = 1;
    View v{};
    std::vector<void *> allocs;
    int evaluator = AZSP_EVAL_HASH;
    azsp_predict_fn cb = nullptr;
    void *cb_user = nullptr;
    ResNetDev *net = nullptr;
    bool net_loaded = false;
    bool searched = false;
    bool play_open = false;       // between azsp_play_begin and azsp_play_commit
    // pinned host staging
    PlyResultDev *h_res = nullptr;
    int32_t *h_nleaf = nullptr;
    unsigned long long *h_counters = nullptr;
    double *d_uniforms = nullptr;
    int8_t *d_cells = nullptr;      // [n_slots][cells] device staging for the host evaluator
    u64 *d_tmp_mine = nullptr, *d_tmp_theirs = nullptr;
    int32_t *d_st_counts = nullptr, *d_st_rn = nullptr;     // root statistics staging
    float *d_st_w = nullptr, *d_st_rw = nullptr;
    double *d_st_prior = nullptr;
    double *d_noise = nullptr;        // root exploration noise staging
    float *d_stage_p = nullptr, *d_stage_v = nullptr;   // host evaluator: results in list order before the scatter to slots
    size_t leaf_off = 0;              // byte offset of the leaf position inside a slot record
    int32_t eval_off = -1;            // byte offset of the evaluator's row inside a slot record, -1 = separate arrays
    // diagnostics switches (AZSP_DIAG, read once at engine creation; include/azsp.h): the product defaults are both true
    bool fused_rounds_small = true;   // up to 32 games: one launch per simulation round (one-board kernels with the tree step fused in)
    bool fused_tree = true;           // device evaluators: expansion of round r-1 and descent of round r as ONE launch
    // evaluation cache (AZSP_FLAG_EVAL_CACHE): n_slots private entries + ent_table (power of two) hashed ones
    EvalEnt *d_ent = nullptr;
    size_t ent_table = 0;
    int ent_shift = 0, ent_epoch = 0;
    bool timed_now = true;            // AZSP_FLAG_TIMING with timing_sample > 1: whether the round being enqueued is in a timed window
    int last_infer_cap = 0;           // board capacity of the most recent inference launch (selects the kernel variant)
    double *d_st_wd = nullptr, *d_st_rwd = nullptr;   // float64 root statistics staging
    double *d_sqrt = nullptr, *d_ctab = nullptr, *d_expl = nullptr;   // [n_sims + 2] exploration tables
    double c_init = 1.0, c_base = 0.0;
    int tab_len = 0;
    std::vector<int8_t> h_cells;
    std::vector<float> h_p, h_v, h_p8;
    std::vector<u64> h_mine, h_theirs;
    // timing
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    azsp_stats stats{};
    uint64_t plies = 0;
    uint64_t rounds_host = 0;
    double infer_union_ms = 0;
};

template <class T>
static int dev_alloc(azsp_engine *e, T *&p, size_t n) {
    void *q = nullptr;
    hipError_t err = hipMalloc(&q, n * sizeof(T) + 64);
    if (err != hipSuccess) return fail(AZSP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(err));
    e->allocs.push_back(q);
    p = (T *)q;
    return 0;
}
#define DEV_ALLOC(ptr, n) do { int _rc = dev_alloc(e, ptr, (size_t)(n)); if (_rc) return _rc; } while (0)

// ---- timing -------------------------------------------------------------------------------------
static EventPair *ev_begin(azsp_engine *e, int kind, hipStream_t st) {
    if (!(e->cfg.flags & AZSP_FLAG_TIMING)) return nullptr;
    if (e->ev_used == e->ev_pool.size()) {
        EventPair p{};
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        e->ev_pool.push_back(p);
    }
    EventPair *p = &e->ev_pool[e->ev_used++];
    p->kind = kind;
    (void)hipEventRecord(p->a, st);
    return p;
}
static void ev_end(EventPair *p, hipStream_t st) { if (p) (void)hipEventRecord(p->b, st); }
// a pair for a launch that records its own start / end (hipExtLaunchKernelGGL): nothing is enqueued here
static EventPair *ev_pair(azsp_engine *e, int kind) {
    if (!(e->cfg.flags & AZSP_FLAG_TIMING) || !e->timed_now) return nullptr;
    if (kind == EV_INFER) e->stats.infer_launches_timed++;
    if (e->ev_used == e->ev_pool.size()) {
        EventPair p{};
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        e->ev_pool.push_back(p);
    }
    EventPair *p = &e->ev_pool[e->ev_used++];
    p->kind = kind;
    return p;
}

static void ev_collect(azsp_engine *e) {       // call only after the streams have been synchronised
    // Inference launches of different slot groups overlap in time; besides the per-launch sum, keep
    // the length of the union of their [start, end] intervals (same device => one time base).
    std::vector<std::pair<float, float>> iv;
    hipEvent_t base = nullptr;
    for (size_t i = 0; i < e->ev_used; ++i) {
        const EventPair &p = e->ev_pool[i];
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
        switch (p.kind) {
            case EV_INFER: {
                e->stats.infer_ms += ms;
                if (!base) base = p.a;
                float t0 = 0;
                if (base != p.a && hipEventElapsedTime(&t0, base, p.a) != hipSuccess) break;
                iv.emplace_back(t0, t0 + ms);
                break;
            }
            case EV_SELECT: e->stats.select_ms += ms; break;
            case EV_EXPAND: e->stats.expand_ms += ms; break;
            case EV_PLAY: e->stats.play_ms += ms; break;
        }
    }
    if (!iv.empty()) {
        std::sort(iv.begin(), iv.end());
        float lo = iv[0].first, hi = iv[0].second;
        double tot = 0;
        for (size_t i = 1; i < iv.size(); ++i) {
            if (iv[i].first > hi) { tot += hi - lo; lo = iv[i].first; hi = iv[i].second; }
            else if (iv[i].second > hi) hi = iv[i].second;
        }
        e->infer_union_ms += tot + (hi - lo);
    }
    e->ev_used = 0;
}

static int sync_stream(azsp_engine *e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    ev_collect(e);
    return 0;
}

// ---- evaluation of the compacted leaves of one slot group ------------------------------------------
static int evaluate_leaves(azsp_engine *e, const View &v, hipStream_t st) {
    const int game = e->cfg.game;
    switch (e->evaluator) {
        case AZSP_EVAL_HASH:
            launch_eval_hash(game, v, e->cfg.hash_salt, st);
            return 0;
        case AZSP_EVAL_RESNET: {
            if (!e->net || !e->net_loaded) return fail(AZSP_ERR_STATE, "ResNet evaluator selected but no weights loaded (azsp_load_resnet)");
            EventPair *p = ev_pair(e, EV_INFER);
            hipEvent_t ea = p ? p->a : nullptr, eb = p ? p->b : nullptr;
            e->last_infer_cap = v.n_act;
            if (v.ent != nullptr) {
                // evaluation cache: the list holds entries; positions are read from and rows written into them
                float *row0 = v.ent[0].row;
                resnet_forward(e->net, game, &v.ent[0].mine, (int)(sizeof(EvalEnt) / 8), v.leaf_slot + v.lo, v.nl, v.n_act, row0,
                               (int)(sizeof(EvalEnt) / 4), row0 + 7, (int)(sizeof(EvalEnt) / 4), st, /*by_count=*/true, ea, eb);
            } else {
                // leaf positions sit in the slot records; outputs are indexed by slot, so groups never share rows
                resnet_forward(e->net, game, reinterpret_cast<const u64 *>(v.slot + e->leaf_off), v.slot_bytes / 8, v.leaf_slot + v.lo, v.nl,
                               v.n_act, v.eval_p, v.eval_pstride, v.eval_v, v.eval_vstride, st, false, ea, eb);
            }
            e->stats.infer_launches++;
            return 0;
        }
        case AZSP_EVAL_HOST: {
            if (!e->cb) return fail(AZSP_ERR_STATE, "host evaluator selected but no callback set");
            if (v.lo != 0) return fail(AZSP_ERR_STATE, "host evaluator needs a single slot group");
            const int A = e->gi.A, cells = e->gi.cells, AP = e->gi.AP;
            launch_leaf_cells(game, v, e->d_cells, st);
            HIP_TRY(hipMemcpyAsync(e->h_nleaf, v.nl, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const int m = *e->h_nleaf;
            if (m <= 0) return 0;
            HIP_TRY(hipMemcpyAsync(e->h_cells.data(), e->d_cells, (size_t)m * cells, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (e->cb(e->cb_user, e->h_cells.data(), m, e->h_p.data(), e->h_v.data()))
                return fail(AZSP_ERR_CALLBACK, "host evaluator callback reported failure");
            for (int j = 0; j < m; ++j)
                for (int a = 0; a < AP; ++a) e->h_p8[(size_t)j * AP + a] = a < A ? e->h_p[(size_t)j * A + a] : 0.0f;
            HIP_TRY(hipMemcpyAsync(e->d_stage_p, e->h_p8.data(), (size_t)m * AP * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(e->d_stage_v, e->h_v.data(), (size_t)m * sizeof(float), hipMemcpyHostToDevice, st));
            launch_scatter_eval(game, v, e->d_stage_p, e->d_stage_v, st);      // list order -> slot order
            HIP_TRY(hipStreamSynchronize(st));
            return 0;
        }
    }
    return fail(AZSP_ERR_ARG, "unknown evaluator kind %d", e->evaluator);
}

// MCTS.run_batch on the first n_act slots (monte_carlo_tree_search.py:111-171).
// Games are independent, so the slots are cut into contiguous groups and each group's dependent
// chain select -> compact -> evaluate -> expand is enqueued on its own HIP stream: the tail of one
// group's inference launch overlaps the next group's, and the latency-bound tree kernels of one
// group hide behind the other's MFMA work.  Results do not depend on the grouping.
static int search_enqueue(azsp_engine *e, int n_act) {
    const int game = e->cfg.game;
    int G = e->n_groups;
    if (e->evaluator == AZSP_EVAL_HOST || n_act < 64 * G) G = 1;
    // Device evaluators: expansion of round s-1 and descent of round s run as ONE launch that appends its leaves with a
    // wave-level atomic (two launches per round).  The host-callback evaluator keeps the stable, slot-ordered
    // select -> compact -> evaluate -> expand sequence (the reference hands its leaves over in slot order).
    const bool fused = e->fused_tree && e->evaluator != AZSP_EVAL_HOST;
    // FUSED rounds (a few games, one-board kernels of the exact-fp32 Connect-4 network, rule set 0): ONE launch per round
    // and group -- the inference kernel expands the leaf it evaluated and finds the game's next leaf itself (a game whose
    // descent ends on a terminal leaf backs it up and descends again in the same launch).  Round r reads list r & 1 /
    // counter r % 3, fills list (r + 1) & 1 / counter (r + 1) % 3 and clears counter (r + 2) % 3.  Bit-exact (tested).
    // Only up to 32 games, where a round is launch latency (one game 106.6 -> 101.0 us per round); for large batches the
    // same idea measured SLOWER (2101 against 2209 episodes/s: the tree step is a chain of dependent loads during which
    // the workgroup's other waves idle) and its kernels were retired after round 2.
    // (Not with the evaluation cache: a round whose leaves are all known then costs a tree launch and an empty inference
    // launch, ~20 us instead of ~100 -- one game, 100 simulations: 24 % of the rounds.)
    bool fused_rounds = e->fused_rounds_small && e->evaluator == AZSP_EVAL_RESNET && e->net && e->net_loaded && e->d_ent == nullptr &&
                        e->cfg.search_rules == AZSP_RULES_SIMPLE && !e->v.reuse && e->v.root_noise == nullptr;
    View gv[MAX_GROUPS];
    for (int k = 0; k < G; ++k) {
        gv[k] = e->v;
        gv[k].grp = k;
        gv[k].nl = e->v.n_leaf + 4 * k;
        gv[k].nl_other = gv[k].nl + 1;
        gv[k].lo = (int)((long long)n_act * k / G);
        gv[k].n_act = (int)((long long)n_act * (k + 1) / G) - gv[k].lo;
    }
    for (int k = 0; k < G && fused_rounds; ++k) fused_rounds = resnet_has_fused(e->net, game, gv[k].n_act, e->v.tab_lds);
    // Evaluation cache (AZSP_FLAG_EVAL_CACHE; protocol in azsp_internal.h): needs a deterministic device evaluator and
    // the fused tree launches; every search gets the next epoch number (the table is cleared when they run out).
    const bool cache = e->d_ent != nullptr && fused && !fused_rounds;
    if (cache) {
        if (++e->ent_epoch > 4095) {
            HIP_TRY(hipMemsetAsync(e->d_ent, 0, (e->ent_table + (size_t)e->cfg.n_slots) * sizeof(EvalEnt), e->stream));
            e->ent_epoch = 1;
        }
        for (int k = 0; k < G; ++k) {
            gv[k].ent = e->d_ent;
            gv[k].ent_mask = (uint32_t)(e->ent_table - 1);
            gv[k].ent_shift = e->ent_shift;
            gv[k].ent_epoch = e->ent_epoch;
        }
    }
    if (G > 1) {
        HIP_TRY(hipEventRecord(e->ev_fork, e->stream));
        for (int k = 1; k < G; ++k) HIP_TRY(hipStreamWaitEvent(e->gstream[k], e->ev_fork, 0));
    }
    // Evaluation cache with several slot groups: group k may use an entry that a LOWER group claimed during this very
    // search (ent_claim), so the tree steps of the groups run as a ring -- step t of group k after step t of group k - 1,
    // step t + 1 of group 0 after step t of the last group.  A lower group j is then never more than one step ahead of k:
    // every claim of j that step t of k can see was made in j's steps <= t, their rows are written by j's inference
    // launches up to the one after its step t, and k reads them in its step t + 1, which starts after j's step t + 1,
    // i.e. after that launch (stream order on j).  Tree launches still overlap the other groups' inference launches.
    const bool ring = cache && G > 1;
    auto ring_wait = [&](int k, bool first_step) -> int {           // before tree step t of group k
        if (!ring) return 0;
        if (k > 0) HIP_TRY(hipStreamWaitEvent(e->gstream[k], e->ev_tree[k - 1], 0));
        else if (!first_step) HIP_TRY(hipStreamWaitEvent(e->gstream[0], e->ev_tree[G - 1], 0));
        return 0;
    };
    auto ring_done = [&](int k) -> int { if (ring) HIP_TRY(hipEventRecord(e->ev_tree[k], e->gstream[k])); return 0; };
    // AZSP_FLAG_TIMING with timing_sample = n > 1: events only in windows of 32 consecutive rounds out of every 32 n (the
    // root stage counts as round 0): attaching timestamps to a dispatch costs a few microseconds of its own, which shows
    // where a round is launch latency (evaluation cache, few games).  (Long windows: where launches of several groups
    // overlap, the union of a window's intervals misses the untimed neighbours at its two ends.)
    // The windows run over ALL rounds of the engine's life, not per search: the first rounds of a ply meet more known positions
    // than its last ones (with the evaluation cache), so a window that always sat at a ply's start would not be representative.
    const int tsamp = e->cfg.timing_sample > 1 ? e->cfg.timing_sample : 1;
    const uint64_t round0 = e->rounds_host;
    auto in_window = [&](int round) { return tsamp == 1 || ((round0 + (uint64_t)round) % (uint64_t)(32 * tsamp)) < 32; };
    e->timed_now = in_window(0);
    for (int k = 0; k < G; ++k) {
        hipStream_t st = e->gstream[k];
        if (cache) {
            // the roots join the list from k_root_prepare (entries, claimed once per position): zero its counters first
            HIP_TRY(hipMemsetAsync(gv[k].nl, 0, 4 * sizeof(int32_t), st));
            { int rc = ring_wait(k, true); if (rc) return rc; }
            launch_root_prepare(game, gv[k], st);
            { int rc = ring_done(k); if (rc) return rc; }
        } else {
            launch_root_prepare(game, gv[k], st);
            launch_compact(gv[k], st);
        }
        if (fused_rounds) continue;              // the roots are evaluated by round 0
        int rc = evaluate_leaves(e, gv[k], st);
        if (rc) return rc;
    }
    if (fused_rounds) {
        for (int r = 0; r <= e->cfg.n_sims; ++r)
            for (int k = 0; k < G; ++k) {
                e->timed_now = in_window(r);
                hipStream_t st = e->gstream[k];
                View w = gv[k];
                int32_t *cnt = e->v.n_leaf + 4 * k;
                int32_t *cur_list = (r & 1) ? e->v.leaf_slot2 : e->v.leaf_slot, *next_list = (r & 1) ? e->v.leaf_slot : e->v.leaf_slot2;
                EventPair *p = ev_pair(e, EV_INFER);
                resnet_forward_fused(e->net, game, w, reinterpret_cast<const u64 *>(w.slot + e->leaf_off), w.slot_bytes / 8,
                                     cur_list + w.lo, cnt + r % 3, next_list, cnt + (r + 1) % 3, cnt + (r + 2) % 3, w.n_act, st,
                                     p ? p->a : nullptr, p ? p->b : nullptr);
                e->stats.infer_launches++;
            }
        for (int k = 1; k < G; ++k) {
            HIP_TRY(hipEventRecord(e->ev_join[k], e->gstream[k]));
            HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join[k], 0));
        }
        e->timed_now = true;
        e->v.lo = 0;
        e->v.grp = 0;
        e->v.n_act = n_act;
        hipError_t le2 = hipGetLastError();
        if (le2 != hipSuccess) return fail(AZSP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le2));
        e->stats.plies++;
        e->rounds_host += (uint64_t)e->cfg.n_sims + 1;
        e->searched = true;
        return 0;
    }
    // rule set 1: the root expansion is simulation 1 (search.py:66-91)
    const int descents = e->cfg.search_rules == AZSP_RULES_GEN1 ? e->cfg.n_sims - 1 : e->cfg.n_sims;
    if (!fused)
        for (int k = 0; k < G; ++k) launch_expand(game, gv[k], e->gstream[k]);
    for (int s = 0; s < descents; ++s) {
        e->timed_now = in_window(s + 1);
        for (int k = 0; k < G; ++k) {
            hipStream_t st = e->gstream[k];
            if (fused) {
                // round s + 1 fills the group's counter of parity (s + 1) & 1; the root stage used parity 0
                View w = gv[k];
                w.nl = e->v.n_leaf + 4 * k + ((s + 1) & 1);
                w.nl_other = e->v.n_leaf + 4 * k + (s & 1);
                { int rc = ring_wait(k, false); if (rc) return rc; }
                EventPair *p = ev_pair(e, EV_SELECT);
                launch_expand_select(game, w, st, p ? p->a : nullptr, p ? p->b : nullptr);
                { int rc = ring_done(k); if (rc) return rc; }
                e->stats.select_launches++;
                int rc = evaluate_leaves(e, w, st);
                if (rc) return rc;
                continue;
            }
            EventPair *p = ev_begin(e, EV_SELECT, st);
            launch_select(game, gv[k], st);
            launch_compact(gv[k], st);
            ev_end(p, st);
            e->stats.select_launches++;
            int rc = evaluate_leaves(e, gv[k], st);
            if (rc) return rc;
            p = ev_begin(e, EV_EXPAND, st);
            launch_expand(game, gv[k], st);
            ev_end(p, st);
            e->stats.expand_launches++;
        }
    }
    if (fused)
        for (int k = 0; k < G; ++k) {           // the last round's leaves (or, with no descent at all, the roots)
            View w = gv[k];
            w.nl = e->v.n_leaf + 4 * k + (descents & 1);
            w.nl_other = e->v.n_leaf + 4 * k + ((descents + 1) & 1);
            { int rc = ring_wait(k, false); if (rc) return rc; }
            EventPair *p = ev_pair(e, EV_EXPAND);
            launch_expand(game, w, e->gstream[k], p ? p->a : nullptr, p ? p->b : nullptr);
            { int rc = ring_done(k); if (rc) return rc; }
            e->stats.expand_launches++;
        }
    for (int k = 1; k < G; ++k) {
        HIP_TRY(hipEventRecord(e->ev_join[k], e->gstream[k]));
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join[k], 0));
    }
    e->timed_now = true;
    e->v.lo = 0;
    e->v.grp = 0;
    e->v.n_act = n_act;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(AZSP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
    e->stats.plies++;
    e->rounds_host += (uint64_t)descents + 1;
    e->searched = true;
    return 0;
}

// (diagnostics, AZSP_PROBE_VERBOSE: host time spent enqueuing searches, printed when the engine is destroyed -- whether
//  the host or the GPU sets the pace of a launch-bound configuration)
static double g_enqueue_s = 0.0;
static uint64_t g_enqueue_rounds = 0;
static int search(azsp_engine *e, int n_act) {
    static const bool verbose = getenv("AZSP_PROBE_VERBOSE") != nullptr;
    if (!verbose) return search_enqueue(e, n_act);
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t r0 = e->rounds_host;
    const int rc = search_enqueue(e, n_act);
    g_enqueue_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_enqueue_rounds += e->rounds_host - r0;
    return rc;
}

// A new generation run, new weights or another evaluator: every cached row is void.  (Cleared on the main stream, which
// every search forks from.)
static int cache_clear(azsp_engine *e) {
    if (!e->d_ent) return 0;
    HIP_TRY(hipMemsetAsync(e->d_ent, 0, (e->ent_table + (size_t)e->cfg.n_slots) * sizeof(EvalEnt), e->stream));
    e->ent_epoch = 0;
    return 0;
}

template <class G>
static void rules_eval(const int8_t *cells, int32_t action, int8_t *next_cells, int32_t *legal_mask, int32_t *terminal) {
    typename G::B m, t, nm, nt;
    cells_to_board<G>(cells, m, t);
    if (legal_mask)                    // int32[(A + 31) / 32] words, bit a = action a is legal
        for (int a = 0; a < G::A; ++a) {
            if ((a & 31) == 0) legal_mask[a >> 5] = 0;
            if (G::legal(m, t, a)) legal_mask[a >> 5] |= (int32_t)(1u << (a & 31));
        }
    if (terminal) *terminal = G::terminal(m, t);
    if (next_cells) {
        if (action < 0 || action >= G::A) { memcpy(next_cells, cells, G::CELLS); return; }
        G::play(m, t, action, nm, nt);
        board_to_cells<G>(nm, nt, next_cells);
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int azsp_abi_version(void) { return AZSP_ABI_VERSION; }

const char *azsp_last_error(void) { return g_err.c_str(); }

int azsp_game_info(int32_t game, int32_t *rows, int32_t *cols, int32_t *n_actions, int32_t *max_plies) {
    GameInfo gi;
    if (!game_info(game, gi)) return fail(AZSP_ERR_ARG, "unknown game id %d", game);
    if (rows) *rows = gi.rows;
    if (cols) *cols = gi.cols;
    if (n_actions) *n_actions = gi.A;
    if (max_plies) *max_plies = gi.max_plies;
    return 0;
}

int azsp_game_rules(int32_t game, const int8_t *cells, int32_t action, int8_t *next_cells, int32_t *legal_mask,
                    int32_t *terminal) {
    if (!cells) return fail(AZSP_ERR_ARG, "null board");
    switch (game) {
        case AZSP_GAME_CONNECT4: rules_eval<Connect4>(cells, action, next_cells, legal_mask, terminal); return 0;
        case AZSP_GAME_CONNECT2: rules_eval<Connect2>(cells, action, next_cells, legal_mask, terminal); return 0;
        case AZSP_GAME_GOMOKU9: rules_eval<Gomoku9>(cells, action, next_cells, legal_mask, terminal); return 0;
    }
    return fail(AZSP_ERR_ARG, "unknown game id %d", game);
}

int32_t azsp_record_bytes(int32_t game) {
    GameInfo gi;
    if (!game_info(game, gi)) return fail(AZSP_ERR_ARG, "unknown game id %d", game);
    return ((gi.cells + 1 + 7) / 8) * 8 + 8 * gi.A;
}

size_t azsp_resnet_blob_floats(int32_t game, int32_t res_blocks, int32_t channels) {
    return resnet_blob_floats(game, res_blocks, channels);
}

void azsp_destroy(azsp_engine *e) {
    if (!e) return;
    if (g_enqueue_rounds && getenv("AZSP_PROBE_VERBOSE"))
        fprintf(stderr, "[azsp host] %.3f s enqueuing %llu simulation rounds: %.1f us of host time per round\n", g_enqueue_s,
                (unsigned long long)g_enqueue_rounds, 1e6 * g_enqueue_s / (double)g_enqueue_rounds);
    DeviceScope _dev_scope(e->cfg.device);
    for (int k = 0; k < MAX_GROUPS; ++k) if (e->gstream[k]) (void)hipStreamSynchronize(e->gstream[k]);
    for (auto &p : e->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (int k = 1; k < MAX_GROUPS; ++k) {
        if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
        if (e->ev_tree[k]) (void)hipEventDestroy(e->ev_tree[k]);
        if (e->gstream[k]) (void)hipStreamDestroy(e->gstream[k]);
    }
    if (e->net) resnet_destroy(e->net);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->h_res) (void)hipHostFree(e->h_res);
    if (e->h_nleaf) (void)hipHostFree(e->h_nleaf);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// sqrt(N), c(N) and c(N)*sqrt(N) for N = 0 .. tab_len-1, computed on the host: libm's sqrt/log are what
// Python's math.sqrt/math.log call, so the tables are bit-identical to the reference's per-call values.
static int upload_exploration_tables(azsp_engine *e) {
    const int n = e->tab_len;
    std::vector<double> sq(n), ct(n), ex(n);
    for (int i = 0; i < n; ++i) {
        sq[i] = sqrt((double)i);
        ct[i] = e->c_base > 0.0 ? log(((double)i + e->c_base + 1.0) / e->c_base) + e->c_init : e->c_init;
        ex[i] = ct[i] * sq[i];                   // c == 1.0: exactly sqrt(N)
    }
    HIP_TRY(hipMemcpy(e->d_sqrt, sq.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_ctab, ct.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_expl, ex.data(), n * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

static int create_impl(azsp_engine *e) {
    const azsp_config &c = e->cfg;
    const GameInfo &gi = e->gi;
    ON_DEVICE(c.device);
    if (const char *x = getenv("AZSP_DIAG")) {       // diagnostics switches (include/azsp.h), read once per engine
        e->fused_rounds_small = strstr(x, "no_fused_rounds") == nullptr;
        e->fused_tree = strstr(x, "unfused_tree") == nullptr;
    }
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->gstream[0] = e->stream;
    e->n_groups = c.n_groups > 0 ? (c.n_groups > MAX_GROUPS ? MAX_GROUPS : c.n_groups) : (c.n_slots >= 3072 ? 3 : (c.n_slots >= 1024 ? 2 : 1));
    // Evaluation cache: a position met by several slot groups in the SAME search is evaluated once per group (a row is only
    // read where stream order guarantees it, azsp_internal.h), and games in lock-step meet the same new positions in the
    // same round -- measured at the headline: 2.94 x fewer evaluations with one group, 2.31 x with two, 2.16 x with three.
    // With a third of the evaluations a round no longer fills the machine twice over, so one group's launch leaves little
    // for a second group to overlap with: 3981 / 3595 / 3387 episodes/s (fp32), 7738 / 8070 / 8055 (f16x2).  With the ring of
    // tree steps (search_enqueue: a group may use what a lower group claimed in the same search) several groups reach
    // 2.87 x / 2.85 x: 4321 / 4386 episodes/s with 1 / 2 groups (fp32), 9964 / 10659 / 10022 with 1 / 2 / 3 (f16x2, whose
    // one- and three-board forms leave more of a round to the tree launch that a second group hides).  Two groups for large
    // runs unless the caller says otherwise.
    if (c.n_groups <= 0 && (c.flags & AZSP_FLAG_EVAL_CACHE)) e->n_groups = c.n_slots >= 1024 ? 2 : 1;
    for (int k = 1; k < e->n_groups; ++k) {
        HIP_TRY(hipStreamCreateWithFlags(&e->gstream[k], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming));
    }
    if (c.flags & AZSP_FLAG_EVAL_CACHE)
        for (int k = 0; k < e->n_groups; ++k) HIP_TRY(hipEventCreateWithFlags(&e->ev_tree[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    View &v = e->v;
    const int N = c.n_slots, A = gi.A, S = c.n_sims, W = gi.W, AP = gi.AP;
    const int EPS = c.arena_episodes > N ? c.arena_episodes : N;
    v.cap_eps = EPS;
    v.root_noise = nullptr;
    v.noise_eps = 0.0;
    v.noise_keep = 1.0;
    v.sampling_plies = -1;
    v.n_slots = v.n_act = N;
    v.lo = 0;
    v.grp = 0;
    v.A = A;
    v.n_sims = S;
    // expansions per tree: one root + one per simulation; with tree reuse a tree lives for a whole game
    v.reuse = (c.flags & AZSP_FLAG_TREE_REUSE) ? 1 : 0;
    v.E = v.reuse ? (S + 1) * gi.max_plies : S + 1;
    v.C = ((1 + A * v.E) + 7) & ~7;
    v.max_plies = gi.max_plies;
    v.cells = gi.cells;
    {
        int32_t blk = 0, slot = 0, dcap = 0, eoff = -1;
        tree_geometry(c.game, &blk, &slot, &dcap, &eoff);
        e->eval_off = eoff;
        v.rec_block = blk;
        v.rec_slot = (int64_t)v.E * blk;
        v.slot_bytes = slot;
        v.D = dcap;
        e->leaf_off = 3 * 16;                      // SlotHead is three 16-byte chunks (azsp_internal.h)
    }
    DEV_ALLOC(v.rec, (size_t)N * (size_t)v.rec_slot);
    DEV_ALLOC(v.slot, (size_t)N * v.slot_bytes);
    HIP_TRY(hipMemset(v.rec, 0, (size_t)N * (size_t)v.rec_slot));
    HIP_TRY(hipMemset(v.slot, 0, (size_t)N * v.slot_bytes));
    DEV_ALLOC(v.root_prior, (size_t)N * A);
    DEV_ALLOC(v.leaf_slot, N); DEV_ALLOC(v.leaf_slot2, N); DEV_ALLOC(v.n_leaf, MAX_GROUPS * 4);
    v.nl = v.n_leaf;
    v.nl_other = v.n_leaf + 1;
    HIP_TRY(hipMemset(v.leaf_slot, 0, (size_t)N * sizeof(int32_t)));
    HIP_TRY(hipMemset(v.leaf_slot2, 0, (size_t)N * sizeof(int32_t)));
    HIP_TRY(hipMemset(v.n_leaf, 0, MAX_GROUPS * 4 * sizeof(int32_t)));
    if (e->eval_off >= 0) {           // narrow action spaces: (p[0..A), value at [7]) inside the slot record
        v.eval_p = reinterpret_cast<float *>(v.slot + e->eval_off);
        v.eval_v = v.eval_p + 7;
        v.eval_pstride = v.eval_vstride = v.slot_bytes / 4;
    } else {
        DEV_ALLOC(v.eval_p, (size_t)N * AP); DEV_ALLOC(v.eval_v, N);
        v.eval_pstride = AP;
        v.eval_vstride = 1;
    }
    DEV_ALLOC(e->d_stage_p, (size_t)N * AP); DEV_ALLOC(e->d_stage_v, N);
    e->tab_len = v.reuse ? (S + 1) * gi.max_plies + 2 : S + 2;    // largest N_parent + 1 a search can see
    DEV_ALLOC(e->d_sqrt, e->tab_len); DEV_ALLOC(e->d_ctab, e->tab_len); DEV_ALLOC(e->d_expl, e->tab_len);
    v.sqrt_tab = e->d_sqrt;
    v.c_tab = e->d_ctab;
    v.expl_tab = e->d_expl;
    v.tab_len = e->tab_len;
    v.tab_lds = e->tab_len < 2048 ? e->tab_len : 2048;     // leading entries staged in LDS by the tree kernels (<= 32 KB)
    v.rules = c.search_rules;
    v.wsum_d = nullptr;
    if (c.search_rules != AZSP_RULES_SIMPLE) {
        const size_t NC = (size_t)N * v.C;
        DEV_ALLOC(v.wsum_d, NC);
        HIP_TRY(hipMemset(v.wsum_d, 0, NC * sizeof(double)));
    }
    for (int i = 0; i < 2; ++i) v.init_mine[i] = v.init_theirs[i] = 0;
    DEV_ALLOC(v.cur_mine, (size_t)N * W); DEV_ALLOC(v.cur_theirs, (size_t)N * W); DEV_ALLOC(v.cur_len, N);
    const size_t NH = (size_t)N * gi.max_plies, NA = (size_t)EPS * gi.max_plies;
    DEV_ALLOC(v.h_mine, NH * W); DEV_ALLOC(v.h_theirs, NH * W); DEV_ALLOC(v.h_counts, NH * AP);
    DEV_ALLOC(v.done, N); DEV_ALLOC(v.reward, N); DEV_ALLOC(v.chosen, N); DEV_ALLOC(v.emit_off, N);
    DEV_ALLOC(v.h_action, NH); DEV_ALLOC(v.out_actions, NA);
    DEV_ALLOC(v.ep_offset, EPS + 1); DEV_ALLOC(v.ep_slot, EPS); DEV_ALLOC(v.ep_ply, EPS);
    DEV_ALLOC(v.out_states, NA * gi.cells); DEV_ALLOC(v.out_policies, NA * A); DEV_ALLOC(v.out_values, NA);
    DEV_ALLOC(v.totals, 4); DEV_ALLOC(v.counters, 8);
    DEV_ALLOC(e->d_uniforms, N); DEV_ALLOC(e->d_cells, (size_t)N * gi.cells);
    DEV_ALLOC(e->d_tmp_mine, (size_t)N * W); DEV_ALLOC(e->d_tmp_theirs, (size_t)N * W);
    DEV_ALLOC(e->d_st_counts, (size_t)N * A); DEV_ALLOC(e->d_st_w, (size_t)N * A); DEV_ALLOC(e->d_st_prior, (size_t)N * A);
    DEV_ALLOC(e->d_st_rw, N); DEV_ALLOC(e->d_st_rn, N);
    DEV_ALLOC(e->d_noise, (size_t)N * A);
    DEV_ALLOC(e->d_st_wd, (size_t)N * A); DEV_ALLOC(e->d_st_rwd, N);

    if (c.flags & AZSP_FLAG_EVAL_CACHE) {
        if (gi.W != 1 || AP != 8)
            return fail(AZSP_ERR_ARG, "AZSP_FLAG_EVAL_CACHE needs a game whose positions have an injective 64-bit key (Connect-4, Connect2)");
        // automatic size: a generation run meets about n_slots * (n_sims + 1) positions per ply for ~30 plies, 40-50 % of
        // them distinct (measured at the headline: 5.2 M distinct of 11.2 M); 64 x the per-ply figure keeps the load
        // factor of the linear-probing table near 0.2, where EVAL_PROBES steps practically always suffice
        size_t want = c.eval_cache_entries > 0 ? (size_t)c.eval_cache_entries : (size_t)N * (size_t)(S + 1) * 64;
        if (want < 1024) want = 1024;
        if (want > ((size_t)1 << 26)) want = (size_t)1 << 26;       // entry index * 16 floats stays inside int32
        int lg = 10;
        while (((size_t)1 << lg) < want) ++lg;
        e->ent_table = (size_t)1 << lg;
        e->ent_shift = 64 - lg;
        DEV_ALLOC(e->d_ent, e->ent_table + (size_t)N);
        HIP_TRY(hipMemset(e->d_ent, 0, (e->ent_table + (size_t)N) * sizeof(EvalEnt)));
    }
    { int rc = upload_exploration_tables(e); if (rc) return rc; }
    HIP_TRY(hipMemset(v.counters, 0, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void **)&e->h_res, sizeof(PlyResultDev), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&e->h_nleaf, 64, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&e->h_counters, 64, hipHostMallocDefault));
    e->h_cells.resize((size_t)N * gi.cells);
    e->h_p.resize((size_t)N * A);
    e->h_v.resize(N);
    e->h_p8.resize((size_t)N * AP);
    e->h_mine.resize((size_t)N * W);
    e->h_theirs.resize((size_t)N * W);
    if (c.res_blocks > 0 && c.channels > 0) {
        e->net = resnet_create(c.game, c.res_blocks, c.channels, (c.flags & AZSP_FLAG_F16_SPLIT) != 0);
        if (!e->net) return fail(AZSP_ERR_ARG, "unsupported ResNet geometry: %d blocks x %d channels", c.res_blocks, c.channels);
        e->evaluator = AZSP_EVAL_RESNET;
    }
    launch_reset_slots(c.game, v, e->stream);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int azsp_create(const azsp_config *cfg, azsp_engine **out) {
    if (!cfg || !out) return fail(AZSP_ERR_ARG, "null argument");
    if (cfg->struct_size != sizeof(azsp_config)) return fail(AZSP_ERR_ARG, "azsp_config.struct_size %u != %zu", cfg->struct_size, sizeof(azsp_config));
    GameInfo gi;
    if (!game_info(cfg->game, gi)) return fail(AZSP_ERR_ARG, "unknown game id %d", cfg->game);
    if (cfg->n_slots <= 0 || cfg->n_slots > (1 << 20)) return fail(AZSP_ERR_ARG, "n_slots %d out of range", cfg->n_slots);
    if (cfg->n_sims < 0 || cfg->n_sims > 65535) return fail(AZSP_ERR_ARG, "n_sims %d out of range", cfg->n_sims);
    if (cfg->search_rules < AZSP_RULES_SIMPLE || cfg->search_rules > AZSP_RULES_SIMPLE_NUMPY1)
        return fail(AZSP_ERR_ARG, "unknown search rule set %d", cfg->search_rules);
    if (cfg->search_rules == AZSP_RULES_GEN1 && cfg->game != AZSP_GAME_CONNECT4)
        return fail(AZSP_ERR_ARG, "AZSP_RULES_GEN1 is defined for Connect-4 only");
    if (cfg->search_rules == AZSP_RULES_GEN1 && (cfg->flags & AZSP_FLAG_TREE_REUSE))
        return fail(AZSP_ERR_ARG, "AZSP_FLAG_TREE_REUSE is defined for the alphazero_simple rule sets only");
    if ((cfg->flags & AZSP_FLAG_TREE_REUSE) && (long long)gi.A * (cfg->n_sims + 1) * gi.max_plies > (1ll << 30))
        return fail(AZSP_ERR_ARG, "tree reuse: %d sims x %d plies x %d actions exceeds the node index range", cfg->n_sims, gi.max_plies, gi.A);
    if (cfg->search_rules == AZSP_RULES_GEN1 && cfg->n_sims < 1)
        return fail(AZSP_ERR_ARG, "AZSP_RULES_GEN1 needs n_sims >= 1 (the root expansion is simulation 1)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AZSP_ERR_HIP, "no HIP device available: libazsp has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(AZSP_ERR_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
    azsp_engine *e = new azsp_engine();
    e->cfg = *cfg;
    e->gi = gi;
    int rc = create_impl(e);
    if (rc) { std::string keep = g_err; azsp_destroy(e); g_err = keep; return rc; }
    *out = e;
    return 0;
}

int azsp_set_evaluator(azsp_engine *e, int32_t kind, azsp_predict_fn fn, void *user) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    if (kind == AZSP_EVAL_HOST && !fn) return fail(AZSP_ERR_ARG, "AZSP_EVAL_HOST needs a callback");
    if (kind == AZSP_EVAL_RESNET && !e->net) return fail(AZSP_ERR_STATE, "engine was created without a ResNet geometry");
    if (kind < 0 || kind > AZSP_EVAL_HOST) return fail(AZSP_ERR_ARG, "unknown evaluator kind %d", kind);
    e->evaluator = kind;
    e->cb = fn;
    e->cb_user = user;
    ON_DEVICE(e->cfg.device);
    return cache_clear(e);
}

int azsp_load_resnet(azsp_engine *e, const float *blob, size_t n_floats) {
    if (!e || !blob) return fail(AZSP_ERR_ARG, "null argument");
    if (!e->net) return fail(AZSP_ERR_STATE, "engine was created without a ResNet geometry");
    ON_DEVICE(e->cfg.device);
    const size_t want = resnet_blob_floats(e->cfg.game, e->cfg.res_blocks, e->cfg.channels);
    if (n_floats != want) return fail(AZSP_ERR_ARG, "weight blob has %zu floats, expected %zu", n_floats, want);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (resnet_load(e->net, blob, n_floats, e->stream)) return fail(AZSP_ERR_HIP, "weight upload failed");
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->net_loaded = true;
    return cache_clear(e);
}

int azsp_sync(azsp_engine *e) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    ON_DEVICE(e->cfg.device);
    return sync_stream(e);
}

static int upload_boards(azsp_engine *e, const int8_t *boards, int n) {
    const int W = e->gi.W;
    for (int i = 0; i < n; ++i)
        host_cells_to_bits(e->cfg.game, boards + (size_t)i * e->gi.cells, &e->h_mine[(size_t)i * W], &e->h_theirs[(size_t)i * W]);
    HIP_TRY(hipMemcpyAsync(e->d_tmp_mine, e->h_mine.data(), (size_t)n * W * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_tmp_theirs, e->h_theirs.data(), (size_t)n * W * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    launch_set_boards(e->cfg.game, e->v, e->d_tmp_mine, e->d_tmp_theirs, n, e->stream);
    HIP_TRY(hipStreamSynchronize(e->stream));      // h_mine/h_theirs are pageable: copies must finish before reuse
    return 0;
}

int azsp_predict(azsp_engine *e, const int8_t *boards, int32_t n, float *p_out, float *v_out) {
    if (!e || !boards || n < 0) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    const int N = e->cfg.n_slots, A = e->gi.A, AP = e->gi.AP;
    for (int lo = 0; lo < n; lo += N) {
        const int m = (n - lo < N) ? n - lo : N;
        int rc = upload_boards(e, boards + (size_t)lo * e->gi.cells, m);
        if (rc) return rc;
        e->v.n_act = m;
        e->v.lo = 0;
        e->v.grp = 0;
        launch_root_prepare(e->cfg.game, e->v, e->stream);
        launch_compact(e->v, e->stream);
        rc = evaluate_leaves(e, e->v, e->stream);
        if (rc) return rc;
        const bool in_slot = e->eval_off >= 0;         // rows of AP floats, one per slot record (value = element 7)
        if (in_slot) {
            HIP_TRY(hipMemcpy2DAsync(e->h_p8.data(), (size_t)AP * sizeof(float), e->v.eval_p, (size_t)e->v.slot_bytes,
                                     (size_t)AP * sizeof(float), (size_t)m, hipMemcpyDeviceToHost, e->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(e->h_p8.data(), e->v.eval_p, (size_t)m * AP * sizeof(float), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync(e->h_v.data(), e->v.eval_v, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        }
        rc = sync_stream(e);
        if (rc) return rc;
        for (int j = 0; j < m; ++j) {
            if (p_out) for (int a = 0; a < A; ++a) p_out[(size_t)(lo + j) * A + a] = e->h_p8[(size_t)j * AP + a];
            if (v_out) v_out[lo + j] = in_slot ? e->h_p8[(size_t)j * AP + 7] : e->h_v[j];
        }
    }
    e->v.n_act = N;
    e->searched = false;
    return 0;
}

int azsp_probe_clock(azsp_engine *e, const int8_t *boards, int32_t n, double *clock_ghz, double *span_us) {
    if (!e || !boards || n <= 0 || n > e->cfg.n_slots) return fail(AZSP_ERR_ARG, "bad argument");
    if (e->evaluator != AZSP_EVAL_RESNET || e->cfg.game != AZSP_GAME_CONNECT4 || e->cfg.channels > 64 || e->cfg.channels <= 32)
        return fail(AZSP_ERR_STATE, "the clock probe exists for the 64-channel Connect-4 network (fp32 and f16x2 kernels)");
    resnet_trace_next();
    int rc = azsp_predict(e, boards, n, nullptr, nullptr);
    if (rc) return rc;
    resnet_trace_result(clock_ghz, span_us);
    if (clock_ghz && *clock_ghz <= 0.0) return fail(AZSP_ERR_STATE, "no traced launch ran");
    return 0;
}

int azsp_run_batch(azsp_engine *e, const int8_t *boards, int32_t n, int32_t *counts, float *child_w, double *prior,
                   float *root_w, int32_t *root_n) {
    if (!e || !boards || n <= 0) return fail(AZSP_ERR_ARG, "bad argument");
    if (n > e->cfg.n_slots) return fail(AZSP_ERR_ARG, "n %d exceeds n_slots %d", n, e->cfg.n_slots);
    ON_DEVICE(e->cfg.device);
    const int A = e->gi.A;
    int rc = upload_boards(e, boards, n);
    if (rc) return rc;
    rc = search(e, n);
    if (rc) return rc;
    int32_t *d_counts = e->d_st_counts, *d_rn = e->d_st_rn;
    float *d_w = e->d_st_w, *d_rw = e->d_st_rw;
    double *d_prior = e->d_st_prior;
    launch_root_stats(e->cfg.game, e->v, d_counts, d_w, d_prior, d_rw, d_rn, nullptr, nullptr, e->stream);
    if (counts) HIP_TRY(hipMemcpyAsync(counts, d_counts, (size_t)n * A * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (child_w) HIP_TRY(hipMemcpyAsync(child_w, d_w, (size_t)n * A * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (prior) HIP_TRY(hipMemcpyAsync(prior, d_prior, (size_t)n * A * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (root_w) HIP_TRY(hipMemcpyAsync(root_w, d_rw, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (root_n) HIP_TRY(hipMemcpyAsync(root_n, d_rn, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    rc = sync_stream(e);
    e->v.n_act = e->cfg.n_slots;
    e->searched = false;          // the slots no longer hold a generation run
    return rc;
}

int azsp_reset(azsp_engine *e) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    ON_DEVICE(e->cfg.device);
    e->v.n_act = e->cfg.n_slots;
    launch_reset_slots(e->cfg.game, e->v, e->stream);
    e->searched = false;
    e->play_open = false;
    int rc = cache_clear(e);         // a position is evaluated at most once per generation run, never across runs
    if (rc) return rc;
    rc = sync_stream(e);
    if (rc) return rc;
    memset(e->h_res, 0, sizeof(PlyResultDev));      // the fetch/export calls check their ranges against this mirror
    return 0;
}

int azsp_search(azsp_engine *e) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    ON_DEVICE(e->cfg.device);
    return search(e, e->cfg.n_slots);
}

int azsp_set_search_options(azsp_engine *e, const azsp_search_options *o) {
    if (!e || !o) return fail(AZSP_ERR_ARG, "null argument");
    if (o->struct_size != sizeof(azsp_search_options)) return fail(AZSP_ERR_ARG, "azsp_search_options.struct_size mismatch");
    if (o->noise_eps < 0.0 || o->noise_eps > 1.0) return fail(AZSP_ERR_ARG, "noise_eps out of [0,1]");
    if (!(o->c_init == o->c_init) || !(o->c_base == o->c_base) || o->c_base < 0.0)
        return fail(AZSP_ERR_ARG, "c_init / c_base must be numbers, c_base >= 0");
    ON_DEVICE(e->cfg.device);
    HIP_TRY(hipStreamSynchronize(e->stream));      // no search may be reading the tables
    e->v.sampling_plies = o->sampling_plies;
    e->v.noise_eps = o->noise_eps;
    e->v.noise_keep = 1.0 - o->noise_eps;
    e->c_init = o->c_init;
    e->c_base = o->c_base;
    return upload_exploration_tables(e);
}

int azsp_search_with_noise(azsp_engine *e, const double *noise) {
    if (!e || !noise) return fail(AZSP_ERR_ARG, "null argument");
    ON_DEVICE(e->cfg.device);
    HIP_TRY(hipMemcpyAsync(e->d_noise, noise, (size_t)e->cfg.n_slots * e->gi.A * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));            // the caller's buffer may be pageable
    e->v.root_noise = e->d_noise;
    int rc = search(e, e->cfg.n_slots);
    e->v.root_noise = nullptr;
    return rc;
}

int azsp_root_counts(azsp_engine *e, int32_t *counts) {
    if (!e || !counts) return fail(AZSP_ERR_ARG, "null argument");
    if (!e->searched) return fail(AZSP_ERR_STATE, "azsp_root_counts before azsp_search");
    ON_DEVICE(e->cfg.device);
    const int N = e->cfg.n_slots, A = e->gi.A;
    int32_t *d_counts = e->d_st_counts;
    launch_root_stats(e->cfg.game, e->v, d_counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, e->stream);
    HIP_TRY(hipMemcpyAsync(counts, d_counts, (size_t)N * A * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e);
}

int azsp_root_values(azsp_engine *e, int32_t n, double *child_w, double *root_w, int32_t *root_n) {
    if (!e || n <= 0 || n > e->cfg.n_slots) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    const int A = e->gi.A;
    View v = e->v;
    v.n_act = n;
    v.lo = 0;
    launch_root_stats(e->cfg.game, v, nullptr, nullptr, nullptr, nullptr, e->d_st_rn, e->d_st_wd, e->d_st_rwd, e->stream);
    if (child_w) HIP_TRY(hipMemcpyAsync(child_w, e->d_st_wd, (size_t)n * A * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (root_w) HIP_TRY(hipMemcpyAsync(root_w, e->d_st_rwd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (root_n) HIP_TRY(hipMemcpyAsync(root_n, e->d_st_rn, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e);
}

int azsp_set_initial_board(azsp_engine *e, const int8_t *cells) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    u64 m[2] = {0, 0}, t[2] = {0, 0};
    if (cells) {
        int32_t term = 0;
        int rc = azsp_game_rules(e->cfg.game, cells, -1, nullptr, nullptr, &term);
        if (rc) return rc;
        if (term != TERM_NONE) return fail(AZSP_ERR_ARG, "the initial position is already decided");
        host_cells_to_bits(e->cfg.game, cells, m, t);
    }
    for (int i = 0; i < 2; ++i) { e->v.init_mine[i] = m[i]; e->v.init_theirs[i] = t[i]; }
    return 0;
}

int azsp_fetch_moves(azsp_engine *e, int64_t first, int64_t n, int32_t *moves, int64_t sample_cap) {
    if (!e || first < 0 || n < 0 || !moves) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    if (first + n > e->h_res->episodes_total) return fail(AZSP_ERR_ARG, "episodes not emitted yet");
    if (n == 0) return 0;
    int64_t off[2];
    HIP_TRY(hipMemcpy(&off[0], e->v.ep_offset + first, sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&off[1], e->v.ep_offset + first + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t ns = off[1] - off[0];
    if (ns > sample_cap) return fail(AZSP_ERR_ARG, "sample_cap %lld < %lld samples", (long long)sample_cap, (long long)ns);
    std::vector<uint8_t> tmp((size_t)ns);
    HIP_TRY(hipMemcpy(tmp.data(), e->v.out_actions + off[0], (size_t)ns, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < ns; ++i) moves[i] = tmp[(size_t)i];
    return 0;
}

int azsp_play_begin(azsp_engine *e, const double *uniforms, int32_t *n_done) {
    if (!e || !uniforms) return fail(AZSP_ERR_ARG, "null argument");
    if (!e->searched) return fail(AZSP_ERR_STATE, "azsp_play before azsp_search");
    if (e->cfg.n_sims == 0)
        return fail(AZSP_ERR_STATE, "n_sims = 0: the visit distribution is 0/0 (the reference raises in np.random.choice)");
    if (e->cfg.search_rules == AZSP_RULES_GEN1 && e->cfg.n_sims < 2)
        return fail(AZSP_ERR_STATE, "n_sims = 1 under AZSP_RULES_GEN1: pi = n / (N - 1) divides by zero (node.py:24-29 raises)");
    ON_DEVICE(e->cfg.device);
    const int N = e->cfg.n_slots;
    HIP_TRY(hipMemcpyAsync(e->d_uniforms, uniforms, (size_t)N * sizeof(double), hipMemcpyHostToDevice, e->stream));
    EventPair *p = ev_begin(e, EV_PLAY, e->stream);
    launch_finish(e->cfg.game, e->v, e->d_uniforms, e->stream);
    if (n_done) launch_count_done(e->v, e->h_nleaf, e->stream);
    ev_end(p, e->stream);
    e->searched = false;
    e->play_open = true;
    if (n_done) {
        int rc = sync_stream(e);
        if (rc) return rc;
        *n_done = *e->h_nleaf;
    }
    return 0;
}

int azsp_play_commit(azsp_engine *e, int64_t allow, azsp_ply_result *out) {
    if (!e || !out) return fail(AZSP_ERR_ARG, "null argument");
    if (!e->play_open) return fail(AZSP_ERR_STATE, "azsp_play_commit without azsp_play_begin");
    ON_DEVICE(e->cfg.device);
    EventPair *p = ev_begin(e, EV_PLAY, e->stream);
    launch_emit(e->cfg.game, e->v, e->h_res, allow < 0 ? -1ll : (long long)allow, e->stream);
    ev_end(p, e->stream);
    e->play_open = false;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(AZSP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
    int rc = sync_stream(e);
    if (rc) return rc;
    out->ply = e->h_res->ply;
    out->n_finished = e->h_res->n_finished;
    out->episodes_total = e->h_res->episodes_total;
    out->samples_total = e->h_res->samples_total;
    out->uniforms_used = e->h_res->uniforms_used;
    out->finished = e->h_res->finished;
    out->last_counted_slot = e->h_res->last_counted_slot;
    out->reserved = 0;
    return 0;
}

int azsp_play(azsp_engine *e, const double *uniforms, azsp_ply_result *out) {
    if (!out) return fail(AZSP_ERR_ARG, "null argument");
    int rc = azsp_play_begin(e, uniforms, nullptr);
    if (rc) return rc;
    return azsp_play_commit(e, -1, out);
}

int azsp_fetch_episodes(azsp_engine *e, int64_t first, int64_t n, int64_t *ep_offset, int32_t *ep_slot, int32_t *ep_ply,
                        int8_t *states, double *policies, int32_t *values, int64_t sample_cap) {
    if (!e || first < 0 || n < 0) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    // No stream synchronisation: emitted episodes were complete when azsp_play returned, and the arena
    // is append-only until the next azsp_play, so a search already enqueued may run during the copy.
    if (first + n > e->h_res->episodes_total) return fail(AZSP_ERR_ARG, "episodes [%lld,%lld) not emitted yet (have %lld)", (long long)first, (long long)(first + n), (long long)e->h_res->episodes_total);
    if (n == 0) return 0;
    std::vector<int64_t> off((size_t)n + 1);
    HIP_TRY(hipMemcpy(off.data(), e->v.ep_offset + first, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t s0 = off[0], ns = off[n] - off[0];
    if ((states || policies || values) && ns > sample_cap) return fail(AZSP_ERR_ARG, "sample_cap %lld < %lld samples", (long long)sample_cap, (long long)ns);
    if (ep_offset) for (int64_t i = 0; i <= n; ++i) ep_offset[i] = off[i] - s0;
    if (ep_slot) HIP_TRY(hipMemcpy(ep_slot, e->v.ep_slot + first, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (ep_ply) HIP_TRY(hipMemcpy(ep_ply, e->v.ep_ply + first, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int A = e->gi.A, cells = e->gi.cells;
    if (states) HIP_TRY(hipMemcpy(states, e->v.out_states + s0 * cells, (size_t)ns * cells, hipMemcpyDeviceToHost));
    if (policies) HIP_TRY(hipMemcpy(policies, e->v.out_policies + s0 * A, (size_t)ns * A * sizeof(double), hipMemcpyDeviceToHost));
    if (values) HIP_TRY(hipMemcpy(values, e->v.out_values + s0, (size_t)ns * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int64_t azsp_export_records_device(azsp_engine *e, int64_t first, int64_t n, void *dev_dst, int64_t cap_records) {
    if (!e || first < 0 || n < 0 || !dev_dst) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(AZSP_ERR_HIP, "stream sync failed");
    if (first + n > e->h_res->episodes_total) return fail(AZSP_ERR_ARG, "episodes not emitted yet");
    if (n == 0) return 0;
    int64_t off[2];
    if (hipMemcpy(&off[0], e->v.ep_offset + first, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&off[1], e->v.ep_offset + first + n, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(AZSP_ERR_HIP, "ep_offset copy failed");
    const int64_t ns = off[1] - off[0];
    if (ns > cap_records) return fail(AZSP_ERR_ARG, "cap_records %lld < %lld", (long long)cap_records, (long long)ns);
    launch_export_records(e->cfg.game, e->v, off[0], ns, dev_dst, e->stream);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(AZSP_ERR_HIP, "export kernel failed: %s", hipGetErrorString(hipGetLastError()));
    return ns;
}

int64_t azsp_export_dataset_device(azsp_engine *e, int64_t first, int64_t n, float *planes, float *policies, float *values,
                                   int64_t cap_samples) {
    if (!e || first < 0 || n < 0 || !planes || !policies || !values) return fail(AZSP_ERR_ARG, "bad argument");
    ON_DEVICE(e->cfg.device);
    if (first + n > e->h_res->episodes_total) return fail(AZSP_ERR_ARG, "episodes not emitted yet");
    if (n == 0) return 0;
    int64_t off[2];
    if (hipMemcpy(&off[0], e->v.ep_offset + first, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&off[1], e->v.ep_offset + first + n, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(AZSP_ERR_HIP, "ep_offset copy failed");
    const int64_t ns = off[1] - off[0];
    if (ns > cap_samples) return fail(AZSP_ERR_ARG, "cap_samples %lld < %lld", (long long)cap_samples, (long long)ns);
    // separate stream-ordered launch on the engine's main stream; the arena rows are final
    launch_export_dataset(e->cfg.game, e->v, off[0], ns, planes, policies, values, e->stream);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(AZSP_ERR_HIP, "export kernel failed: %s", hipGetErrorString(hipGetLastError()));
    return ns;
}

int azsp_get_stats(azsp_engine *e, azsp_stats *out) {
    if (!e || !out) return fail(AZSP_ERR_ARG, "null argument");
    ON_DEVICE(e->cfg.device);
    int rc = sync_stream(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->h_counters, e->v.counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    e->stats.evals = e->h_counters[0];
    e->stats.terminal_leaves = e->h_counters[1];
    e->stats.path_nodes = e->h_counters[2];
    e->stats.rounds = e->rounds_host;
    e->stats.infer_union_ms = e->infer_union_ms;
    e->stats.evals_executed = e->h_counters[4];
    e->stats.infer_boards = e->evaluator == AZSP_EVAL_RESNET ? e->h_counters[4] : 0;
    e->stats.infer_exec_flops = e->net ? resnet_exec_flops_total(e->net, false) : 0.0;
    e->stats.infer_exec_flops_per_board = e->net ? resnet_exec_flops_per_board(e->net, e->cfg.game, e->last_infer_cap) : 0.0;
    e->stats.eval_cache_entries = e->d_ent ? (uint64_t)e->ent_table : 0;
    *out = e->stats;
    return 0;
}

int azsp_reset_stats(azsp_engine *e) {
    if (!e) return fail(AZSP_ERR_ARG, "null engine");
    ON_DEVICE(e->cfg.device);
    int rc = sync_stream(e);
    if (rc) return rc;
    HIP_TRY(hipMemset(e->v.counters, 0, 8 * sizeof(unsigned long long)));
    if (e->net) (void)resnet_exec_flops_total(e->net, true);
    e->stats = azsp_stats{};
    e->rounds_host = 0;
    e->infer_union_ms = 0;
    return 0;
}

}  // extern "C"
