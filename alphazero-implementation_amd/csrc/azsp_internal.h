// Internal declarations shared by the translation units of libazsp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "azsp_games.h"

namespace azsp {

constexpr int MAX_GROUPS = 8; // slot groups searched concurrently on separate HIP streams

// ------------------------------------------------------------------------------------------------
// Search tree in HBM: one record per EXPANDED node ("block"), holding all its children.
// ------------------------------------------------------------------------------------------------
// A descent touches one record per level, so a record is one aligned unit that the lanes of a game fetch with a single
// 16-byte load each: A child chunks {N, W, first-child block, prior} + one header chunk, padded to a multiple of 128
// bytes (Connect-4: 7 x 16 + 16 = 128 B = exactly one cache line per level; Gomoku 9x9: 82 x 16 -> 1408 B).
//   * N  visit_count, W value_sum (float32: numpy-2 semantics; the float64 rule sets keep W in the side array wsum_d)
//   * kid  0 while the child is unexpanded, else 1 + index of the block holding ITS children
//   * prior: the reference's float64 prior is f64(p32 * mask) / f64 sum (monte_carlo_tree_search.py:119-121); the record
//     keeps the float32 numerator per child and the float64 sum once per block and every visit redoes the (IEEE,
//     deterministic) division -- 4 bytes per child instead of 8 and nothing lost.  Rule set 1 uses the model's float32
//     as it comes (sum = 1, -1 marks "no child").  Root noise (an option beyond the reference) replaces the root's priors
//     by float64 values in the side array root_prior (flags bit 0 of the root's block header).
//   * node ids (path entries, leaf_node, root): 0 = the root of a fresh tree (its own N / W live in the slot record),
//     1 + e * A + a = child a of block e.  Blocks are bump-allocated in expansion order (deterministic).
//   * positions are not stored: the descent replays the moves from the slot's current position (G::play per level).
struct alignas(16) NodeChunk { int32_t n; float w; int32_t kid; float p; };
struct alignas(16) BlockHead { double psum; int32_t flags; int32_t parent; };

// Per-slot search state, one aligned record per slot (read with one or two 16-byte loads per lane at kernel start):
//   chunk 0  pending (leaf awaits evaluation), path_len (0 for a root expansion: no backup), leaf_node, n_blk
//   chunk 1  root (node id of the current root; 0 unless the tree is being reused), root_kid (fresh root: 1 + block of its
//            children, 0 = unexpanded), root_n / root_w (N and W of a fresh root)
//   chunk 2  next_root (tree reuse: child chosen by k_finish, or -1), sims_done (FUSED rounds: simulations of this ply done),
//            eval_ent (evaluation cache: the entry that holds, or will hold, the evaluator's row for the pending leaf)
//   then     leaf position u64 mine[W], theirs[W]; then int32 path[MAX_PLIES + 2] (root .. leaf of the current descent)
struct alignas(16) SlotHead {
    int32_t pending, path_len, leaf_node, n_blk;
    int32_t root, root_kid, root_n;
    float root_w;
    int32_t next_root, sims_done, eval_ent, pad2;
};
template <class G>
struct TreeGeo {
    static constexpr int A = G::A, LPG = G::LPG, APL = G::APL;
    static constexpr int BLOCK_BYTES = ((A + 1) * 16 + 127) & ~127;
    static constexpr int HEAD_CHUNKS = 3;                                  // SlotHead
    static constexpr int LEAF_CHUNK = HEAD_CHUNKS;                         // first chunk of the leaf position
    // narrow action spaces (A <= 7: AP = 8 floats): the evaluator writes the leaf's (p[0..A), value at [7]) INTO the slot
    // record, right behind the leaf position -- the 32 bytes arrive with the record's first line instead of as two more
    // dependent requests per simulation; wide ones (Gomoku: 88 floats) keep a separate row per slot
    static constexpr bool EVAL_IN_SLOT = G::AP == 8;
    static constexpr int EVAL_CHUNK = HEAD_CHUNKS + G::W;                  // first chunk of the evaluator's row (EVAL_IN_SLOT)
    static constexpr int PATH_CHUNK = EVAL_CHUNK + (EVAL_IN_SLOT ? 2 : 0); // first chunk of the path
    static constexpr int D = G::MAX_PLIES + 2;                             // path capacity
    static constexpr int SLOT_BYTES = (PATH_CHUNK * 16 + D * 4 + 127) & ~127;
    static constexpr int SLOT_CHUNKS = SLOT_BYTES / 16;
    static constexpr int CPL = (SLOT_CHUNKS + LPG - 1) / LPG;              // slot chunks per lane
    static_assert(APL * LPG >= A + 1, "a block (children + header) must be covered by APL 16-byte loads per lane");
    static_assert(LPG > PATH_CHUNK, "the slot head and the leaf position come with the first 16-byte load of each lane");
};

// ------------------------------------------------------------------------------------------------
// Evaluation cache (AZSP_FLAG_EVAL_CACHE): position -> (p, v), so that a position is evaluated ONCE per generation run.
// ------------------------------------------------------------------------------------------------
// The reference evaluates every leaf of every game in every round (monte_carlo_tree_search.py:154-157) and rebuilds
// every tree every ply (:116-123); with no noise in the search, equal positions get equal (p, v) from a deterministic
// evaluator, and the device evaluators are bitwise independent of the batch a position sits in (tested), so looking a
// position's row up instead of recomputing it changes no bit of the search.
// One array of 64-byte entries: [0, n_slots) one PRIVATE entry per slot (positions without a key, a full table, or a
// position another slot group inserted during this very search), then a power-of-two open-addressing table (linear
// probing, at most EVAL_PROBES steps).  An entry is claimed by ONE 64-bit compare-and-swap on its key word:
//   bits 0..48  G::pos_key (injective, so a key match IS a position match -- no second compare, no lock)
//   bits 49..51 slot group of the claiming launch      bits 52..63 epoch = number of the search (1..4095)
// The claiming game ("owner") stores the position and puts the ENTRY on its group's leaf list; the inference kernel
// reads the position from the entry and writes (p, v) into it; every game -- owner or not -- keeps the entry's index in
// its slot record and reads the row from there when it expands the leaf in its group's next tree launch.  Ordering
// needs nothing beyond streams and events: a row is read only (a) from an entry claimed by the reader's own slot group
// (its inference launch sits between the claim and the read on the group's stream), (b) from an entry claimed in an
// earlier search (all group streams join the main stream at the end of every search), or (c) from an entry a LOWER
// group claimed during this search -- the groups' tree steps run as a ring (search_enqueue), which puts that group's
// inference launch before the read.  A key claimed by a HIGHER group during the current search is not usable yet: the
// game falls back to its private entry (a duplicate evaluation, same bits).
struct alignas(64) EvalEnt { u64 key; u64 mine, theirs, pad; float row[8]; };   // row = p[0..A), value at [7]  (AP == 8 games)
static_assert(sizeof(EvalEnt) == 64, "the inference kernels address entries with strides of 8 words / 16 floats");
constexpr int EVAL_PROBES = 16;
constexpr u64 EVAL_POS_MASK = (1ull << 49) - 1;

// Device-resident state of one engine.
struct View {
    int32_t lo;        // first slot of the group this launch works on (groups run on separate streams)
    int32_t grp;       // group index: selects n_leaf[grp]
    int32_t n_act;     // slots in the group: [lo, lo + n_act)
    int32_t n_slots;
    int32_t cap_eps;   // episode capacity of the arena (>= n_slots)
    int32_t A;         // actions
    int32_t C;         // node-id capacity per tree (stride of the wsum_d side array)
    int32_t E;         // block capacity per tree = n_sims + 1 (a whole game's worth with tree reuse)
    int32_t D;         // path capacity = MAX_PLIES + 2
    int32_t n_sims;
    int32_t max_plies; // history capacity per slot
    int32_t cells;
    int32_t rules;     // AZSP_RULES_*: which of the reference's two search rule sets the kernels follow
    int32_t reuse;     // AZSP_FLAG_TREE_REUSE: the chosen child's subtree becomes the next ply's tree

    // search tree
    uint8_t *rec;           // [n_slots][E] blocks of rec_block bytes (TreeGeo<G>::BLOCK_BYTES)
    int64_t  rec_slot;      // bytes per slot = E * rec_block
    int32_t  rec_block;
    int32_t  slot_bytes;    // TreeGeo<G>::SLOT_BYTES
    uint8_t *slot;          // [n_slots] slot records
    double  *wsum_d;        // W as float64 by node id [n_slots][C]: rule sets 1 and 2 (else nullptr)
    double  *root_prior;    // [n_slots][A] float64 root priors when root noise was mixed in
    // per-round lists
    int32_t *leaf_slot;     // [n_slots] per group, at offset lo: list of the slots whose leaf awaits evaluation
    int32_t *n_leaf;        // [MAX_GROUPS][4] leaf counters (the fused tree kernel alternates two per group, FUSED rounds rotate three)
    int32_t *leaf_slot2;    // [n_slots] second leaf list (FUSED rounds read one list while filling the other)
    int32_t *nl;            // the counter this launch reads / fills: n_leaf + grp * 2 + parity
    int32_t *nl_other;      // the group's other counter (the fused kernel zeroes it for the round after)
    float   *eval_p;        // evaluator output of slot g: p at eval_p + g * eval_pstride, value at eval_v + g * eval_vstride
    float   *eval_v;        // (inside the slot records when TreeGeo<G>::EVAL_IN_SLOT, else [n_slots][AP] and [n_slots])
    int32_t  eval_pstride, eval_vstride;      // in floats
    const double *sqrt_tab; // [tab_len] host-computed sqrt(N): bit-identical to math.sqrt
    const double *c_tab;    // [tab_len] exploration factor c(N): 1.0 unless options say otherwise
    const double *expl_tab; // [tab_len] c(N) * sqrt(N)  (== sqrt_tab bit for bit while c == 1)
    int32_t tab_len;        // n_sims + 2, or a whole game's simulations with tree reuse
    int32_t tab_lds;        // how many leading entries the tree kernels stage in LDS
    // search options beyond the reference (all off by default)
    const double *root_noise;   // [n_slots][A] or nullptr
    double  noise_eps, noise_keep;   // eps and (1 - eps), both computed on the host
    int32_t sampling_plies;     // < 0: always sample (reference behaviour)
    // live games
    u64     *cur_mine, *cur_theirs;     // [n_slots][W] canonical position (side to move = mine)
    int32_t *cur_len;                   // [n_slots] plies played in the current game
    u64     *h_mine, *h_theirs;         // [n_slots][max_plies][W] history: root positions
    int32_t *h_counts;                  // [n_slots][max_plies][AP] root child visit counts
    uint8_t *done;                      // [n_slots] game ended this ply
    int8_t  *reward;                    // [n_slots] reward for the side to move at the end
    int32_t *chosen;                    // [n_slots] action sampled this ply
    uint8_t *h_action;                  // [n_slots][max_plies] history: the action taken from each root position
    u64     init_mine[2], init_theirs[2];   // position every game starts from (canonical; all zero = empty board)
    // episode arena
    int32_t *emit_off;                  // [n_slots] sample offset of the slot's episode, -1 if none
    int64_t *ep_offset;                 // [n_slots + 1]
    int32_t *ep_slot, *ep_ply;          // [n_slots]
    int8_t  *out_states;                // [cap][cells]
    double  *out_policies;              // [cap][A]
    int32_t *out_values;                // [cap]
    uint8_t *out_actions;               // [cap] move played from the sample's position
    int64_t *totals;                    // [0] episodes, [1] samples, [2] ply index
    // evaluation cache (nullptr = off: evaluator rows live in the slot records / eval_p as above)
    EvalEnt *ent;
    uint32_t ent_mask;      // table entries - 1 (the table starts at entry n_slots)
    int32_t  ent_shift;     // 64 - log2(table entries)
    int32_t  ent_epoch;     // number of this search, 1..4095
    // device-side counters: [0] positions that needed the evaluator (the reference's evaluations), [1] terminal leaves,
    // [2] path nodes, [3] rounds, [4] positions put on a leaf list = evaluations executed (== [0] without the cache)
    unsigned long long *counters;
};

struct PlyResultDev {   // mirrors azsp_ply_result; written by k_emit_scan into pinned host memory
    int32_t ply, n_finished;
    int64_t episodes_total, samples_total;
    int32_t uniforms_used, finished;
    int32_t last_counted_slot, pad;
};

// launchers (azsp_tree.hip)
void tree_geometry(int game, int32_t *block_bytes, int32_t *slot_bytes, int32_t *path_cap, int32_t *eval_off);   // eval_off: byte offset of the evaluator's row in a slot record, -1 = separate arrays
void launch_root_prepare(int game, const View &v, hipStream_t s);
void launch_select(int game, const View &v, hipStream_t s);
void launch_compact(const View &v, hipStream_t s);
void launch_eval_hash(int game, const View &v, uint64_t salt, hipStream_t s);      // (rows into the cache entries when v.ent)
void launch_expand(int game, const View &v, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_expand_select(int game, const View &v, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);   // expand of round r-1 fused with select of round r
void launch_leaf_cells(int game, const View &v, int8_t *dst, hipStream_t s);
void launch_scatter_eval(int game, const View &v, const float *p_compact, const float *v_compact, hipStream_t s);   // host evaluator: compact order -> slot order
void launch_finish(int game, const View &v, const double *uniforms, hipStream_t s);
void launch_count_done(const View &v, int32_t *out, hipStream_t s);
void launch_emit(int game, const View &v, PlyResultDev *host_result, long long cap, hipStream_t s);
void launch_reset_slots(int game, const View &v, hipStream_t s);
void launch_set_boards(int game, const View &v, const u64 *mine, const u64 *theirs, int n, hipStream_t s);
void launch_root_stats(int game, const View &v, int32_t *counts, float *child_w, double *prior, float *root_w,
                       int32_t *root_n, double *child_wd, double *root_wd, hipStream_t s);
void launch_export_records(int game, const View &v, int64_t first_sample, int64_t n_samples, void *dst,
                           hipStream_t s);

void launch_export_dataset(int game, const View &v, int64_t first_sample, int64_t n_samples, float *planes, float *pol,
                           float *val, hipStream_t s);

// ResNet evaluator (azsp_resnet.hip)
struct ResNetDev;
ResNetDev *resnet_create(int game, int blocks, int channels, bool f16_split);
void resnet_destroy(ResNetDev *);
int resnet_load(ResNetDev *, const float *blob, size_t n_floats, hipStream_t s);   // 0 ok
size_t resnet_blob_floats(int game, int blocks, int channels);
// evaluates the leaf positions of the slots leaf_slot[j], j < *n_leaf (<= max_leaves), into out_p[slot][AP] / out_v[slot];
// the position of slot g: mine = leaf0 + g * board_stride (W words), theirs = the W words after it
// by_count: pick the kernel variant by the list's length on the device (all variants enqueued, gated), not by max_leaves
void resnet_forward(ResNetDev *, int game, const u64 *leaf0, int board_stride, const int32_t *leaf_slot,
                    const int32_t *n_leaf, int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s,
                    bool by_count = false, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// (ev_start / ev_stop: optional events that take the launch's own start / end timestamps from its dispatch packet)
// (results of slot g: p at out_p + g * p_stride, value at out_v + g * v_stride -- strides in floats)
// FUSED rounds (one-board Connect-4 kernels of the fp32 network, up to 32 games): one launch evaluates the listed leaves
// AND carries those games on to their next leaf (azsp_tree_dev.h advance_game)
bool resnet_has_fused(const ResNetDev *, int game, int max_leaves, int tab_lds);
void resnet_forward_fused(ResNetDev *, int game, const View &view, const u64 *leaf0, int board_stride, const int32_t *leaf_slot,
                          const int32_t *n_leaf, int32_t *next_list, int32_t *next_count, int32_t *zero_count, int max_leaves,
                          hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
double resnet_flops_per_board(const ResNetDev *);
double resnet_exec_flops_per_board(const ResNetDev *, int game, int max_leaves);   // matrix-pipe FLOP executed per board (variant of that capacity)
double resnet_exec_flops_total(const ResNetDev *, bool reset);   // summed on the device over every launch since the last reset
// diagnostics: arm the stamped instantiation for this thread's next fp32 launch; read back its clock / span
void resnet_trace_next();
void resnet_trace_result(double *ghz, double *span_us);

// split-fp16 variant (azsp_resnet_f16.hip)
struct ResNet16Dev;
ResNet16Dev *resnet16_create(int game, int blocks, int channels);
void resnet16_destroy(ResNet16Dev *);
int resnet16_load(ResNet16Dev *, const float *blob, size_t n_floats, hipStream_t s);
void resnet16_forward(ResNet16Dev *, const u64 *leaf0, int board_stride, const int32_t *leaf_slot, const int32_t *n_leaf,
                      int max_leaves, float *out_p, int p_stride, float *out_v, int v_stride, hipStream_t s, bool by_count = false,
                      double *flop_acc = nullptr, double flops_per_board = 0.0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
double resnet16_exec_flops_per_board(const ResNet16Dev *, double flops_per_board, int max_leaves);      // executed fp16 MFMA FLOP per board of the form a launch of that capacity takes

}  // namespace azsp
