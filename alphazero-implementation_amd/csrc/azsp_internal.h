// Internal declarations shared by the translation units of libazsp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "azsp_games.h"

namespace azsp {

constexpr int MAX_GROUPS = 8; // slot groups searched concurrently on separate HIP streams

// Device-resident state of one engine.  Structure-of-arrays; every per-node array is [slot][C].
//
//   node 0            root (of a fresh tree; with tree reuse root[slot] walks down the same arrays)
//   node 1 + e*A + a  child for action a of the e-th expanded node of this tree ("block" e);
//                     blocks are bump-allocated in expansion order (deterministic).
//   kids[x]           first child of x (1 + e*A), 0 while x is unexpanded
//   blk_parent[e]     the node whose children are block e;  st_*[e] = its canonical position
struct View {
    int32_t lo;        // first slot of the group this launch works on (groups run on separate streams)
    int32_t grp;       // group index: selects n_leaf[grp]
    int32_t n_act;     // slots in the group: [lo, lo + n_act)
    int32_t n_slots;
    int32_t cap_eps;   // episode capacity of the arena (>= n_slots)
    int32_t A;         // actions
    int32_t C;         // node capacity per tree (array stride)
    int32_t E;         // block capacity per tree = n_sims + 1
    int32_t D;         // path stride
    int32_t n_sims;
    int32_t max_plies; // history capacity per slot
    int32_t cells;
    int32_t rules;     // AZSP_RULES_*: which of the reference's two search rule sets the kernels follow
    int32_t reuse;     // AZSP_FLAG_TREE_REUSE: the chosen child's subtree becomes the next ply's tree

    // search tree (SoA)
    int32_t *cnt;       // N  visit_count                       [n_slots][C]
    float   *wsum;      // W  value_sum (float32, numpy-2 semantics)              rule set 0
    double  *wsum_d;    // W  value_sum (Python float)                           rule set 1 (else nullptr)
    double  *prior;     // P  prior (float64; 0 = no child, monte_carlo_tree_search.py:87)
    int32_t *kids;      // first-child index
    int32_t *blk_parent;    // [n_slots][E]
    u64     *st_mine, *st_theirs;   // [n_slots][E][W]   (W = 64-bit words per side, from the game trait)
    int32_t *n_blk;         // [n_slots]
    int32_t *root;          // [n_slots] node index of the current root (0 unless the tree is being reused)
    int32_t *next_root;     // [n_slots] child chosen by k_finish, or -1: the next search starts a fresh tree
    // per-round scratch
    int32_t *path;          // [n_slots][D]  root .. leaf of the current descent
    int32_t *path_len;      // [n_slots]     0 for a root expansion (no backup)
    int32_t *leaf_node;     // [n_slots]
    u64     *leaf_mine, *leaf_theirs;   // [n_slots][W] canonical position at the leaf
    uint8_t *pending;       // [n_slots] leaf awaits evaluation
    int32_t *leaf_slot;     // [n_slots] per group, at offset lo: compacted (slot order) list of pending slots
    int32_t *n_leaf;        // [MAX_GROUPS][2] leaf counters (two per group: the fused tree kernel alternates them)
    int32_t *nl;            // the counter this launch reads / fills: n_leaf + grp * 2 + parity
    int32_t *nl_other;      // the group's other counter (the fused kernel zeroes it for the round after)
    int32_t *leaf_index;    // [n_slots] compacted index (absolute, lo-based) of the slot's pending leaf
    float   *eval_p;        // [n_slots][AP] evaluator output (AP = A padded to 8), at offset lo, indexed by compacted leaf
    float   *eval_v;        // [n_slots]
    const double *sqrt_tab; // [n_sims + 2, or a whole game's simulations with tree reuse] host-computed sqrt(N): bit-identical to math.sqrt
    const double *c_tab;    // [n_sims + 2] exploration factor c(N): 1.0 unless options say otherwise
    const double *expl_tab; // [n_sims + 2] c(N) * sqrt(N)  (== sqrt_tab bit for bit while c == 1)
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
    // device-side counters: [0] evals, [1] terminal leaves, [2] path nodes, [3] rounds
    unsigned long long *counters;
};

struct PlyResultDev {   // mirrors azsp_ply_result; written by k_emit_scan into pinned host memory
    int32_t ply, n_finished;
    int64_t episodes_total, samples_total;
    int32_t uniforms_used, finished;
    int32_t last_counted_slot, pad;
};

// launchers (azsp_tree.hip)
void launch_root_prepare(int game, const View &v, hipStream_t s);
void launch_select(int game, const View &v, hipStream_t s);
void launch_compact(const View &v, hipStream_t s);
void launch_eval_hash(int game, const View &v, uint64_t salt, hipStream_t s);
void launch_expand(int game, const View &v, hipStream_t s);
void launch_expand_select(int game, const View &v, hipStream_t s);      // expand of round r-1 fused with select of round r
void launch_leaf_cells(int game, const View &v, int8_t *dst, hipStream_t s);
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
// evaluates boards leaf_mine/theirs[leaf_slot[j]] for j < *n_leaf (<= max_leaves) into out_p/out_v
void resnet_forward(ResNetDev *, int game, const u64 *mine, const u64 *theirs, const int32_t *leaf_slot,
                    const int32_t *n_leaf, int max_leaves, float *out_p, float *out_v, hipStream_t s);
double resnet_flops_per_board(const ResNetDev *);
// diagnostics: arm the stamped instantiation for this thread's next fp32 launch; read back its clock / span
void resnet_trace_next();
void resnet_trace_result(double *ghz, double *span_us);

// split-fp16 variant (azsp_resnet_f16.hip)
struct ResNet16Dev;
ResNet16Dev *resnet16_create(int game, int blocks, int channels);
void resnet16_destroy(ResNet16Dev *);
int resnet16_load(ResNet16Dev *, const float *blob, size_t n_floats, hipStream_t s);
void resnet16_forward(ResNet16Dev *, const u64 *mine, const u64 *theirs, const int32_t *leaf_slot, const int32_t *n_leaf,
                      int max_leaves, float *out_p, float *out_v, hipStream_t s);

}  // namespace azsp
