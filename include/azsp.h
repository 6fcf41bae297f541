/* azsp.h -- C ABI of libazsp.so, the MI355X-native self-play episode generator.
 *
 * The reference (pierreveron/alphazero-implementation) is pure Python and has no FFI; this header
 * is the boundary a ctypes binding sits on (INTEGRATION.md shows the stub).  Every entry point
 * names the reference interface it stands behind (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes in; nothing is retained after
 *     the call returns except the evaluator callback and its user pointer.
 *   - return value: 0 = ok, < 0 = error code; azsp_last_error() gives the message of the last
 *     failing call on the calling thread.  Nothing throws across the ABI.
 *   - one engine = one HIP device + one private non-blocking HIP stream.  Every entry point switches the
 *     calling thread to the engine's device and restores the thread's previous current device before it
 *     returns.  Device pointers handed to the azsp_export_*_device calls must live on the engine's device;
 *     the engine's stream does not wait for work the caller has pending on other streams (synchronise those
 *     first), and the call returns after its own stream has drained.  Calls on one engine must
 *     be serialised by the caller; different engines may be used from different threads (the
 *     reference runs generation on a daemon thread next to training:
 *     src/alphazero_less_simple/core/training/datamodule.py:19-47,111-117).
 *   - boards cross the ABI in the reference's own encoding, narrowed to int8: row-major cells,
 *     row 0 = top, +1 = side to move ("canonical", src/alphazero_simple/connect4_game.py:100-101),
 *     -1 = opponent, 0 = empty.
 */
#ifndef AZSP_H
#define AZSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZSP_ABI_VERSION 3

typedef struct azsp_engine azsp_engine;

/* Game plugin ids: device-side traits standing in for BaseGame subclasses
 * (src/alphazero_simple/base_game.py:6-62). */
enum { AZSP_GAME_CONNECT4 = 0,   /* src/alphazero_simple/connect4_game.py:6-101 */
       AZSP_GAME_CONNECT2 = 1,   /* src/alphazero_simple/connect2_game.py:6-80  */
       AZSP_GAME_GOMOKU9  = 2 }; /* 9x9, five in a row: NOT in the reference -- device side of the new BaseGame plugin
                                    alphazero-implementation_amd/azsp/games.py GomokuGame (BASELINE.json configs[4]) */

/* Search rule sets: the reference holds two generations of the same batched search; the arithmetic differs.
 *   AZSP_RULES_SIMPLE  src/alphazero_simple/monte_carlo_tree_search.py:11-223 (what alphazero_less_simple's
 *                      EpisodeGenerator calls): W float32, score = -(W/n) + P*sqrt(N)/(n+1), priors masked and
 *                      renormalised, children where prob != 0, root value discarded, n_sims descents per ply.
 *   AZSP_RULES_GEN1    src/alphazero_implementation/core/search/mcts/search.py:10-91 + node.py:7-73: W float64,
 *                      score = W/n + c*P*sqrt(N)/(1+n) (no negation), priors taken as the model gives them, a child
 *                      for every legal action, the root's expansion is simulation 1 (so pi = n/(N-1)), no sign
 *                      flip above a terminal node.  Connect-4 only (the package hard-codes simulator.game.connect).
 *   AZSP_RULES_SIMPLE_NUMPY1  AZSP_RULES_SIMPLE as the same source runs under numpy 1.x (requirements.txt does not
 *                      pin numpy): `value_sum += np.float32` then promotes to float64, so W and -(W/n) are float64;
 *                      everything else is unchanged.  numpy 1.x is not installable in the build environment: this
 *                      variant is checked against the oracle's restatement only ("parity unpinned"). */
enum { AZSP_RULES_SIMPLE = 0, AZSP_RULES_GEN1 = 1, AZSP_RULES_SIMPLE_NUMPY1 = 2 };

/* Evaluator kinds: what stands behind BaseModel.predict (src/alphazero_simple/base_model.py:52-58). */
enum { AZSP_EVAL_RESNET = 0,  /* on-device fp32 MFMA forward of ResNet (src/alphazero_simple/resnet.py:30-157) */
       AZSP_EVAL_HASH   = 1,  /* deterministic integer-hash (P,V); parity-test evaluator            */
       AZSP_EVAL_HOST   = 2 };/* host callback: any BaseModel.predict; leaves are handed over in slot order */

enum { AZSP_OK = 0, AZSP_ERR_ARG = -1, AZSP_ERR_HIP = -2, AZSP_ERR_STATE = -3, AZSP_ERR_CALLBACK = -4,
       AZSP_ERR_NOMEM = -5 };

#define AZSP_FLAG_TIMING 1u      /* record HIP events around every kernel class (azsp_get_stats) */
#define AZSP_FLAG_F16_SPLIT 2u   /* device ResNet on the fp16 matrix pipe with every fp32 operand split into
                                    two fp16 numbers (hi*hi + hi*lo + lo*hi, fp32 accumulate): fp32-class
                                    accuracy (<= 1e-5 on p, v), 3 MFMAs at 16x the fp32-MFMA rate.  Connect-4
                                    and Gomoku 9x9, up to 256 channels (azsp_create fails otherwise).  Without
                                    the flag the network runs on v_mfma_f32_32x32x2_f32 (exact fp32).     */

#define AZSP_FLAG_TREE_REUSE 4u  /* NOT in the reference (it rebuilds every tree every ply, monte_carlo_tree_search.py:116-123;
                                    listed as an option in SURVEY 8f): after a move the chosen child's subtree, with
                                    its statistics, is the next ply's tree and n_sims further simulations are added
                                    to it.  Trees then live for a whole game: node capacity per slot grows from
                                    A*(n_sims+1) to A*(n_sims+1)*max_plies (HBM is sized for it).  Root noise is only
                                    mixed in where a root is expanded afresh.                                   */

#define AZSP_FLAG_EVAL_CACHE 8u  /* NOT in the reference, and invisible in its results: every position is evaluated at most
                                    ONCE per generation run.  The reference evaluates every leaf of every game in every
                                    round (monte_carlo_tree_search.py:154-157) and rebuilds every tree every ply (:116-123)
                                    with no noise, so equal positions -- the 4096 identical first plies, transpositions,
                                    the part of last ply's tree below the move just played -- are evaluated over and over
                                    with identical results.  The device evaluators are deterministic and bitwise independent
                                    of the batch a position sits in, so a position -> (p, v) table in HBM (filled by the
                                    inference kernel, probed by the tree kernels) changes NO bit of any visit count, move or
                                    episode (tested on against off); azsp_stats then reports the reference's evaluation
                                    count (evals) and the executed one (evals_executed) side by side.  Cleared by azsp_reset,
                                    azsp_load_resnet and azsp_set_evaluator.  Device evaluators only (AZSP_EVAL_HOST runs
                                    without it); games with an injective 64-bit position key (Connect-4, Connect2).  */

/* Diagnostics switches: the environment variable AZSP_DIAG, read ONCE per engine by azsp_create, may hold
 *   no_fused_rounds  runs of up to 32 games use the two-launch simulation rounds of larger runs instead of the one-launch
 *                    form (same results; the tests compare the two)
 *   unfused_tree     device evaluators use the four-launch slot-ordered rounds of the host evaluator (same results)
 * Nothing else in the product library reads the environment except tuning knobs that choose among kernel variants which
 * return the same bits (AZSP_RESNET_VARIANT, AZSP_RESNET_GATES, AZSP_F16_WR, AZSP_F16_SMALL_WR, AZSP_F16_GATES) and
 * AZSP_RESNET_TRACE / AZSP_PROBE_VERBOSE (phase-time printouts). */

typedef struct azsp_config {
    uint32_t struct_size;    /* sizeof(azsp_config), for forward compatibility                       */
    int32_t  game;           /* AZSP_GAME_*                                                           */
    int32_t  n_slots;        /* concurrent games = AlphaZeroConfig.num_episodes (config.py:11)       */
    int32_t  n_sims;         /* AlphaZeroConfig.num_simulations (config.py:8-10)                      */
    int32_t  device;         /* HIP device ordinal                                                    */
    int32_t  res_blocks;     /* ResNet num_res_blocks (resnet.py:35), 0 if no device net              */
    int32_t  channels;       /* ResNet num_channels  (resnet.py:36)                                   */
    uint32_t flags;          /* AZSP_FLAG_*                                                           */
    uint64_t hash_salt;      /* AZSP_EVAL_HASH only                                                   */
    int32_t  n_groups;       /* slot groups searched concurrently on separate HIP streams (1..8);
                                0 = automatic.  Results do not depend on it.                          */
    int32_t  arena_episodes; /* episode capacity of the output arena; 0 = n_slots.  The sharded exact mode
                                (azsp_play_begin/commit) lets one shard emit more than its own n_slots.    */
    int32_t  search_rules;   /* AZSP_RULES_*                                                           */
    int32_t  eval_cache_entries; /* AZSP_FLAG_EVAL_CACHE: table entries (64 B each, rounded up to a power of two,
                                    at most 2^26); 0 = automatic (64 * n_slots * (n_sims + 1))              */
    int32_t  timing_sample;  /* AZSP_FLAG_TIMING: n > 1 = time the kernels of 32 consecutive simulation rounds out of
                                every 32 n only (timestamps on a dispatch cost a few microseconds each, which
                                matters where a round is launch latency); 0 / 1 = every launch            */
    int32_t  reserved;       /* 0                                                                      */
} azsp_config;

/* Host evaluator: boards int8[n][cells] (canonical) -> p float32[n][n_actions], v float32[n].
 * Return 0 on success.  Called on the thread that called into the engine. */
typedef int (*azsp_predict_fn)(void *user, const int8_t *boards, int32_t n, float *p_out, float *v_out);

/* Result of one ply of lock-step play (azsp_play). */
typedef struct azsp_ply_result {
    int32_t ply;               /* index of the ply just played                                       */
    int32_t n_finished;        /* episodes completed (and counted) in this ply                        */
    int64_t episodes_total;    /* episodes emitted so far by this generation run                      */
    int64_t samples_total;     /* samples emitted so far                                              */
    int32_t uniforms_used;     /* how many of the n_slots uniforms the reference would have drawn:
                                  n_slots, or (last counted slot + 1) in the stopping ply
                                  (episode_generator.py:84-87 returns inside the slot loop)         */
    int32_t finished;          /* 1 once episodes_total >= n_slots (always 0 when the caller passed a cap)     */
    int32_t last_counted_slot; /* slot of the completion that exhausted the cap / the stop rule, else -1   */
    int32_t reserved;
} azsp_ply_result;

typedef struct azsp_stats {
    uint64_t plies, rounds;            /* azsp_search calls, simulation rounds                       */
    uint64_t evals;                    /* positions that needed the evaluator (roots + leaves): what the reference
                                          evaluates, whether or not AZSP_FLAG_EVAL_CACHE found some of them done */
    uint64_t terminal_leaves;          /* simulations that ended on a terminal leaf (no evaluation)  */
    uint64_t infer_launches;           /* launches of the dominant inference kernel                  */
    double   infer_ms;                 /* sum of their HIP-event durations (AZSP_FLAG_TIMING)        */
    uint64_t infer_boards;             /* boards those launches evaluated                            */
    double   select_ms, expand_ms, play_ms;   /* tree kernels, HIP-event sums                        */
    uint64_t select_launches, expand_launches;
    uint64_t path_nodes;               /* sum of backup path lengths (for algorithmic bytes)         */
    double   infer_union_ms;           /* time during which >= 1 inference launch was running (launches
                                          of different slot groups overlap)                            */
    uint64_t evals_executed;           /* positions actually evaluated (== evals without AZSP_FLAG_EVAL_CACHE; with it
                                          the count can vary by a few between identical runs: two slot groups that
                                          meet the same new position in the same search both evaluate it) */
    double   infer_exec_flops_per_board;   /* matrix-pipe FLOP the inference kernel variant of a full-size launch EXECUTES per board
                                          slot of a full workgroup: padding rows included, statically skipped
                                          off-board taps excluded, x 3 for the split-fp16 kernel (hi*hi, hi*lo, lo*hi) */
    uint64_t eval_cache_entries;       /* table size in use (0 = cache off)                              */
    uint64_t infer_launches_timed;     /* how many of infer_launches carried timing events (infer_ms / infer_union_ms cover these) */
    double   infer_exec_flops;         /* matrix-pipe FLOP executed by all inference launches, summed ON THE DEVICE: every
                                          launch adds (workgroups that had boards) x (FLOP one workgroup of that variant
                                          executes) -- exact per workgroup, whichever variant ran                      */
} azsp_stats;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int  azsp_abi_version(void);
int  azsp_create(const azsp_config *cfg, azsp_engine **out);
void azsp_destroy(azsp_engine *e);
const char *azsp_last_error(void);

/* geometry of the selected game: BaseGame.get_board_size / get_action_size (base_game.py:17-25) */
int  azsp_game_info(int32_t game, int32_t *rows, int32_t *cols, int32_t *n_actions, int32_t *max_plies);

/* The device game trait evaluated on the HOST (the trait code in csrc/azsp_games.h is shared by host and
 * device): for a canonical board, the legal-move bit mask (get_valid_moves, connect4_game.py:48-56;
 * int32[(A + 31) / 32] words, bit a of the mask = action a), the
 * reward for the side to move (get_reward_for_player(board, +1), :90-98; 2 = game not over) and the canonical
 * board after `action` by the side to move, seen by the opponent (get_next_state + get_canonical_board(-1),
 * :28-42,:100-101).  No GPU needed: this is what the CPU conformance test pins against the reference fixtures. */
int  azsp_game_rules(int32_t game, const int8_t *cells, int32_t action, int8_t *next_cells, int32_t *legal_mask,
                     int32_t *terminal);

/* ---- evaluator ------------------------------------------------------------------------------ */
/* Upload ResNet weights.  `blob` is the float32 concatenation, in this order, of the state_dict
 * tensors of src/alphazero_simple/resnet.py:44-72 (each tensor row-major as PyTorch stores it):
 *   input_conv.0.{weight,bias}, input_conv.1.{weight,bias,running_mean,running_var},
 *   for i in blocks: residual_blocks.i.{conv1.weight,conv1.bias,bn1.weight,bn1.bias,bn1.running_mean,
 *                    bn1.running_var,conv2.weight,conv2.bias,bn2.weight,bn2.bias,bn2.running_mean,bn2.running_var},
 *   policy_head.0.{weight,bias}, policy_head.1.{weight,bias,running_mean,running_var}, policy_head.4.{weight,bias},
 *   value_head.0.{weight,bias},  value_head.1.{weight,bias,running_mean,running_var},  value_head.4.{weight,bias}
 * Eval-mode BatchNorm (eps 1e-5) is folded into the preceding convolution at upload. */
int  azsp_load_resnet(azsp_engine *e, const float *blob, size_t n_floats);
size_t azsp_resnet_blob_floats(int32_t game, int32_t res_blocks, int32_t channels);
int  azsp_set_evaluator(azsp_engine *e, int32_t kind, azsp_predict_fn fn, void *user);

/* BaseModel.predict (resnet.py:130-157) with the engine's current evaluator.  Uses the slots as staging: like
 * azsp_run_batch it overwrites the live games of a generation run in progress (call azsp_reset before the next
 * azsp_search); episodes already emitted stay fetchable. */
int  azsp_predict(azsp_engine *e, const int8_t *boards, int32_t n, float *p_out, float *v_out);

/* ---- search facade: MCTS.run_batch (src/alphazero_simple/monte_carlo_tree_search.py:111-171) --- */
/* Searches n (<= n_slots) caller-supplied canonical boards.  Outputs (any may be NULL):
 *   counts  int32[n][A]   root.children[a].visit_count (0 where no child)
 *   child_w float [n][A]  root.children[a].value_sum
 *   prior   double[n][A]  root.children[a].prior (0 where no child)
 *   root_w  float [n]     root.value_sum          root_n int32[n]  root.visit_count */
int  azsp_run_batch(azsp_engine *e, const int8_t *boards, int32_t n, int32_t *counts, float *child_w,
                    double *prior, float *root_w, int32_t *root_n);
/* value_sum of the first n trees of the search just run, as float64 (exact in both rule sets; with
 * AZSP_RULES_GEN1 azsp_run_batch's float outputs are rounded): child_w double[n][A], root_w double[n],
 * root_n int32[n] -- Node.value = value_sum / visit_count (node.py:50-55, AlphaZeroSearch.run search.py:60-64) */
int  azsp_root_values(azsp_engine *e, int32_t n, double *child_w, double *root_w, int32_t *root_n);

/* ---- episode generation: EpisodeGenerator.generate_episodes
 *      (src/alphazero_less_simple/core/training/episode_generator.py:23-87) --------------------- */
int  azsp_reset(azsp_engine *e);                       /* :25-29  fresh games in every slot; the episode arena is
                                                          empty again (fetch/export of earlier episodes fail)      */
/* The position every game starts from (canonical cells; NULL = empty board, the default).  Stands behind
 * generate_episodes(initial_state) of src/alphazero_implementation/core/training/episode_generator.py:35-45.
 * Takes effect at the next azsp_reset and at every slot restart. */
int  azsp_set_initial_board(azsp_engine *e, const int8_t *cells);
int  azsp_search(azsp_engine *e);                      /* :31-36  run_batch on the slots' boards     */
int  azsp_root_counts(azsp_engine *e, int32_t *counts);/* int32[n_slots][A] of the search just run   */
/* Search options the reference does NOT have (its generator fixes temperature = 1 and uses no noise,
 * episode_generator.py:54; they are listed in its src/deepmind_alphazero_pseudocode.py:233-290).  All off by
 * default, so the default behaviour is the reference's.
 *   sampling_plies >= 0: after that many plies of a game the move is arg-max of the visit counts
 *                        (Node.select_action(temperature=0), monte_carlo_tree_search.py:49-50); < 0 = always sample
 *   noise_eps > 0      : azsp_search_with_noise mixes caller-drawn noise into the root priors,
 *                        prior*(1-eps) + noise*eps on existing children (noise double[n_slots][A])
 *   c_init, c_base     : exploration factor c(N) multiplying P*sqrt(N)/(n+1): c_base <= 0 -> the constant c_init
 *                        (1.0 = the reference's implicit c_puct, monte_carlo_tree_search.py:14-17; the
 *                        `exploration_weight` of search.py:16,35 under AZSP_RULES_GEN1); c_base > 0 ->
 *                        log((N + c_base + 1) / c_base) + c_init, the pb_c of deepmind_alphazero_pseudocode.py:252-255 */
typedef struct azsp_search_options { uint32_t struct_size; int32_t sampling_plies; double noise_eps;
                                     double c_init, c_base; } azsp_search_options;
int  azsp_set_search_options(azsp_engine *e, const azsp_search_options *opts);
int  azsp_search_with_noise(azsp_engine *e, const double *noise);
/* :38-87  pi, sample (one uniform per slot, slot order), apply, terminal test, emit, restart.      */
int  azsp_play(azsp_engine *e, const double *uniforms, azsp_ply_result *out);
/* The same in two phases, for running ONE reference generator sharded over several engines/GPUs with the
 * reference's exact global stop rule (episode_generator.py:84-87; SURVEY.md 8e "mode A"): begin = sample and
 * apply the moves and report how many games of this shard ended; the caller combines the counts of all shards
 * in slot order and tells each shard how many of its completions are still within the first N (allow);
 * commit = emit exactly those (slot order) and restart the finished slots.  allow < 0 = azsp_play's rule. */
int  azsp_play_begin(azsp_engine *e, const double *uniforms, int32_t *n_done);
int  azsp_play_commit(azsp_engine *e, int64_t allow, azsp_ply_result *out);
/* Copy emitted episodes [first, first+n) to host (any pointer may be NULL):
 *   ep_offset int64[n+1]  sample range of each episode, relative to the first copied sample
 *   ep_slot int32[n], ep_ply int32[n]   where/when the episode finished
 *   states  int8 [S][cells]  canonical boards   (Sample.state,  episode.py:11)
 *   policies double[S][A]                       (Sample.policy, episode.py:12)
 *   values  int32[S]                            (Sample.value,  episode.py:13)
 * with S = total samples of those episodes; sample_cap is the capacity of the three sample arrays. */
int  azsp_fetch_episodes(azsp_engine *e, int64_t first, int64_t n, int64_t *ep_offset, int32_t *ep_slot,
                         int32_t *ep_ply, int8_t *states, double *policies, int32_t *values,
                         int64_t sample_cap);
/* Device-side export for the multi-GPU all-gather: packs the samples of episodes [first, first+n)
 * into caller-owned DEVICE memory as records of azsp_record_bytes() bytes:
 *   int8 state[cells] | int8 value | pad to 8 | double policy[A].  Returns the number of records. */
/* The move played from each sample position of episodes [first, first+n): int32[S].  (The gen-1 samples hold
 * simulator State objects, episode.py:12-19: the adapter rebuilds them by replaying these moves.) */
int  azsp_fetch_moves(azsp_engine *e, int64_t first, int64_t n, int32_t *moves, int64_t sample_cap);
int64_t azsp_export_records_device(azsp_engine *e, int64_t first, int64_t n, void *dev_dst, int64_t cap_records);
int32_t azsp_record_bytes(int32_t game);
/* Training tensors of episodes [first, first+n) written into caller-owned DEVICE memory, in the layout
 * BaseModel.format_dataset produces (src/alphazero_simple/base_model.py:61-67 with
 * ResNet._states_to_tensor, resnet.py:115-128; called at datamodule.py:134-142):
 *   planes float32[S][3][cells] = [empty, ==+1, ==-1], policies float32[S][A], values float32[S].
 * Returns S (number of samples) or < 0.  Replaces device->host->Python lists->np.stack->device. */
int64_t azsp_export_dataset_device(azsp_engine *e, int64_t first, int64_t n, float *planes, float *policies,
                                   float *values, int64_t cap_samples);

/* Measurement aid: evaluates n (<= n_slots) boards once with the diagnostic instantiation of the engine's inference
 * kernel (fp32, or f16x2 with AZSP_FLAG_F16_SPLIT) that stamps s_memtime / s_memrealtime around the residual tower, and
 * returns the clock the chip held inside the kernel (GHz) and the launch's span (us).  The roofline peaks assume the
 * nominal 2.4 GHz; what the chip really holds under the kernel is a measurement (2.35 GHz under
 * the fp32 kernel, 1.75-1.8 GHz under the fp16 MFMAs).  Call it after >= 100 launches in the same process: the first
 * launches of a process run while the clock is still ramping up.  64-channel Connect-4 only. */
int  azsp_probe_clock(azsp_engine *e, const int8_t *boards, int32_t n, double *clock_ghz, double *span_us);
int  azsp_get_stats(azsp_engine *e, azsp_stats *out);
int  azsp_reset_stats(azsp_engine *e);
int  azsp_sync(azsp_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* AZSP_H */
