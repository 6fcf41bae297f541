/* az_oracle.c -- CPU ORACLE in plain C (test infrastructure only; never linked into the product).
 *
 * Restates, on the reference's own cell-array board encoding (no bitboards, explicit to_play
 * labels -- deliberately a different formulation from the HIP engine so the two check each other):
 *   - Connect-4 / Connect2 rules      /root/reference/src/alphazero_simple/connect4_game.py:19-101
 *                                     /root/reference/src/alphazero_simple/connect2_game.py:18-80
 *   - ucb_score / select_child        src/alphazero_simple/monte_carlo_tree_search.py:11-22,63-78
 *   - Node.expand                     src/alphazero_simple/monte_carlo_tree_search.py:80-90
 *   - MCTS.run_batch                  src/alphazero_simple/monte_carlo_tree_search.py:111-171
 *   - MCTS.backpropagate              src/alphazero_simple/monte_carlo_tree_search.py:214-223
 *   - Node.select_action(T=1)         src/alphazero_simple/monte_carlo_tree_search.py:43-61
 *                                     (np.random.choice = cumsum / normalise / searchsorted-right,
 *                                      one MT19937 53-bit double per call)
 *   - EpisodeGenerator.generate_episodes  src/alphazero_less_simple/core/training/episode_generator.py:23-87
 * dtype semantics are those of the reference running under numpy 2.x:
 * value_sum float32, Q = W/n in float32, priors and PUCT score in float64.
 *
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the JSON fixtures under tests/golden/, which were
 * produced by running the reference itself (tests/golden/make_golden.py).
 *
 * Build: make -C oracle   ->  oracle/_build/libazoracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AMAX 81
#define CMAX 81

typedef int (*azo_predict_fn)(void *user, const int8_t *boards, int32_t n, float *p_out, float *v_out);
typedef double (*azo_uniform_fn)(void *user);

/* ------------------------------------------------------------------------------------------ */
/* games                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int id, rows, cols, k, A, cells; } game_t;

static int game_init(game_t *g, int id) {
    if (id == 0) { g->id = 0; g->rows = 6; g->cols = 7; g->k = 4; g->A = 7; g->cells = 42; return 0; }
    if (id == 1) { g->id = 1; g->rows = 1; g->cols = 4; g->k = 2; g->A = 4; g->cells = 4; return 0; }
    /* id 2: 9x9 Gomoku, five in a row -- NOT in the reference; a new BaseGame plugin
     * (alphazero-implementation_amd/azsp/games.py GomokuGame): action = row*9 + col, no legality check. */
    if (id == 2) { g->id = 2; g->rows = 9; g->cols = 9; g->k = 5; g->A = 81; g->cells = 81; return 0; }
    return -1;
}

/* get_next_state: connect4_game.py:28-42 (lowest empty row, no legality check);
 * connect2_game.py:28-36 (unconditional overwrite). */
static void next_state(const game_t *g, const int8_t *b, int player, int action, int8_t *out) {
    memcpy(out, b, (size_t)g->cells);
    if (g->id == 0) {
        for (int r = g->rows - 1; r >= 0; --r)
            if (out[r * g->cols + action] == 0) { out[r * g->cols + action] = (int8_t)player; break; }
    } else if (g->id == 1) {
        out[action] = (int8_t)player;
    } else {
        if (out[action] == 0) out[action] = (int8_t)player;
    }
}

static int line(const game_t *g, const int8_t *b, int r, int c, int dr, int dc, int player) {
    for (int i = 0; i < g->k; ++i)
        if (b[(r + i * dr) * g->cols + (c + i * dc)] != player) return 0;
    return 1;
}

/* is_win: connect4_game.py:58-88; connect2_game.py:54-66 */
static int is_win(const game_t *g, const int8_t *b, int player) {
    if (g->id == 1) {
        int run = 0;
        for (int c = 0; c < g->cols; ++c) {
            run = (b[c] == player) ? run + 1 : 0;
            if (run == g->k) return 1;
        }
        return 0;
    }
    for (int r = 0; r < g->rows; ++r)
        for (int c = 0; c + g->k <= g->cols; ++c)
            if (line(g, b, r, c, 0, 1, player)) return 1;
    for (int r = 0; r + g->k <= g->rows; ++r)
        for (int c = 0; c < g->cols; ++c)
            if (line(g, b, r, c, 1, 0, player)) return 1;
    for (int r = 0; r + g->k <= g->rows; ++r)
        for (int c = 0; c + g->k <= g->cols; ++c)
            if (line(g, b, r, c, 1, 1, player)) return 1;
    for (int r = g->k - 1; r < g->rows; ++r)
        for (int c = 0; c + g->k <= g->cols; ++c)
            if (line(g, b, r, c, -1, 1, player)) return 1;
    return 0;
}

static int has_moves(const game_t *g, const int8_t *b) {       /* connect4_game.py:44-46 / connect2 :38-42 */
    int n = (g->id == 0) ? g->cols : g->cells;                   /* connect-4 looks at the top row only */
    for (int c = 0; c < n; ++c) if (b[c] == 0) return 1;
    return 0;
}

static void valid_moves(const game_t *g, const int8_t *b, int *valid) {   /* :48-56 / :44-52 */
    for (int c = 0; c < g->A; ++c) valid[c] = (b[c] == 0);     /* connect-4: top row; connect2 / gomoku: every cell */
}

#define REW_NONE 2
static int reward(const game_t *g, const int8_t *b, int player) {  /* :90-98 / :68-78 */
    if (is_win(g, b, player)) return 1;
    if (is_win(g, b, -player)) return -1;
    if (has_moves(g, b)) return REW_NONE;
    return 0;
}

static void canonical(const game_t *g, const int8_t *b, int player, int8_t *out) {  /* :100-101 */
    for (int i = 0; i < g->cells; ++i) out[i] = (int8_t)(player * b[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* hash evaluator (definition: oracle/az_oracle.py hash_eval_bits)                            */
/* ------------------------------------------------------------------------------------------ */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void hash_eval(const game_t *g, const int8_t *b, uint64_t salt, float *p, float *v) {
    uint64_t mw[2] = {0, 0}, tw[2] = {0, 0};
    for (int r = 0; r < g->rows; ++r)
        for (int c = 0; c < g->cols; ++c) {
            /* connect-4 / connect2: column-major with a sentinel row; gomoku: row-major with a sentinel column */
            int bit = (g->id == 2) ? r * (g->cols + 1) + c : c * (g->rows + 1) + (g->rows - 1 - r);
            int8_t x = b[r * g->cols + c];
            if (x == 1) mw[bit >> 6] |= 1ull << (bit & 63); else if (x == -1) tw[bit >> 6] |= 1ull << (bit & 63);
        }
    /* boards wider than 64 bits: fold the words (identity for one word) */
    uint64_t mine = mw[0], theirs = tw[0];
    if (g->id == 2) { mine = mix64(mw[0]) ^ mw[1]; theirs = mix64(tw[0]) ^ tw[1]; }
    int valid[AMAX];
    valid_moves(g, b, valid);
    uint64_t h0 = mix64(mine ^ mix64(theirs) ^ salt);
    uint32_t w[AMAX], s = 0;
    for (int a = 0; a < g->A; ++a) {
        w[a] = 0;
        if (valid[a]) {
            uint64_t r = mix64(h0 + (uint64_t)a + 1ull) >> 52;
            w[a] = 1u + (uint32_t)((r * r * r) >> 24);
        }
        s += w[a];
    }
    float fs = (float)s;
    for (int a = 0; a < g->A; ++a) p[a] = (float)w[a] / fs;
    *v = (float)((int)(mix64(h0 + 0x1000ull) >> 48) - 32768) / 32768.0f;
}

typedef struct { const game_t *g; uint64_t salt; } hash_ctx;
static int hash_predict(void *user, const int8_t *boards, int32_t n, float *p, float *v) {
    hash_ctx *h = (hash_ctx *)user;
    for (int i = 0; i < n; ++i)
        hash_eval(h->g, boards + (size_t)i * h->g->cells, h->salt, p + (size_t)i * h->g->A, v + i);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* MT19937 exactly as numpy's legacy RandomState: seed(int) = init_genrand, random_sample =    */
/* genrand_res53 (numpy/random/src/mt19937/mt19937.c)                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void mt_seed(mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static uint32_t mt_next(mt_t *s) {
    if (s->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static double mt_uniform(void *user) {
    mt_t *s = (mt_t *)user;
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* ------------------------------------------------------------------------------------------ */
/* tree                                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n;            /* visit_count   mcts:27 */
    float   w;            /* value_sum (float32 under numpy 2.x; exact small ints before that) */
    double  prior;        /* prior         mcts:29 (float64 element of the renormalised array) */
    int32_t parent;       /* -1 for the root */
    int32_t state;        /* index into the tree's state pool, -1 = None */
    int32_t kid_base;     /* children: tree->kids[kid_base + action] = node or -1 where prob == 0 (mcts:87); -1 = none */
    int8_t  to_play;
    int8_t  expanded;
} node_t;

typedef struct {
    node_t *nodes; int n_nodes, cap_nodes;
    int8_t *states; int n_states;
    int32_t *kids; int n_kids;          /* pool: A slots per expanded node */
} tree_t;

static int kid_of(const tree_t *t, const node_t *x, int a) { return x->kid_base < 0 ? -1 : t->kids[x->kid_base + a]; }

static int node_new(tree_t *t, double prior, int to_play, int parent) {
    node_t *x = &t->nodes[t->n_nodes];
    x->n = 0; x->w = 0.0f; x->prior = prior; x->parent = parent; x->state = -1;
    x->kid_base = -1;
    x->to_play = (int8_t)to_play; x->expanded = 0;
    return t->n_nodes++;
}

/* action_probs * valid_moves; /= np.sum  (mcts:119-121,165-167): float32 * int -> float64,
 * np.sum over < 8 float64 elements is a plain left-to-right loop. */
/* np.sum of a contiguous float64 array (numpy pairwise_sum): n < 8 plain loop; n <= 128 eight running sums
 * over strides of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail one by one. */
static double np_sum(const double *a, int n) {
    if (n < 8) { double s = 0.0; for (int i = 0; i < n; ++i) s += a[i]; return s; }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s += a[i];
    return s;
}

static void masked_priors(const game_t *g, const float *p32, const int8_t *board, double *pr) {
    int valid[AMAX];
    valid_moves(g, board, valid);
    for (int a = 0; a < g->A; ++a) pr[a] = (double)p32[a] * (double)valid[a];
    const double s = np_sum(pr, g->A);
    for (int a = 0; a < g->A; ++a) pr[a] = pr[a] / s;
}

static void expand(const game_t *g, tree_t *t, int node, const int8_t *state, int to_play, const double *pr) {
    node_t *x = &t->nodes[node];                    /* mcts:80-90 */
    x->to_play = (int8_t)to_play;
    x->state = t->n_states++;
    memcpy(t->states + (size_t)x->state * CMAX, state, (size_t)g->cells);
    x->kid_base = t->n_kids;
    t->n_kids += g->A;
    for (int a = 0; a < g->A; ++a) {
        t->kids[t->nodes[node].kid_base + a] = -1;
        if (pr[a] != 0) { int c = node_new(t, pr[a], -to_play, node); t->kids[t->nodes[node].kid_base + a] = c; t->nodes[node].expanded = 1; }
    }
}

static double ucb(const node_t *parent, const node_t *child) {      /* mcts:11-22 */
    double prior_score = child->prior * sqrt((double)parent->n) / (double)(child->n + 1);
    if (child->n > 0) {
        float q = child->w / (float)child->n;                        /* float32 divide */
        return (double)(-q) + prior_score;
    }
    return 0.0 + prior_score;
}

static void backprop(tree_t *t, int leaf, float value, int to_play, int64_t *path_nodes) {  /* mcts:214-223 */
    for (int x = leaf; x >= 0; x = t->nodes[x].parent) {
        node_t *nd = &t->nodes[x];
        nd->w = nd->w + ((nd->to_play == to_play) ? value : -value);
        nd->n += 1;
        if (path_nodes) ++*path_nodes;
    }
}

typedef struct {
    int64_t evals, terminal_leaves, rounds, path_nodes, select_steps, max_nodes, plies;
} azo_stats;

/* MCTS.run_batch, mcts:111-171.  boards: canonical int8[n][cells]. trees must be pre-allocated. */
static int run_batch(const game_t *g, int sims, const int8_t *boards, const int *to_plays, int n, tree_t *trees,
                     azo_predict_fn fn, void *user, azo_stats *st,
                     int8_t *leaf_boards, int *leaf_tree, int *leaf_node, float *p, float *v) {
    const int cells = g->cells, A = g->A;
    double pr[AMAX];
    if (fn(user, boards, n, p, v)) return -4;                        /* :114 (values discarded) */
    st->evals += n;
    for (int i = 0; i < n; ++i) {
        tree_t *t = &trees[i];
        t->n_nodes = 0; t->n_states = 0; t->n_kids = 0;
        int root = node_new(t, 0.0, to_plays[i], -1);
        masked_priors(g, p + (size_t)i * A, boards + (size_t)i * cells, pr);
        expand(g, t, root, boards + (size_t)i * cells, to_plays[i], pr);       /* :116-123 */
    }
    int8_t tmp[CMAX], nxt[CMAX];
    for (int s = 0; s < sims; ++s) {                                 /* :125 */
        int m = 0;
        for (int i = 0; i < n; ++i) {                                /* :129 */
            tree_t *t = &trees[i];
            int node = 0, action = -1;
            while (t->nodes[node].expanded) {                        /* :130-133, select_child :63-78 */
                double best = -INFINITY; int best_a = -1, best_c = -1;
                const node_t *par = &t->nodes[node];
                for (int a = 0; a < A; ++a) {
                    int c = kid_of(t, par, a);
                    if (c < 0) continue;
                    double sc = ucb(par, &t->nodes[c]);
                    if (sc > best) { best = sc; best_a = a; best_c = c; }
                }
                action = best_a; node = best_c;
                st->select_steps++;
            }
            int parent = t->nodes[node].parent;
            const int8_t *pstate = t->states + (size_t)t->nodes[parent].state * CMAX;
            next_state(g, pstate, 1, action, tmp);                   /* :140 */
            canonical(g, tmp, -1, nxt);                              /* :142 */
            int val = reward(g, nxt, 1);                             /* :145 */
            int leaf_to_play = -t->nodes[parent].to_play;
            if (val == REW_NONE) {                                   /* :146-149 */
                memcpy(leaf_boards + (size_t)m * cells, nxt, (size_t)cells);
                leaf_tree[m] = i; leaf_node[m] = node; ++m;
            } else {                                                 /* :150-152 */
                backprop(t, node, (float)val, leaf_to_play, &st->path_nodes);
                st->terminal_leaves++;
            }
        }
        if (m > 0) {                                                 /* :154-169 */
            if (fn(user, leaf_boards, m, p, v)) return -4;
            st->evals += m;
            for (int j = 0; j < m; ++j) {
                tree_t *t = &trees[leaf_tree[j]];
                int node = leaf_node[j];
                int leaf_to_play = -t->nodes[t->nodes[node].parent].to_play;
                const int8_t *b = leaf_boards + (size_t)j * cells;
                masked_priors(g, p + (size_t)j * A, b, pr);
                expand(g, t, node, b, leaf_to_play, pr);
                backprop(t, node, v[j], leaf_to_play, &st->path_nodes);
            }
        }
        st->rounds++;
    }
    for (int i = 0; i < n; ++i) if (trees[i].n_nodes > st->max_nodes) st->max_nodes = trees[i].n_nodes;
    return 0;
}

static tree_t *trees_alloc(const game_t *g, int n, int sims) {
    tree_t *t = (tree_t *)calloc((size_t)n, sizeof(tree_t));
    if (!t) return NULL;
    int cap = 1 + g->A * (sims + 1);
    for (int i = 0; i < n; ++i) {
        t[i].cap_nodes = cap;
        t[i].nodes = (node_t *)malloc((size_t)cap * sizeof(node_t));
        t[i].states = (int8_t *)malloc((size_t)(sims + 2) * CMAX);
        t[i].kids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(sims + 2) * g->A);
        if (!t[i].nodes || !t[i].states || !t[i].kids) return NULL;
    }
    return t;
}
static void trees_free(tree_t *t, int n) {
    if (!t) return;
    for (int i = 0; i < n; ++i) { free(t[i].nodes); free(t[i].states); free(t[i].kids); }
    free(t);
}

/* ------------------------------------------------------------------------------------------ */
/* exported: search                                                                           */
/* ------------------------------------------------------------------------------------------ */
int azo_run_batch(int game, int sims, const int8_t *boards, const int32_t *to_plays, int n,
                  azo_predict_fn fn, void *user, uint64_t salt,
                  int32_t *counts, float *child_w, double *prior, float *root_w, int32_t *root_n,
                  int32_t *n_nodes, int64_t *stats7) {
    game_t g;
    if (game_init(&g, game) || n <= 0 || sims < 0) return -1;
    hash_ctx hc = { &g, salt };
    if (!fn) { fn = hash_predict; user = &hc; }
    tree_t *trees = trees_alloc(&g, n, sims);
    int8_t *lb = (int8_t *)malloc((size_t)n * g.cells);
    int *lt = (int *)malloc(sizeof(int) * (size_t)n), *ln = (int *)malloc(sizeof(int) * (size_t)n);
    int *tp = (int *)malloc(sizeof(int) * (size_t)n);
    float *p = (float *)malloc(sizeof(float) * (size_t)n * g.A), *v = (float *)malloc(sizeof(float) * (size_t)n);
    if (!trees || !lb || !lt || !ln || !tp || !p || !v) return -5;
    for (int i = 0; i < n; ++i) tp[i] = to_plays ? to_plays[i] : 1;
    azo_stats st; memset(&st, 0, sizeof st);
    int rc = run_batch(&g, sims, boards, tp, n, trees, fn, user, &st, lb, lt, ln, p, v);
    if (!rc)
        for (int i = 0; i < n; ++i) {
            const node_t *root = &trees[i].nodes[0];
            for (int a = 0; a < g.A; ++a) {
                int c = kid_of(&trees[i], root, a);
                if (counts) counts[i * g.A + a] = c < 0 ? 0 : trees[i].nodes[c].n;
                if (child_w) child_w[i * g.A + a] = c < 0 ? 0.0f : trees[i].nodes[c].w;
                if (prior) prior[i * g.A + a] = c < 0 ? 0.0 : trees[i].nodes[c].prior;
            }
            if (root_w) root_w[i] = root->w;
            if (root_n) root_n[i] = root->n;
            if (n_nodes) n_nodes[i] = trees[i].n_nodes;
        }
    if (stats7) { stats7[0] = st.evals; stats7[1] = st.terminal_leaves; stats7[2] = st.rounds; stats7[3] = st.path_nodes;
                  stats7[4] = st.select_steps; stats7[5] = st.max_nodes; stats7[6] = 0; }
    trees_free(trees, n); free(lb); free(lt); free(ln); free(tp); free(p); free(v);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* exported: episode generation (episode_generator.py:23-87)                                  */
/* ------------------------------------------------------------------------------------------ */
/* Node.select_action(temperature=1), mcts:43-61 */
static int sample_action(const game_t *g, const tree_t *t, double u) {
    const node_t *root = &t->nodes[0];
    int acts[AMAX], m = 0;
    double d[AMAX];
    for (int a = 0; a < g->A; ++a)
        if (kid_of(t, root, a) >= 0) { acts[m] = a; d[m] = pow((double)t->nodes[kid_of(t, root, a)].n, 1.0); ++m; }
    double tot = 0.0;                                   /* builtin sum(): 0 + d0 + d1 + ... */
    for (int i = 0; i < m; ++i) tot = tot + d[i];
    for (int i = 0; i < m; ++i) d[i] = d[i] / tot;
    double cdf[AMAX], run = 0.0;                        /* np.random.choice: cumsum, /= last */
    for (int i = 0; i < m; ++i) { run += d[i]; cdf[i] = run; }
    double last = cdf[m - 1];
    int idx = 0;
    for (int i = 0; i < m; ++i) { cdf[i] /= last; if (cdf[i] <= u) idx = i + 1; }   /* searchsorted side='right' */
    return acts[idx];
}

/* Output capacity: ep arrays n_slots(+1); sample arrays sample_cap; trace arrays trace_cap rows
 * (one row per (ply, slot) visited, in order).  seed >= 0 selects the built-in MT19937
 * (== np.random.seed(seed)); otherwise `rng` is called once per draw.
 * Returns number of episodes (== n_slots) or < 0. */
int azo_generate(int game, int sims, int n_slots, azo_predict_fn fn, void *user, uint64_t salt,
                 int64_t seed, azo_uniform_fn rng, void *rng_user,
                 int64_t *ep_offset, int32_t *ep_slot, int32_t *ep_ply,
                 int8_t *states, double *policies, int32_t *values, int64_t sample_cap,
                 int32_t *trace_counts, int32_t *trace_actions, int64_t trace_cap, int64_t *trace_rows,
                 double *next_uniform_after, int64_t *stats7) {
    game_t g;
    if (game_init(&g, game) || n_slots <= 0) return -1;
    const int A = g.A, cells = g.cells, N = n_slots;
    hash_ctx hc = { &g, salt };
    if (!fn) { fn = hash_predict; user = &hc; }
    mt_t mt;
    if (seed >= 0) { mt_seed(&mt, (uint32_t)seed); rng = mt_uniform; rng_user = &mt; }
    if (!rng) return -1;

    int max_hist = (game == 0) ? 42 : (game == 1 ? 4 : 81);
    tree_t *trees = trees_alloc(&g, N, sims);
    int8_t *cur = (int8_t *)calloc((size_t)N, (size_t)cells);            /* states       :25 */
    int *player = (int *)malloc(sizeof(int) * (size_t)N);                /* current_players :26 */
    int8_t *canon = (int8_t *)malloc((size_t)N * cells);
    int8_t *h_state = (int8_t *)malloc((size_t)N * max_hist * cells);    /* train_examples_list :27 */
    double *h_pi = (double *)malloc(sizeof(double) * (size_t)N * max_hist * A);
    int *h_player = (int *)malloc(sizeof(int) * (size_t)N * max_hist);
    int *h_len = (int *)calloc((size_t)N, sizeof(int));
    int8_t *lb = (int8_t *)malloc((size_t)N * cells);
    int *lt = (int *)malloc(sizeof(int) * (size_t)N), *ln = (int *)malloc(sizeof(int) * (size_t)N);
    float *p = (float *)malloc(sizeof(float) * (size_t)N * A), *v = (float *)malloc(sizeof(float) * (size_t)N);
    if (!trees || !cur || !player || !canon || !h_state || !h_pi || !h_player || !h_len || !lb || !lt || !ln || !p || !v) return -5;
    for (int i = 0; i < N; ++i) player[i] = 1;
    azo_stats st; memset(&st, 0, sizeof st);
    int64_t n_ep = 0, n_samp = 0, rows = 0;
    int rc = 0, ply = 0, stop = 0;
    int8_t tmp[CMAX];
    if (ep_offset) ep_offset[0] = 0;
    while (!stop) {                                                       /* :30 */
        for (int i = 0; i < N; ++i) canonical(&g, cur + (size_t)i * cells, player[i], canon + (size_t)i * cells);  /* :31-34 */
        rc = run_batch(&g, sims, canon, player, N, trees, fn, user, &st, lb, lt, ln, p, v);   /* :36 */
        if (rc) break;
        st.plies++;
        for (int i = 0; i < N && !stop; ++i) {                            /* :38-44 */
            const tree_t *t = &trees[i];
            int cnt[AMAX]; long sum = 0;
            for (int a = 0; a < A; ++a) { int c = kid_of(t, &t->nodes[0], a); cnt[a] = c < 0 ? 0 : t->nodes[c].n; sum += cnt[a]; }
            int k = h_len[i];
            if (k >= max_hist) { rc = -3; stop = 1; break; }
            memcpy(h_state + ((size_t)i * max_hist + k) * cells, canon + (size_t)i * cells, (size_t)cells);
            for (int a = 0; a < A; ++a) h_pi[((size_t)i * max_hist + k) * A + a] = (double)cnt[a] / (double)sum;   /* :45-48 */
            h_player[(size_t)i * max_hist + k] = player[i];
            h_len[i] = k + 1;                                             /* :50-52 */
            int act = sample_action(&g, t, rng(rng_user));                /* :54 */
            if (trace_counts && rows < trace_cap) {
                for (int a = 0; a < A; ++a) trace_counts[rows * A + a] = cnt[a];
                trace_actions[rows] = act;
            }
            ++rows;
            next_state(&g, cur + (size_t)i * cells, player[i], act, tmp); /* :55-58 */
            memcpy(cur + (size_t)i * cells, tmp, (size_t)cells);
            player[i] = -player[i];
            int rew = reward(&g, cur + (size_t)i * cells, player[i]);     /* :59 */
            if (rew != REW_NONE) {                                        /* :61-78 */
                if (n_samp + h_len[i] > sample_cap) { rc = -5; stop = 1; break; }
                for (int q = 0; q < h_len[i]; ++q) {
                    memcpy(states + (size_t)(n_samp + q) * cells, h_state + ((size_t)i * max_hist + q) * cells, (size_t)cells);
                    memcpy(policies + (size_t)(n_samp + q) * A, h_pi + ((size_t)i * max_hist + q) * A, sizeof(double) * (size_t)A);
                    values[n_samp + q] = (h_player[(size_t)i * max_hist + q] != player[i]) ? -rew : rew;   /* :73-74 */
                }
                n_samp += h_len[i];
                if (ep_slot) ep_slot[n_ep] = i;
                if (ep_ply) ep_ply[n_ep] = ply;
                ++n_ep;
                if (ep_offset) ep_offset[n_ep] = n_samp;
                memset(cur + (size_t)i * cells, 0, (size_t)cells);        /* :80-82 */
                player[i] = 1; h_len[i] = 0;
                if (n_ep >= N) stop = 1;                                  /* :84-87 */
            }
        }
        ++ply;
    }
    if (trace_rows) *trace_rows = rows;
    if (next_uniform_after) *next_uniform_after = rng(rng_user);
    if (stats7) { stats7[0] = st.evals; stats7[1] = st.terminal_leaves; stats7[2] = st.rounds; stats7[3] = st.path_nodes;
                  stats7[4] = st.select_steps; stats7[5] = st.max_nodes; stats7[6] = st.plies; }
    trees_free(trees, N); free(cur); free(player); free(canon); free(h_state); free(h_pi); free(h_player); free(h_len);
    free(lb); free(lt); free(ln); free(p); free(v);
    return rc ? rc : (int)n_ep;
}

/* numpy-compatible MT19937 stream, for tests: fills out[n] after np.random.seed(seed) */
void azo_mt_uniforms(uint32_t seed, double *out, int n) {
    mt_t mt; mt_seed(&mt, seed);
    for (int i = 0; i < n; ++i) out[i] = mt_uniform(&mt);
}

void azo_hash_eval(int game, const int8_t *board, uint64_t salt, float *p, float *v) {
    game_t g; if (game_init(&g, game)) return;
    hash_eval(&g, board, salt, p, v);
}
