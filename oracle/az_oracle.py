"""CPU ORACLE (test infrastructure only -- never imported by the product path).

A plain-Python/numpy restatement of the reference's self-play hot path, written so that
every arithmetic step keeps the dtype the reference has *as it runs under numpy 2.x*
(SURVEY.md section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

What is restated (paths relative to /root/reference/):
  * Connect-4 / Connect2 rules     src/alphazero_simple/connect4_game.py:19-101,
                                   src/alphazero_simple/connect2_game.py:18-80
  * PUCT score, tree node          src/alphazero_simple/monte_carlo_tree_search.py:11-99
  * batched search ``run_batch``   src/alphazero_simple/monte_carlo_tree_search.py:111-171
  * backup                         src/alphazero_simple/monte_carlo_tree_search.py:214-223
  * move sampling (T=1/0/inf)      src/alphazero_simple/monte_carlo_tree_search.py:43-61
  * lock-step episode generation   src/alphazero_less_simple/core/training/episode_generator.py:23-87

Pinning: ``tests/golden/*.json`` were produced by *running the reference itself* in the build
container (tests/golden/make_golden.py); tests/test_oracle_golden.py checks this module against
every one of them bit-for-bit, and against the reference's own known-answer NodeTests
(src/alphazero_simple/tests.py:90-241).

The module also defines the *hash evaluator*: a deterministic (P, V) "network" computed from an
integer hash of the board.  It exists because torch-CPU inference is not bit-reproducible across
batch shapes/hosts; with it, visit counts / moves / outcomes can be
compared bit-exactly between the reference, this oracle, the C oracle and the HIP engine.
"""
from __future__ import annotations

import math
from typing import Callable, Sequence

import numpy as np

M64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------------
# Game rules on the reference's board encoding (np.int64, row 0 = top, +1 / -1 / 0)
# --------------------------------------------------------------------------------------------
class C4Rules:
    """Connect-4 6x7, four in a row.  connect4_game.py:13-101."""

    name = "connect4"
    rows, cols, k = 6, 7, 4
    action_size = 7

    def __init__(self):
        r, c, k = self.rows, self.cols, self.k
        win = []
        for y in range(r):                       # horizontal  connect4_game.py:61-65
            for x in range(c - k + 1):
                win.append([(y, x + i) for i in range(k)])
        for y in range(r - k + 1):               # vertical    :68-71
            for x in range(c):
                win.append([(y + i, x) for i in range(k)])
        for y in range(r - k + 1):               # diagonal \  :74-80
            for x in range(c - k + 1):
                win.append([(y + i, x + i) for i in range(k)])
        for y in range(k - 1, r):                # diagonal /  :83-88
            for x in range(c - k + 1):
                win.append([(y - i, x + i) for i in range(k)])
        self._win_flat = np.array([[yy * c + xx for yy, xx in w] for w in win], dtype=np.int64)

    def init_board(self) -> np.ndarray:          # :19-20
        return np.zeros((self.rows, self.cols), dtype=int)

    def next_state(self, board, player, action):  # :28-42  (no legality check: full column = no-op)
        b = np.copy(board)
        col = b[:, action]
        empties = np.nonzero(col == 0)[0]
        if empties.size:
            b[empties[-1], action] = player      # lowest empty cell (largest row index)
        return b, -player

    def has_moves(self, board) -> bool:          # :44-46
        return bool(np.any(board[0] == 0))

    def valid_moves(self, board) -> list[int]:   # :48-56
        return [1 if board[0][x] == 0 else 0 for x in range(self.cols)]

    def is_win(self, board, player) -> bool:     # :58-88
        cells = np.asarray(board).reshape(-1)[self._win_flat]
        return bool(np.any(np.all(cells == player, axis=1)))

    def reward(self, board, player):             # :90-98  win is tested before draw
        if self.is_win(board, player):
            return 1
        if self.is_win(board, -player):
            return -1
        if self.has_moves(board):
            return None
        return 0

    @staticmethod
    def canonical(board, player):                # :100-101
        return player * board

    # bit layout shared with the C oracle and the HIP engine: bit = col*(rows+1) + height
    def to_bits(self, board) -> tuple[int, int]:
        mine = theirs = 0
        b = np.asarray(board)
        for y in range(self.rows):
            for x in range(self.cols):
                bit = 1 << (x * (self.rows + 1) + (self.rows - 1 - y))
                if b[y, x] == 1:
                    mine |= bit
                elif b[y, x] == -1:
                    theirs |= bit
        return mine, theirs

    def legal_mask_bits(self, board) -> int:
        m = 0
        for x, v in enumerate(self.valid_moves(board)):
            m |= v << x
        return m


class C2Rules:
    """Connect2: 1x4 strip, two in a row.  connect2_game.py:14-80."""

    name = "connect2"
    rows, cols, k = 1, 4, 2
    action_size = 4

    def init_board(self):                        # :18-20
        return np.zeros((self.cols,), dtype=int)

    def next_state(self, board, player, action):  # :28-36 (overwrites unconditionally)
        b = np.copy(board)
        b[action] = player
        return b, -player

    def has_moves(self, board) -> bool:          # :38-42
        return bool(np.any(np.asarray(board) == 0))

    def valid_moves(self, board) -> list[int]:   # :44-52
        return [1 if board[x] == 0 else 0 for x in range(self.cols)]

    def is_win(self, board, player) -> bool:     # :54-66
        b = np.asarray(board)
        return bool(np.any((b[:-1] == player) & (b[1:] == player)))

    def reward(self, board, player):             # :68-78
        if self.is_win(board, player):
            return 1
        if self.is_win(board, -player):
            return -1
        if self.has_moves(board):
            return None
        return 0

    @staticmethod
    def canonical(board, player):                # :80
        return player * board

    def to_bits(self, board) -> tuple[int, int]:
        mine = theirs = 0
        for x in range(self.cols):
            bit = 1 << (x * 2)
            if board[x] == 1:
                mine |= bit
            elif board[x] == -1:
                theirs |= bit
        return mine, theirs

    def legal_mask_bits(self, board) -> int:
        m = 0
        for x, v in enumerate(self.valid_moves(board)):
            m |= v << x
        return m


class G9Rules:
    """9x9 Gomoku, five in a row.  NOT in the reference: a new BaseGame plugin (alphazero-implementation_amd/azsp/
    games.py GomokuGame) restated here the way the Connect-4 rules are (action = row*9 + col, no legality check
    in next_state, win before draw).  The reference's MCTS / EpisodeGenerator run it unchanged as a plugin, which
    is how tests/golden/gomoku_hash.json was produced."""

    name = "gomoku9"
    rows, cols, k = 9, 9, 5
    action_size = 81
    words = 2                      # 90-bit boards: two 64-bit words per side

    def __init__(self):
        r, c, k = self.rows, self.cols, self.k
        win = []
        for dy, dx in ((0, 1), (1, 0), (1, 1), (-1, 1)):
            for y in range(r):
                for x in range(c):
                    y2, x2 = y + dy * (k - 1), x + dx * (k - 1)
                    if 0 <= y2 < r and 0 <= x2 < c:
                        win.append([(y + i * dy) * c + (x + i * dx) for i in range(k)])
        self._win_flat = np.array(win, dtype=np.int64)

    def init_board(self):
        return np.zeros((self.rows, self.cols), dtype=int)

    def next_state(self, board, player, action):
        b = np.copy(board)
        y, x = divmod(int(action), self.cols)
        if b[y, x] == 0:
            b[y, x] = player
        return b, -player

    def has_moves(self, board) -> bool:
        return bool(np.any(np.asarray(board) == 0))

    def valid_moves(self, board) -> list[int]:
        return [1 if v == 0 else 0 for v in np.asarray(board).reshape(-1)]

    def is_win(self, board, player) -> bool:
        cells = np.asarray(board).reshape(-1)[self._win_flat]
        return bool(np.any(np.all(cells == player, axis=1)))

    def reward(self, board, player):
        if self.is_win(board, player):
            return 1
        if self.is_win(board, -player):
            return -1
        if self.has_moves(board):
            return None
        return 0

    @staticmethod
    def canonical(board, player):
        return player * board

    # bit layout: bit = row*10 + col (one always-zero sentinel column per row); 90 bits -> two 64-bit words
    def to_bits(self, board) -> tuple[int, int]:
        mine = theirs = 0
        b = np.asarray(board)
        for y in range(self.rows):
            for x in range(self.cols):
                bit = 1 << (y * (self.cols + 1) + x)
                if b[y, x] == 1:
                    mine |= bit
                elif b[y, x] == -1:
                    theirs |= bit
        return mine, theirs

    def legal_mask_bits(self, board) -> int:
        m = 0
        for a, v in enumerate(self.valid_moves(board)):
            m |= v << a
        return m


RULES = {"connect4": C4Rules, "connect2": C2Rules, "gomoku9": G9Rules}


# --------------------------------------------------------------------------------------------
# Hash evaluator (shared definition: oracle/az_oracle.c and csrc/azsp_eval_hash.h restate it)
# --------------------------------------------------------------------------------------------
def _mix(z: int) -> int:
    """splitmix64 finaliser on 64-bit wrap-around integers."""
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def hash_eval_bits(mine: int, theirs: int, legal: int, n_actions: int, salt: int = 0, n_words: int = 1):
    """(P float32[A], V float32) from bitboards.  All intermediate values are exact integers
    except the final float32 division w/S (IEEE correctly rounded on every platform we use)."""
    def fold(b: int) -> int:          # boards of several 64-bit words: fold them (identity for one word)
        h = b & M64
        for k in range(1, n_words):
            h = _mix(h) ^ ((b >> (64 * k)) & M64)
        return h
    h0 = _mix((fold(mine) ^ _mix(fold(theirs)) ^ salt) & M64)
    w = [0] * n_actions
    for a in range(n_actions):
        if (legal >> a) & 1:
            r = _mix((h0 + a + 1) & M64) >> 52          # 12 bits
            w[a] = 1 + ((r * r * r) >> 24)               # 1..4096, heavy tail -> peaky priors
    s = np.float32(sum(w))
    p = np.array([np.float32(x) / s for x in w], dtype=np.float32)
    v = np.float32(int(_mix((h0 + 0x1000) & M64) >> 48) - 32768) / np.float32(32768.0)
    return p, v


class HashModel:
    """Duck-typed stand-in for BaseModel.predict (base_model.py:52-58): list of boards ->
    (list of float32[A], list of np.float32)."""

    def __init__(self, rules, salt: int = 0):
        self.rules = rules
        self.salt = salt
        self.calls = 0
        self.boards_seen = 0

    def predict(self, boards: Sequence[np.ndarray]):
        self.calls += 1
        self.boards_seen += len(boards)
        ps, vs = [], []
        for b in boards:
            mine, theirs = self.rules.to_bits(b)
            p, v = hash_eval_bits(mine, theirs, self.rules.legal_mask_bits(b),
                                  self.rules.action_size, self.salt, getattr(self.rules, "words", 1))
            ps.append(p)
            vs.append(v)
        return ps, vs


# --------------------------------------------------------------------------------------------
# Tree + PUCT  (monte_carlo_tree_search.py:11-99)
# --------------------------------------------------------------------------------------------
class ONode:
    __slots__ = ("n", "to_play", "prior", "w", "kids", "state", "up")

    def __init__(self, prior, to_play, up=None):   # :26-33
        self.n = 0
        self.to_play = to_play
        self.prior = prior
        self.w = 0.0            # python float until the first np.float32 value arrives (NEP 50)
        self.kids: dict[int, ONode] = {}
        self.state = None
        self.up = up

    def is_expanded(self) -> bool:                 # :35-36
        return len(self.kids) > 0

    def q(self):                                   # :38-41
        return 0 if self.n == 0 else self.w / self.n

    def grow(self, state, to_play, probs):         # :80-90  children only where prob != 0
        self.to_play = to_play
        self.state = state
        for a, pr in enumerate(probs):
            if pr != 0:
                self.kids[a] = ONode(pr, -self.to_play, self)

    def best_child(self, c_of=None):               # :63-78  strict '>' from -inf: first maximum wins
        top, top_a, top_c = -np.inf, -1, None
        for a, c in self.kids.items():
            s = puct(self, c, c_of)
            if s > top:
                top, top_a, top_c = s, a, c
        return top_a, top_c

    def pick_action(self, temperature, rng=np.random):   # :43-61
        counts = np.array([c.n for c in self.kids.values()])
        acts = list(self.kids.keys())
        if temperature == 0:
            return acts[int(np.argmax(counts))]
        if temperature == float("inf"):
            return rng.choice(acts)
        d = counts ** (1 / temperature)
        d = d / sum(d)
        return sample_from(acts, d, rng)


def puct(parent: ONode, child: ONode, c_of=None):  # :11-22
    """`c_of` (EXTENSION, None = the reference): exploration factor c(N); the bonus becomes
    prior * (c(N) * sqrt(N)) / (n + 1) -- with c == 1.0 bit-identical to the reference's expression."""
    root_n = math.sqrt(parent.n) if c_of is None else c_of(parent.n) * math.sqrt(parent.n)
    u = child.prior * root_n / (child.n + 1)
    return (-child.q() if child.n > 0 else 0) + u


def exploration_factor(c_init: float = 1.0, c_base: float = 0.0):
    """c(N) of the engine's search options: the constant c_init, or with c_base > 0 the
    log((N + c_base + 1) / c_base) + c_init of src/deepmind_alphazero_pseudocode.py:252-255."""
    if c_base > 0:
        return lambda n: math.log((n + c_base + 1) / c_base) + c_init
    return lambda n: c_init


def sample_from(acts, p, rng=np.random):
    """What ``np.random.choice(acts, p=p)`` does in the legacy RandomState
    (numpy/random/mtrand.pyx ``choice``): cdf = cumsum(p); cdf /= cdf[-1];
    one ``random_sample()`` double; index = searchsorted(cdf, u, side='right')."""
    cdf = np.cumsum(np.asarray(p, dtype=np.float64))
    cdf /= cdf[-1]
    u = rng.random_sample()
    return acts[int(np.searchsorted(cdf, u, side="right"))]


def backup(leaf: ONode, value, to_play):           # :214-223
    node = leaf
    while node is not None:
        node.w += value if node.to_play == to_play else -value
        node.n += 1
        node = node.up


def masked_priors(p32, valid):                     # :119-121 and :165-167
    pr = p32 * valid            # float32[A] * list[int]  ->  float64[A]
    pr /= np.sum(pr)            # float64; n < 8 -> plain left-to-right sum
    return pr


def run_batch(rules, predict: Callable, sims: int, states, to_plays, root_noise=None, noise_eps: float = 0.0,
              c_of=None, numpy1: bool = False, roots_in=None) -> list[ONode]:
    """monte_carlo_tree_search.py:111-171.  `root_noise` (array [n][A]) / `noise_eps` are an EXTENSION beyond the
    reference (off by default): root priors become prior*(1-eps) + noise*eps on existing children, the scheme of
    the reference's src/deepmind_alphazero_pseudocode.py:285-290.
    `numpy1=True` restates what the same source does under numpy 1.x (PARITY UNPINNED: numpy 1.x is not installed
    here): `value_sum += np.float32` promotes to float64 there (pre-NEP-50 scalar promotion), modelled by widening
    the network value to a Python float before it is added.
    `roots_in` (EXTENSION, tree reuse): per slot an already expanded node to continue from, or None for a fresh root."""
    roots = [None] * len(states)
    fresh = list(range(len(states)))
    if roots_in is not None:
        fresh = [i for i, r in enumerate(roots_in) if r is None or not r.is_expanded()]
        for i, r in enumerate(roots_in):
            if r is not None and r.is_expanded():
                roots[i] = r
    p_batch = predict([states[i] for i in fresh])[0] if fresh else []   # :114 root values discarded
    for i, p in zip(fresh, p_batch):
        st, tp = states[i], to_plays[i]
        root = ONode(0, tp)
        pr = masked_priors(p, rules.valid_moves(st))
        if root_noise is not None:
            keep = 1.0 - noise_eps
            pr = np.array([x * keep + root_noise[i][a] * noise_eps if x != 0 else 0.0 for a, x in enumerate(pr)])
        root.grow(st, tp, pr)                                       # :116-123
        roots[i] = root

    for _ in range(sims):                                           # :125
        pending = []
        for root in roots:                                          # :129
            node, act = root, -1
            while node.is_expanded():                               # :130-133
                act, node = node.best_child(c_of)
            par = node.up
            nxt, _ = rules.next_state(par.state, 1, act)            # :140 mover is always +1
            nxt = rules.canonical(nxt, -1)                          # :142
            val = rules.reward(nxt, 1)                              # :145
            if val is None:
                pending.append((node, nxt))                         # :146-149
            else:
                backup(node, val, par.to_play * -1)                 # :150-152 terminal: never expanded
        if pending:                                                 # :154-169
            p_batch, v_batch = predict([s for _, s in pending])
            for (node, nxt), p, v in zip(pending, p_batch, v_batch):
                par = node.up
                node.grow(nxt, par.to_play * -1, masked_priors(p, rules.valid_moves(nxt)))
                backup(node, float(v) if numpy1 else v, par.to_play * -1)
    return roots


def root_counts(root: ONode, n_actions: int) -> list[int]:
    out = [0] * n_actions
    for a, c in root.kids.items():
        out[a] = c.n
    return out


# --------------------------------------------------------------------------------------------
# Lock-step episode generation  (episode_generator.py:23-87)
# --------------------------------------------------------------------------------------------
def generate_episodes(rules, predict: Callable, sims: int, n_slots: int, rng=np.random,
                      trace: list | None = None, sampling_plies: int | None = None,
                      dirichlet_alpha: float | None = None, dirichlet_eps: float = 0.25, c_of=None,
                      numpy1: bool = False, reuse: bool = False):
    """Yields episodes in the reference's (ply, slot) order; each episode is a list of
    (canonical_state int64, policy float64[A], value int).  ``trace`` (optional) receives one
    dict per (ply, slot) with the root visit counts and the chosen action.

    EXTENSIONS beyond the reference (None = reference behaviour; they define what the engine's options must do):
    `sampling_plies`: from that ply of a game on the move is select_action(temperature=0) (mcts:49-50); the
    slot's uniform is still drawn so the stream stays aligned.  `dirichlet_alpha`: per ply, before the search,
    rng.gamma(alpha, 1, (n_slots, A)) is drawn and mixed into the root priors with weight `dirichlet_eps`.
    `reuse`: after a move that does not end the game the chosen child, with everything below it, is the next ply's
    root (its `up` link is cut, so backups stop there) and `sims` further simulations are added to it."""
    A = rules.action_size
    states = [rules.init_board() for _ in range(n_slots)]            # :25
    players = [1] * n_slots                                          # :26
    hist: list[list] = [[] for _ in range(n_slots)]                  # :27
    done = 0
    ply = 0
    keep = [None] * n_slots
    while True:                                                      # :30
        canon = [rules.canonical(s, p) for s, p in zip(states, players)]   # :31-34
        noise = rng.gamma(dirichlet_alpha, 1.0, (n_slots, A)) if dirichlet_alpha else None
        roots = run_batch(rules, predict, sims, canon, players, noise, dirichlet_eps if dirichlet_alpha else 0.0, c_of, numpy1,
                          keep if reuse else None)                                                      # :36
        for i in range(n_slots):                                     # :38-44 slot order = RNG order
            root = roots[i]
            counts = root_counts(root, A)
            pi = counts / np.sum(counts)                             # :45-48  int/int64 -> float64
            hist[i].append((canon[i], players[i], pi))               # :50-52
            if sampling_plies is not None and len(hist[i]) - 1 >= sampling_plies:
                rng.random_sample()                                  # stream alignment (extension)
                act = root.pick_action(0, rng)
            else:
                act = root.pick_action(1, rng)                       # :54
            states[i], players[i] = rules.next_state(states[i], players[i], act)   # :55-58
            rew = rules.reward(states[i], players[i])                # :59
            if trace is not None:
                trace.append({"ply": ply, "slot": i, "counts": counts, "action": int(act),
                              "root_w": float(root.w), "root_n": int(root.n)})
            keep[i] = None
            if reuse and rew is None:
                keep[i] = root.kids[act]
                keep[i].up = None
            if rew is not None:                                      # :61
                ep = [(s, pi_k, rew * ((-1) ** (pl != players[i])))  # :63-76
                      for s, pl, pi_k in hist[i]]
                yield ep
                states[i] = rules.init_board()                       # :80-82
                players[i] = 1
                hist[i] = []
                done += 1
                if done >= n_slots:                                  # :84-87 first N completions
                    return
        ply += 1
