"""CPU ORACLE for the reference's first-generation package (test infrastructure only -- never imported
by the product path).  **PARITY UNPINNED**: `src/alphazero_implementation` imports the un-vendored
`simulator` package (board-game-simulator-python@0.0.4, requirements.txt:9), which is not installable
here, so the reference's gen-1 search cannot be run and holds no fixtures for it.  What follows restates the
published algorithm from the source text alone; the engine's gen-1 rule set is tested bit-for-bit against
THIS restatement, nothing stronger.

Restated (paths relative to /root/reference/src/alphazero_implementation):
  * Node                      core/search/mcts/node.py:7-73
  * AlphaZeroSearch           core/search/mcts/search.py:10-91   (select_child :27-46, backpropagate :48-58,
                                                                  run_simulations :66-91)
  * EpisodeGenerator          core/training/episode_generator.py:12-81
  * Connect4Model.predict     models/games/connect4/model.py:19-42 (softmax over the legal logits only)

`GState` / `GAction` below are a stand-in for the attributes of `simulator.game.connect.State/Action` that the
call sites above touch (grid, player, actions, has_ended, reward, config, column, sample_next_state): a
height x width Connect-k game, grid cell -1 = empty / 0 / 1 = player, row 0 = bottom
(notebooks/policy_comparison.ipynb cell 6), actions in ascending column order, reward +1 winner / -1 loser /
0 draw or unfinished.  Those conventions are ASSUMED (the package is absent); they are stated here so the
assumption is visible.
"""
from __future__ import annotations

import math
from typing import Callable

import numpy as np


class GConfig:
    def __init__(self, height=6, width=7, count=4, num_players=2):
        self.height, self.width, self.count, self.num_players = height, width, count, num_players

    def sample_initial_state(self):
        return GState(self, np.full((self.height, self.width), -1, dtype=np.int8), 0)


class GAction:
    def __init__(self, state: "GState", column: int):
        self.state, self.column = state, column

    def sample_next_state(self) -> "GState":
        s = self.state
        grid = s.grid.copy()
        for r in range(s.config.height):          # row 0 = bottom
            if grid[r, self.column] == -1:
                grid[r, self.column] = s.player
                break
        return GState(s.config, grid, 1 - s.player, last=(r, self.column))

    def __hash__(self):
        return hash(self.column)

    def __eq__(self, other):
        return isinstance(other, GAction) and other.column == self.column and other.state is self.state


class GState:
    def __init__(self, config: GConfig, grid, player: int, last=None):
        self.config, self.grid, self.player = config, grid, player
        self._winner = self._find_winner(last)
        full = not (grid[config.height - 1] == -1).any()
        self.has_ended = self._winner is not None or full
        self.reward = np.zeros(config.num_players)
        if self._winner is not None:
            self.reward[:] = -1.0
            self.reward[self._winner] = 1.0
        self.actions = [] if self.has_ended else [GAction(self, c) for c in range(config.width)
                                                  if grid[config.height - 1, c] == -1]

    def _find_winner(self, last):
        if last is None:
            return None
        g, k = self.grid, self.config.count
        r0, c0 = last
        who = g[r0, c0]
        for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
            run = 1
            for sgn in (1, -1):
                r, c = r0 + sgn * dr, c0 + sgn * dc
                while 0 <= r < self.config.height and 0 <= c < self.config.width and g[r, c] == who:
                    run += 1
                    r, c = r + sgn * dr, c + sgn * dc
            if run >= k:
                return int(who)
        return None


# ------------------------------------------------------------------------------------------------
# node.py:7-73
# ------------------------------------------------------------------------------------------------
class G1Node:
    def __init__(self, state, parent=None, prior=0.0):      # :8-17
        self.state, self.parent, self.prior = state, parent, prior
        self.children: dict = {}
        self.visit_count = 0
        self.value_sum = 0.0

    @property
    def improved_policy(self):                              # :24-29  n_child / (N - 1)
        return {a: c.visit_count / (self.visit_count - 1) for a, c in self.children.items()}

    def select_next_node(self, rng=np.random):              # :31-42  fresh node, parent = old root
        pol = self.improved_policy
        index = rng.choice(len(pol), p=list(pol.values()))
        action = list(pol.keys())[index]
        return G1Node(action.sample_next_state(), self, pol[action]), action

    @property
    def value(self):                                        # :50-55
        return 0 if self.visit_count == 0 else self.value_sum / self.visit_count

    @property
    def is_expanded(self):                                  # :57-59
        return len(self.children) > 0

    @property
    def is_terminal(self):                                  # :61-63
        return self.state.has_ended


# ------------------------------------------------------------------------------------------------
# search.py:10-91
# ------------------------------------------------------------------------------------------------
def select_child(node: G1Node, c_puct: float):              # :27-46   Q is NOT negated; strict '>'
    best, best_child = float("-inf"), None
    for child in node.children.values():
        u = c_puct * child.prior * math.sqrt(node.visit_count) / (1 + child.visit_count)
        score = child.value + u
        if score > best:
            best, best_child = score, child
    return best_child


def backpropagate(leaf: G1Node, value: float):              # :48-58  no sign flip above a terminal node
    node = leaf
    while node is not None:
        node.value_sum += value
        node.visit_count += 1
        if not node.is_terminal:
            value = -value
        node = node.parent


def run_simulations(predict: Callable, roots: list[G1Node], sims: int, c_puct: float = 1.0):   # :66-91
    for _ in range(sims):
        todo = []
        for root in roots:
            node = root
            while node.is_expanded:
                node = select_child(node, c_puct)
            if node.is_terminal:
                backpropagate(node, node.state.reward.tolist()[node.parent.state.player])
            else:
                todo.append(node)
        if todo:
            pols, vals = predict([n.state for n in todo])
            for node, pol, val in zip(todo, pols, vals):
                for action, prob in pol.items():            # children for EVERY legal action, eagerly
                    node.children[action] = G1Node(action.sample_next_state(), node, prob)
                backpropagate(node, val[node.state.player])


# ------------------------------------------------------------------------------------------------
# episode_generator.py:35-81
# ------------------------------------------------------------------------------------------------
def generate_episodes(predict: Callable, sims: int, n_slots: int, initial_state, c_puct: float = 1.0,
                      rng=np.random, trace: list | None = None):
    """Yields episodes as lists of (state, policy dict, value list), in the reference's (ply, slot) order."""
    episodes = [[] for _ in range(n_slots)]
    nodes = [G1Node(initial_state) for _ in range(n_slots)]
    done, ply = 0, 0
    while True:
        run_simulations(predict, nodes, sims, c_puct)
        for i, node in enumerate(nodes):
            episodes[i].append([node.state, node.improved_policy, [0.0] * initial_state.config.num_players])
            nxt, action = node.select_next_node(rng)
            if trace is not None:
                trace.append({"ply": ply, "slot": i, "action": action.column,
                              "counts": {a.column: c.visit_count for a, c in node.children.items()},
                              "root_w": node.value_sum, "root_n": node.visit_count,
                              "child_w": {a.column: c.value_sum for a, c in node.children.items()}})
            if not nxt.is_terminal:
                nodes[i] = nxt
                continue
            outcome = nxt.state.reward.tolist()             # episode.py:52-54 every sample gets the same vector
            for smp in episodes[i]:
                smp[2] = outcome
            yield episodes[i]
            nodes[i] = G1Node(initial_state)
            episodes[i] = []
            done += 1
            if done >= n_slots:
                return
        ply += 1


# ------------------------------------------------------------------------------------------------
# hash "model": (P, V) from the integer hash evaluator of oracle/az_oracle.py, presented through the
# gen-1 predict surface (models/base/model.py:54-74): dict over the legal actions, value per player
# ------------------------------------------------------------------------------------------------
def canonical_cells(state) -> np.ndarray:
    """gen-1 grid (row 0 = bottom, -1/0/1, absolute players) -> the engine's canonical int8 cells
    (row 0 = top, +1 = side to move, -1 = opponent)."""
    g = np.asarray(state.grid)[::-1]
    out = np.zeros(g.shape, np.int8)
    out[g == state.player] = 1
    out[g == 1 - state.player] = -1
    return out


class HashModelG1:
    def __init__(self, salt: int = 0):
        from .az_oracle import C4Rules, hash_eval_bits
        self.rules, self.salt, self._eval = C4Rules(), salt, hash_eval_bits

    def predict(self, states):
        pols, vals = [], []
        for s in states:
            cells = canonical_cells(s).astype(np.int64)
            mine, theirs = self.rules.to_bits(cells)
            p32, v32 = self._eval(mine, theirs, self.rules.legal_mask_bits(cells), 7, self.salt)
            pols.append({a: float(p32[a.column]) for a in s.actions})
            v = float(v32)
            vals.append([v, -v] if s.player == 0 else [-v, v])      # value[state.player] == v
        return pols, vals
