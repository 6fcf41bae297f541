"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from oracle import az_oracle as oz

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class HashStub:
    """A 'model' the engine recognises (azsp_hash_salt) and the oracle can call (predict)."""

    def __init__(self, game_name: str, salt: int):
        self.azsp_hash_salt = salt
        self._m = oz.HashModel(oz.RULES[game_name](), salt)

    def predict(self, boards):
        return self._m.predict(boards)


class HostOnlyModel:
    """Same evaluator, but only reachable through .predict -> exercises the host-callback path."""

    def __init__(self, game_name: str, salt: int):
        self._m = oz.HashModel(oz.RULES[game_name](), salt)
        self.calls = 0

    def predict(self, boards):
        self.calls += 1
        return self._m.predict(boards)


def random_midgame_boards(rules, rng, n, max_plies):
    out = []
    while len(out) < n:
        b, p = rules.init_board(), 1
        ok = True
        for _ in range(int(rng.randint(0, max_plies))):
            legal = [i for i, v in enumerate(rules.valid_moves(b)) if v]
            b, p = rules.next_state(b, p, int(rng.choice(legal)))
            if rules.reward(b, p) is not None:
                ok = False
                break
        if ok:
            out.append(rules.canonical(b, p))
    return out


def check_episode_invariants(rules, states, policies, values, sims):
    """Size-independent properties of one emitted episode (used at BASELINE's full sizes).  Works for any game whose
    action index is the column (Connect-4) or the flat cell index (Gomoku 9x9: action = row * 9 + column)."""
    flat_actions = rules.action_size == int(np.prod(np.asarray(states[0]).shape))
    L = len(states)
    assert 1 <= L
    counts = np.rint(policies * sims).astype(np.int64)
    assert np.all(counts >= 0) and np.all(counts.sum(axis=1) == sims)
    assert np.array_equal(counts / sims, policies)                  # pi is exactly counts / sims
    assert np.all(states[0] == 0)                                   # games start from the empty board
    for k in range(L - 1):
        # next canonical state = flip(previous + one stone of the mover on a column with pi > 0)
        diff = (-states[k + 1]) - states[k]
        assert np.count_nonzero(diff) == 1 and diff.sum() == 1
        where = np.argwhere(diff != 0)[0]
        col = int(where[0] * diff.shape[1] + where[1]) if flat_actions else int(where[-1])
        assert policies[k][col] > 0
    # the final position must be non-terminal, and some move with pi > 0 ends the game with z
    last = states[-1]
    assert rules.reward(last, 1) is None
    z_last = int(values[-1])
    outcomes = set()
    for a in np.flatnonzero(policies[-1] > 0):
        nb, _ = rules.next_state(last, 1, int(a))
        r = rules.reward(nb, -1)
        if r is not None:
            outcomes.add(-r)            # reward is from the next player's view
    assert z_last in outcomes
    expect = [z_last if (L - 1 - k) % 2 == 0 else -z_last for k in range(L)]
    assert [int(v) for v in values] == expect


# The reference's MCTSTests (src/alphazero_simple/tests.py:9-87) are written against an older API and error out
# today; their intent -- Connect2, a mock evaluator with fixed priors, an ordering of root visit counts -- is
# restated here against the list-based predict API.  (priors, value, canonical start board, sims, assertion)
MCTS_INTENT_CASES = [
    ([0.26, 0.24, 0.24, 0.26], 0.0001, [0, 0, 0, 0], 50,        # tests.py:10-33 centre beats the edge
     lambda c: max(c[1], c[2]) > c[0]),
    ([0.3, 0.7, 0.0, 0.0], 0.0001, [0, 0, 1, -1], 25,           # tests.py:35-52
     lambda c: c[1] > c[0]),
    ([0.51, 0.49, 0.0, 0.0], 0.0001, [0, 0, -1, 1], 25,         # tests.py:54-68 the winning move overtakes the prior
     lambda c: c[0] < c[1]),
    ([0.0, 0.3, 0.3, 0.3], 0.0001, [-1, 0, 0, 0], 100,          # tests.py:70-87
     lambda c: c[1] > c[2] and c[1] > c[3]),
]


class FixedPriorModel:
    def __init__(self, priors, value):
        self.p = np.array(priors, dtype=np.float32)
        self.v = np.float32(value)

    def predict(self, boards):
        return [self.p.copy() for _ in boards], [self.v for _ in boards]


def min_puct_margin(rules, predict, sims, state):
    """Smallest gap between the best and the second-best PUCT score over every selection step of one search from
    `state` (oracle search fed by `predict`): how close the search came to a tie.  Used to qualify a divergence between
    a transcript recorded with torch-CPU network floats and a run with the device network's floats."""
    best = {"gap": float("inf")}
    orig = oz.ONode.best_child

    def spy(self, c_of=None):
        scores = sorted((float(oz.puct(self, c, c_of)) for c in self.kids.values()), reverse=True)
        if len(scores) > 1:
            best["gap"] = min(best["gap"], scores[0] - scores[1])
        return orig(self, c_of)

    oz.ONode.best_child = spy
    try:
        oz.run_batch(rules, predict, sims, [np.asarray(state)], [1])
    finally:
        oz.ONode.best_child = orig
    return best["gap"]
