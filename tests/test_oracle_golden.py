"""Pins oracle/az_oracle.py against (1) the fixtures produced by running the reference itself
(tests/golden/make_golden.py) and (2) the reference's own known-answer tests for Node/ucb_score
(/root/reference/src/alphazero_simple/tests.py:90-241, restated here as data)."""
import json
import os

import numpy as np
import pytest

from oracle import az_oracle as oz

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


# ----------------------------------------------------------------------------------- rules
@pytest.mark.parametrize("fname,rules", [("rules_connect4.json", oz.C4Rules()),
                                         ("rules_connect2.json", oz.C2Rules())])
def test_rules_match_reference(fname, rules):
    for c in load(fname)["cases"]:
        b = np.array(c["board"], dtype=int)
        nb, npl = rules.next_state(b, c["player"], c["action"])
        assert nb.tolist() == c["next"] and npl == c["next_player"]
        assert rules.valid_moves(b) == c["valid"]
        assert rules.has_moves(b) == c["has_moves"]
        assert rules.is_win(b, 1) == c["win_p"] and rules.is_win(b, -1) == c["win_m"]
        assert rules.reward(b, 1) == c["reward_p"] and rules.reward(b, -1) == c["reward_m"]
        if "canon_m" in c:
            assert rules.canonical(b, -1).tolist() == c["canon_m"]


# ----------------------------------------------------------------------------------- node KATs
def _kat_node(parent_n, child_ns):
    node = oz.ONode(0.5, to_play=1)
    node.n = parent_n
    node.grow([0, 0, 0, 0], 1, [0.25, 0.15, 0.5, 0.1])
    for a, n in enumerate(child_ns):
        node.kids[a].n = n
    return node


def test_node_init_kat():                       # tests.py:91-98
    n = oz.ONode(0.5, to_play=1)
    assert n.n == 0 and n.prior == 0.5 and len(n.kids) == 0 and not n.is_expanded() and n.q() == 0


def test_node_argmax_selection_kat():           # tests.py:100-117
    n = oz.ONode(0.5, 1)
    n.n = 1
    n.kids = {0: oz.ONode(0.5, -1), 1: oz.ONode(0.5, -1), 2: oz.ONode(0.5, -1)}
    n.kids[2].n = 1
    assert n.pick_action(temperature=0) == 2


def test_node_expansion_kat():                  # tests.py:119-134
    n = _kat_node(0, [0, 0, 0, 0])
    assert len(n.kids) == 4 and n.is_expanded() and n.to_play == 1
    assert [n.kids[a].prior for a in range(4)] == [0.25, 0.15, 0.5, 0.1]
    assert all(c.to_play == -1 for c in n.kids.values())


def test_ucb_unvisited_children_score_is_prior_kat():   # tests.py:217-241 / :136-158
    n = _kat_node(1, [0, 0, 0, 0])
    for a in range(4):
        assert oz.puct(n, n.kids[a]) == n.kids[a].prior


def test_ucb_one_child_visited_kat():           # tests.py:160-185
    n = _kat_node(1, [0, 0, 1, 0])
    assert oz.puct(n, n.kids[2]) == n.kids[2].prior / 2
    assert oz.puct(n, n.kids[0]) == n.kids[0].prior
    a, _ = n.best_child()
    assert a == 0


def test_ucb_child_visited_twice_kat():         # tests.py:187-215
    n = _kat_node(2, [0, 0, 2, 0])
    a, _ = n.best_child()
    assert a == 0


# ----------------------------------------------------------------------------------- search
def _check_root(root, g, A):
    assert root.n == g["n"]
    assert float(root.w) == g["w"] and isinstance(root.w, np.float32) == g["w_is_f32"]
    assert sorted(root.kids) == sorted(int(a) for a in g["kids"])
    for a, k in g["kids"].items():
        c = root.kids[int(a)]
        assert c.n == k["n"]
        assert float(c.w) == k["w"], (a, c.w, k["w"])
        assert isinstance(c.w, np.float32) == k["w_is_f32"]
        assert float(c.prior) == k["prior"]
        assert c.is_expanded() == k["expanded"]


def test_run_batch_matches_reference():
    for run in load("search_hash.json")["runs"]:
        if run["sims"] > 100:
            continue                    # the 400-sim transcripts are checked by the C oracle test
        rules = oz.RULES[run["game"]]()
        model = oz.HashModel(rules, run["salt"])
        boards = [np.array(b, dtype=int) for b in run["boards"]]
        roots = oz.run_batch(rules, model.predict, run["sims"], boards, run["to_plays"])
        for r, g in zip(roots, run["roots"]):
            _check_root(r, g, rules.action_size)


# ----------------------------------------------------------------------------------- episodes
def test_generate_episodes_matches_reference():
    for run in load("episodes_hash.json")["runs"]:
        if run["n"] * run["sims"] > 1700:
            continue                    # larger transcripts: C oracle test (pure python is slow)
        rules = oz.RULES[run["game"]]()
        model = oz.HashModel(rules, run["salt"])
        np.random.seed(run["seed"])
        trace = []
        eps = list(oz.generate_episodes(rules, model.predict, run["sims"], run["n"], trace=trace))
        assert float(np.random.random_sample()) == run["rng_after"]     # same number of draws
        assert len(eps) == len(run["episodes"]) == run["n"]
        for ep, g in zip(eps, run["episodes"]):
            assert [np.asarray(s).tolist() for s, _, _ in ep] == g["states"]
            assert [np.asarray(p).tolist() for _, p, _ in ep] == g["policies"]
            assert [int(z) for _, _, z in ep] == g["values"]
            assert all(type(z) is int for _, _, z in ep) and g["value_types"] == ["int"]
        it = iter(trace)
        for ply in run["plies"]:
            for slot, act in enumerate(ply["actions"]):
                t = next(it)
                assert t["slot"] == slot and t["counts"] == ply["counts"][slot]
                assert t["action"] == act and t["root_w"] == ply["root_w"][slot]


def test_hash_evaluator_properties():
    rules = oz.C4Rules()
    b = rules.init_board()
    p, v = oz.hash_eval_bits(*rules.to_bits(b), rules.legal_mask_bits(b), 7, 0)
    assert p.dtype == np.float32 and abs(float(p.sum()) - 1) < 1e-6 and -1 <= float(v) < 1
    b[0, 3] = 1
    p, _ = oz.hash_eval_bits(*rules.to_bits(b), rules.legal_mask_bits(b), 7, 0)
    assert p[3] == 0 and np.all(p[[0, 1, 2, 4, 5, 6]] > 0)
