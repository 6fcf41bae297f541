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


# ----------------------------------------------------------------------------------- C oracle
from oracle import c_oracle as co  # noqa: E402


def test_c_oracle_mt19937_matches_numpy():
    for seed in (0, 1, 12345, 2**31 + 5):
        np.random.seed(seed)
        assert np.array_equal(co.mt_uniforms(seed, 1500), np.random.random_sample(1500))


def test_c_oracle_hash_eval_matches_python():
    rng = np.random.RandomState(3)
    r4, r2 = oz.C4Rules(), oz.C2Rules()
    for _ in range(300):
        b = rng.randint(-1, 2, size=(6, 7))
        p, v = oz.hash_eval_bits(*r4.to_bits(b), r4.legal_mask_bits(b), 7, 77)
        pc, vc = co.hash_eval("connect4", b, 77)
        assert np.array_equal(p, pc, equal_nan=True) and v == vc
    for _ in range(50):
        b = rng.randint(-1, 2, size=(4,))
        p, v = oz.hash_eval_bits(*r2.to_bits(b), r2.legal_mask_bits(b), 4, 5)
        pc, vc = co.hash_eval("connect2", b, 5)
        assert np.array_equal(p, pc, equal_nan=True) and v == vc


def test_c_oracle_run_batch_matches_reference():
    for run in load("search_hash.json")["runs"]:
        boards = np.array(run["boards"], dtype=np.int8)
        out = co.run_batch(run["game"], run["sims"], boards, run["to_plays"], run["salt"])
        A = out["counts"].shape[1]
        for i, g in enumerate(run["roots"]):
            assert out["root_n"][i] == g["n"] and float(out["root_w"][i]) == g["w"]
            assert out["n_nodes"][i] == g["nodes"]
            for a in range(A):
                k = g["kids"].get(str(a))
                if k is None:
                    assert out["counts"][i, a] == 0 and out["prior"][i, a] == 0.0
                else:
                    assert out["counts"][i, a] == k["n"]
                    assert float(out["child_w"][i, a]) == k["w"]
                    assert float(out["prior"][i, a]) == k["prior"]


def _check_generate(o, run):
    assert o["n_episodes"] == run["n"]
    assert o["next_uniform"] == run["rng_after"]
    for e, g in enumerate(run["episodes"]):
        lo, hi = o["ep_offset"][e], o["ep_offset"][e + 1]
        shape = (-1,) + tuple(o["shape"])
        assert o["states"][lo:hi].reshape(shape).tolist() == g["states"]
        assert o["policies"][lo:hi].tolist() == g["policies"]
        assert o["values"][lo:hi].tolist() == g["values"]
        assert o["ep_ply"][e] == g["ply"]
    row = 0
    for ply in run["plies"]:
        for slot, act in enumerate(ply["actions"]):
            assert o["trace_counts"][row].tolist() == ply["counts"][slot]
            assert o["trace_actions"][row] == act
            row += 1
    assert row == o["trace_rows"]


def test_c_oracle_generate_matches_reference():
    for run in load("episodes_hash.json")["runs"]:
        _check_generate(co.generate(run["game"], run["sims"], run["n"], run["salt"], seed=run["seed"]), run)


def test_c_oracle_generate_with_numpy_rng_and_python_evaluator():
    """Same transcripts through the callback paths (python predict, global numpy RNG)."""
    run = [r for r in load("episodes_hash.json")["runs"] if r["game"] == "connect4" and r["n"] == 4][0]
    rules = oz.C4Rules()
    model = oz.HashModel(rules, run["salt"])
    np.random.seed(run["seed"])
    cb = co.make_predict_cb(model.predict, (6, 7), 7)
    o = co.generate("connect4", run["sims"], run["n"], seed=None, predict_cb=cb, uniform_cb=co.np_uniform_cb())
    _check_generate(o, run)


# ----------------------------------------------------------------------------------- network oracle
def test_resnet_ref_matches_reference_vectors():
    """oracle/resnet_ref.py vs outputs of the reference's own ResNet.predict (torch CPU, 1 thread)."""
    import torch
    from oracle.resnet_ref import ResNetRef
    torch.set_num_threads(1)
    z = np.load(os.path.join(G, "resnet_forward.npz"))
    boards = list(z["boards"].astype(np.int64))
    for tag in ("s", "m"):
        net = ResNetRef({k[5:]: z[k] for k in z.files if k.startswith(f"{tag}_sd/")})
        p, v = net.predict(boards)
        assert np.max(np.abs(np.stack(p) - z[f"{tag}_p"])) < 1e-6
        assert np.max(np.abs(np.array(v) - z[f"{tag}_v"])) < 1e-6
        assert np.all(np.stack(p)[z[f"{tag}_p"] == 0] == 0)


def test_oracle_with_resnet_reproduces_reference_config1():
    """BASELINE.json configs[0] (25 sims, 1 game, torch-CPU ResNet(4,64), seeds 0/0).  torch-CPU
    kernels are host dependent, so bit-equality of the transcript is only demanded when this host's
    network outputs are bit-identical to the recorded ones; otherwise pi is compared within 0.08
    on the first ply (a single visit of 25) and the test says so."""
    import torch
    from oracle.resnet_ref import ResNetRef
    run = load("episodes_resnet.json")
    torch.manual_seed(run["torch_seed"])
    torch.set_num_threads(1)
    import azsp
    ref_like = azsp.ResNet((6, 7), 7, *run["arch"])          # same init order as the reference module
    net = ResNetRef(ref_like.state_dict())
    np.random.seed(run["seed"])
    rules = oz.C4Rules()
    trace = []
    eps = list(oz.generate_episodes(rules, net.predict, run["sims"], run["n"], trace=trace))
    same = [t["counts"] for t in trace] == [p["counts"][0] for p in run["plies"]]
    if same:
        assert [np.asarray(s).tolist() for s, _, _ in eps[0]] == run["episodes"][0]["states"]
        assert [np.asarray(p).tolist() for _, p, _ in eps[0]] == run["episodes"][0]["policies"]
        assert [int(z) for _, _, z in eps[0]] == run["episodes"][0]["values"]
        assert float(np.random.random_sample()) == run["rng_after"]
    else:
        first = np.array(trace[0]["counts"]) / run["sims"]
        assert np.max(np.abs(first - np.array(run["plies"][0]["counts"][0]) / run["sims"])) <= 0.08
        pytest.skip("torch-CPU outputs differ bitwise on this host; transcript compared on ply 0 only")


def test_reference_mcts_tests_intent_holds_in_the_oracle():
    from helpers import MCTS_INTENT_CASES, FixedPriorModel
    rules = oz.C2Rules()
    for priors, value, board, sims, check in MCTS_INTENT_CASES:
        root = oz.run_batch(rules, FixedPriorModel(priors, value).predict, sims, [np.array(board)], [1])[0]
        counts = oz.root_counts(root, 4)
        assert sum(counts) == sims and check(counts), (priors, counts)


# ----------------------------------------------------------------------------------- Gomoku (new plugin game)
def test_gomoku_search_and_episodes_match_reference_algorithm_with_plugin_game():
    """tests/golden/gomoku_hash.json: the reference's MCTS.run_batch / generate_episodes run with the NEW 9x9
    Gomoku BaseGame plugin (81 actions: numpy pairwise np.sum over the priors, 81-way sampling).  Both oracles
    must reproduce it bit for bit."""
    g = load("gomoku_hash.json")
    rules = oz.G9Rules()
    for run in g["search"]:
        boards = [np.array(b, dtype=int) for b in run["boards"]]
        roots = oz.run_batch(rules, oz.HashModel(rules, run["salt"]).predict, run["sims"], boards, run["to_plays"])
        out = co.run_batch("gomoku9", run["sims"], np.array(run["boards"], dtype=np.int8), run["to_plays"], run["salt"])
        for i, (r, gr) in enumerate(zip(roots, run["roots"])):
            _check_root(r, gr, 81)
            assert out["root_n"][i] == gr["n"] and float(out["root_w"][i]) == gr["w"] and out["n_nodes"][i] == gr["nodes"]
            for a in range(81):
                k = gr["kids"].get(str(a))
                if k is None:
                    assert out["counts"][i, a] == 0 and out["prior"][i, a] == 0.0
                else:
                    assert out["counts"][i, a] == k["n"] and float(out["child_w"][i, a]) == k["w"]
                    assert float(out["prior"][i, a]) == k["prior"]
    for run in g["episodes"]:
        _check_generate(co.generate("gomoku9", run["sims"], run["n"], run["salt"], seed=run["seed"]), run)
        np.random.seed(run["seed"])
        eps = list(oz.generate_episodes(rules, oz.HashModel(rules, run["salt"]).predict, run["sims"], run["n"]))
        assert float(np.random.random_sample()) == run["rng_after"]
        for ep, ge in zip(eps, run["episodes"]):
            assert [np.asarray(s).tolist() for s, _, _ in ep] == ge["states"]
            assert [np.asarray(p).tolist() for _, p, _ in ep] == ge["policies"]
            assert [int(z) for _, _, z in ep] == ge["values"]


def test_host_gomoku_game_matches_oracle_rules():
    import azsp
    game, rules = azsp.GomokuGame(), oz.G9Rules()
    rng = np.random.RandomState(5)
    for _ in range(30):
        b, p = game.get_init_board(), 1
        for _ in range(81):
            legal = [i for i, v in enumerate(game.get_valid_moves(b)) if v]
            assert game.get_valid_moves(b) == rules.valid_moves(b)
            a = int(rng.choice(legal))
            nb, np_ = game.get_next_state(b, p, a)
            ob, op = rules.next_state(b, p, a)
            assert np.array_equal(nb, ob) and np_ == op
            b, p = nb, np_
            assert game.get_reward_for_player(b, p) == rules.reward(b, p)
            assert game.is_win(b, 1) == rules.is_win(b, 1) and game.is_win(b, -1) == rules.is_win(b, -1)
            if rules.reward(b, p) is not None:
                break
    full = np.ones((9, 9), dtype=int)
    assert np.array_equal(game.get_next_state(full, -1, 40)[0], full)          # occupied cell: no-op
