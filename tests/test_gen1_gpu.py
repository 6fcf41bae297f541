"""GPU parity tests of the engine's second rule set (AZSP_RULES_GEN1: the search of the reference's
first-generation package `src/alphazero_implementation`) and of the gen-1 plugin surface in azsp/gen1.py.

PARITY UNPINNED: that package needs the un-vendored `simulator` library, so the reference could not be run for
it and holds no fixtures; these tests compare the engine bit for bit (float64 value sums, visit counts, moves,
RNG position) with oracle/gen1_oracle.py, a restatement written from the source text."""
import random

import numpy as np
import pytest

from oracle import az_oracle as oz
from oracle import gen1_oracle as g1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen1():
    from azsp import _lib, gen1 as mod
    _lib.load()
    return mod


def _midgame_states(rng, n, max_plies, factory):
    out = []
    while len(out) < n:
        s = factory()
        for _ in range(int(rng.randint(0, max_plies))):
            if s.has_ended:
                break
            s = s.actions[int(rng.randint(len(s.actions)))].sample_next_state()
        if not s.has_ended:
            out.append(s)
    return out


def _mirror(s, cfg):
    """the same position as an oracle GState (independent rules implementation)."""
    return g1.GState(cfg, np.asarray(s.grid).copy(), s.player)


@pytest.mark.parametrize("sims,c_puct", [(1, 1.0), (2, 1.0), (25, 1.0), (100, 1.0), (60, 2.5), (400, 0.3)])
def test_gen1_search_statistics_bit_exact(gen1, sims, c_puct):
    """AlphaZeroSearch.run_simulations on mid-game positions: root and child visit counts, float64 value sums
    and priors equal the restated search.py / node.py arithmetic exactly (Q not negated, root counted at its
    expansion, no flip above terminal nodes, children for every legal move)."""
    from azsp import connect_state as cs
    rng = np.random.RandomState(sims)
    cfg, ocfg = cs.Config(), g1.GConfig()
    states = _midgame_states(rng, 40, 38, cfg.sample_initial_state) + [cfg.sample_initial_state()]
    model = gen1.HashModel(17)
    search = gen1.AlphaZeroSearch(model=model, num_simulations=sims, exploration_weight=c_puct)
    nodes = [gen1.Node(s) for s in states]
    search.run_simulations(nodes)
    oroots = [g1.G1Node(_mirror(s, ocfg)) for s in states]
    g1.run_simulations(g1.HashModelG1(17).predict, oroots, sims, c_puct)
    for node, ref in zip(nodes, oroots):
        assert node.visit_count == ref.visit_count == sims
        assert node.value_sum == ref.value_sum
        assert [a.column for a in node.children] == [a.column for a in ref.children]
        for (a, c), (ra, rc) in zip(node.children.items(), ref.children.items()):
            assert (c.visit_count, c.value_sum, c.prior) == (rc.visit_count, rc.value_sum, rc.prior), (a.column,)
        if sims > 1:
            assert list(node.improved_policy.values()) == list(ref.improved_policy.values())
        assert node.value == ref.value


def test_gen1_second_run_on_a_used_node_is_refused(gen1):
    """search.py:66-91 continues a node's existing tree when run() is called again; the device search cannot (it starts
    from an unexpanded root), so a used Node is refused instead of being restarted with mismatched statistics."""
    from azsp import connect_state as cs
    search = gen1.AlphaZeroSearch(model=gen1.HashModel(3), num_simulations=20)
    root = gen1.Node(cs.Config().sample_initial_state())
    policy, _ = search.run(root)
    assert abs(sum(policy.values()) - 1.0) < 1e-12 and root.visit_count == 20
    with pytest.raises(ValueError, match="fresh nodes"):
        search.run(root)
    again, _ = search.run(gen1.Node(root.state))          # a new Node on the same state is the supported way
    assert again == policy


def _compare_episodes(eps, ref):
    assert len(eps) == len(ref)
    for ep, r in zip(eps, ref):
        assert len(ep) == len(r)
        for smp, (rs, rpol, rval) in zip(ep.samples, r):
            assert np.array_equal(np.asarray(smp.state.grid), rs.grid) and smp.state.player == rs.player
            assert [a.column for a in smp.policy] == [a.column for a in rpol]
            assert list(smp.policy.values()) == list(rpol.values())
            assert list(smp.value) == list(rval)


@pytest.mark.parametrize("n,sims,seed,c_puct", [(1, 25, 0, 1.0), (8, 30, 1, 1.0), (64, 100, 2, 1.0), (16, 50, 3, 1.7)])
def test_gen1_generate_episodes_bit_exact(gen1, n, sims, seed, c_puct):
    """The gen-1 EpisodeGenerator surface end to end: episodes in (ply, slot) order, State objects, policy dicts
    n/(N-1), outcome vectors and the final position of the global numpy RNG."""
    from azsp import connect_state as cs
    np.random.seed(seed)
    ref = list(g1.generate_episodes(g1.HashModelG1(5).predict, sims, n, g1.GConfig().sample_initial_state(), c_puct))
    after = float(np.random.random_sample())
    np.random.seed(seed)
    gen = gen1.EpisodeGenerator(model=gen1.HashModel(5), num_simulations=sims, num_episodes=n,
                                game_initial_state=cs.Config(6, 7, 4).sample_initial_state(), exploration_weight=c_puct)
    eps = list(gen.generate_episodes())
    assert float(np.random.random_sample()) == after
    _compare_episodes(eps, ref)
    # a second call reuses the engine and starts from fresh games
    np.random.seed(seed)
    _compare_episodes(list(gen.generate_episodes()), ref)


def test_gen1_generate_from_a_given_initial_state(gen1):
    """generate_episodes(initial_state) (episode_generator.py:35-45): every game and every slot restart begins at
    the given position, here three plies in with player 1 to move."""
    from azsp import connect_state as cs
    s, o = cs.Config().sample_initial_state(), g1.GConfig().sample_initial_state()
    for col in (3, 3, 2):
        s = [a for a in s.actions if a.column == col][0].sample_next_state()
        o = [a for a in o.actions if a.column == col][0].sample_next_state()
    assert s.player == 1
    np.random.seed(11)
    ref = list(g1.generate_episodes(g1.HashModelG1(8).predict, 40, 12, o))
    np.random.seed(11)
    gen = gen1.EpisodeGenerator(model=gen1.HashModel(8), num_simulations=40, num_episodes=12,
                                game_initial_state=cs.Config().sample_initial_state())
    _compare_episodes(list(gen.generate_episodes(s)), ref)


class _HostHashModel:
    """A gen-1 Model only reachable through .predict (State objects in, dicts out): exercises the host-callback
    evaluator and the state factory."""

    def __init__(self, salt, scale=None):
        self._m = g1.HashModelG1(salt)
        self.calls = 0
        self.scale = scale

    def predict(self, states):
        self.calls += 1
        pols, vals = self._m.predict(states)
        if self.scale is not None:       # full-mantissa float32 values (the hash values are multiples of 2^-15, whose
            vals = [[float(np.float32(x) * np.float32(self.scale)) for x in v] for v in vals]   # sums are always exact)
        return pols, vals

    def get_inference_clone(self):
        return self


@pytest.mark.parametrize("scale,sims", [(None, 20), (0.7310586, 80)])
def test_gen1_host_model_predict_path(gen1, scale, sims):
    from azsp import connect_state as cs
    n, seed = 6, 4
    np.random.seed(seed)
    ref = list(g1.generate_episodes(_HostHashModel(2, scale).predict, sims, n, g1.GConfig().sample_initial_state()))
    np.random.seed(seed)
    model = _HostHashModel(2, scale)
    gen = gen1.EpisodeGenerator(model=model, num_simulations=sims, num_episodes=n,
                                game_initial_state=cs.Config().sample_initial_state())
    _compare_episodes(list(gen.generate_episodes()), ref)
    assert model.calls > 0


def test_gen1_device_network_equals_its_own_host_predict(gen1):
    """ResNetModel: the device ResNet evaluator inside the engine and the same network called back through
    Model.predict (one fused-kernel call per round, bitwise batch-invariant) give identical episodes."""
    import torch
    from azsp import connect_state as cs
    from azsp.model import ResNet
    torch.manual_seed(0)
    net = ResNet((6, 7), 7, 2, 32).eval()
    wrapped = gen1.ResNetModel(net)

    class HostOnly:                      # hides state_dict: forces the host-callback evaluator
        def __init__(self, m):
            self.m = m

        def predict(self, states):
            return self.m.predict(states)

    def run(model):
        np.random.seed(21)
        gen = gen1.EpisodeGenerator(model=model, num_simulations=24, num_episodes=5,
                                    game_initial_state=cs.Config().sample_initial_state())
        return list(gen.generate_episodes())

    dev, host = run(wrapped), run(HostOnly(wrapped))
    assert len(dev) == len(host) == 5
    for a, b in zip(dev, host):
        assert [s.state for s in a.samples] == [s.state for s in b.samples]
        assert [list(s.policy.values()) for s in a.samples] == [list(s.policy.values()) for s in b.samples]
        assert [s.value for s in a.samples] == [s.value for s in b.samples]
    # the policy / value surface itself (models/games/connect4/model.py:19-42): softmax over the legal columns
    st = dev[0].samples[3].state
    pol, val = wrapped.predict([st])
    assert abs(sum(pol[0].values()) - 1.0) < 1e-6 and val[0][0] == -val[0][1]
    assert [a.column for a in pol[0]] == [a.column for a in st.actions]


def test_gen1_players_and_arena(gen1):
    """ui/cli/player.py + arena.py without the TUI: batched games run to the end with legal moves only, and the
    searching player (arg-max of the improved policy) takes a win that is one move away (a terminal child carries
    the mover's utility +1, search.py:78-80).  Not a strength test: the reference's gen-1 search scores
    non-terminal children from the opponent's side (search.py:33-40), which the engine reproduces."""
    from azsp import connect_state as cs
    random.seed(0)
    np.random.seed(0)
    strong = gen1.AlphaZeroPlayer(gen1.HashModel(1), mcts_simulation=200, temperature=0)
    finals = gen1.play_match(strong, gen1.RandomPlayer(), cs.Config().sample_initial_state(), 24)
    assert len(finals) == 24 and all(s.has_ended for s in finals)
    assert all(int(np.count_nonzero(s.grid == 0)) - int(np.count_nonzero(s.grid == 1)) in (0, 1) for s in finals)
    s = cs.Config().sample_initial_state()
    for col in (0, 1, 0, 1, 0, 1):                     # three in column 0 for player 0, three in column 1 for player 1
        s = [a for a in s.actions if a.column == col][0].sample_next_state()
    assert s.player == 0 and strong.play(s).column == 0
    assert [a.column for a in strong.play_many([s, s])] == [0, 0]
    soft = gen1.AlphaZeroPlayer(gen1.HashModel(1), mcts_simulation=30, temperature=1.0)
    assert soft.play(cs.Config().sample_initial_state()).column in range(7)
    assert gen1.AlphaZeroPlayer(gen1.HashModel(1), temperature=float("inf")).play(s).column in range(7)


# ------------------------------------------------------------------------------------ exploration factor (rule set 0)
@pytest.mark.parametrize("c_init,c_base", [(2.0, 0.0), (0.5, 0.0), (1.25, 19652.0), (1.25, 50.0)])
def test_exploration_factor_option_bit_exact(c_init, c_base):
    """c(N) in front of P*sqrt(N)/(n+1) -- constant, or the pb_c of the reference's
    deepmind_alphazero_pseudocode.py:252-255 -- is not in the reference's search (c = 1): pinned against the oracle's
    documented extension.  Default options (c = 1) are what every other test runs."""
    import azsp
    from helpers import HashStub
    n, sims, seed, salt = 10, 60, 3, 7
    rules = oz.C4Rules()
    np.random.seed(seed)
    ref = list(oz.generate_episodes(rules, oz.HashModel(rules, salt).predict, sims, n,
                                    c_of=oz.exploration_factor(c_init, c_base)))
    after = float(np.random.random_sample())
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                c_init=c_init, c_base=c_base)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == after
    assert len(eps) == n
    for ep, r in zip(eps, ref):
        assert [s.state.tolist() for s in ep.samples] == [np.asarray(s).tolist() for s, _, _ in r]
        assert [s.policy.tolist() for s in ep.samples] == [np.asarray(p).tolist() for _, p, _ in r]
        assert [s.value for s in ep.samples] == [int(z) for _, _, z in r]


class _FullMantissaModel:
    """The hash evaluator's values are multiples of 2^-15, whose sums are exact in float32 AND float64; scaling them
    by an odd constant gives values that use the whole float32 mantissa, so the two accumulator widths diverge."""

    def __init__(self, salt):
        self._m = oz.HashModel(oz.C4Rules(), salt)

    def predict(self, boards):
        p, v = self._m.predict(boards)
        return p, [np.float32(x) * np.float32(0.7310586) for x in v]


def test_numpy1_value_sum_variant_bit_exact():
    """AZSP_RULES_SIMPLE_NUMPY1: the alphazero_simple search with the float64 value sums numpy 1.x produces
    (PARITY UNPINNED -- oracle restatement only).  The run must also differ somewhere from the float32 run, or the
    variant would be vacuous."""
    import azsp
    from azsp.engine import Engine
    n, sims, seed, salt = 16, 120, 5, 11
    rules = oz.C4Rules()
    model = _FullMantissaModel(salt)
    np.random.seed(seed)
    tr_wide, tr_f32 = [], []
    ref = list(oz.generate_episodes(rules, model.predict, sims, n, numpy1=True, trace=tr_wide))
    np.random.seed(seed)
    list(oz.generate_episodes(rules, model.predict, sims, n, trace=tr_f32))
    assert any(a["root_w"] != b["root_w"] for a, b in zip(tr_wide, tr_f32))
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                numpy1_value_sums=True)
    eps = list(gen.generate_episodes(model))
    assert len(eps) == n
    for ep, r in zip(eps, ref):
        assert [s.state.tolist() for s in ep.samples] == [np.asarray(s).tolist() for s, _, _ in r]
        assert [s.policy.tolist() for s in ep.samples] == [np.asarray(p).tolist() for _, p, _ in r]
        assert [s.value for s in ep.samples] == [int(z) for _, _, z in r]
    # value sums themselves: one batched search compared in float64, and against the float32 rule set
    boards = [rules.init_board()] * 4
    roots = oz.run_batch(rules, model.predict, sims, boards, [1] * 4, numpy1=True)
    narrow = oz.run_batch(rules, model.predict, sims, boards, [1] * 4)
    for rules_id, want in ((2, roots), (0, narrow)):
        with Engine("connect4", 64, sims, search_rules=rules_id) as e:
            e.use_host_evaluator(model.predict)
            e.run_batch(np.stack(boards))
            val = e.root_values(4)
        for i, root in enumerate(want):
            assert val["root_w"][i] == float(root.w) and val["root_n"][i] == root.n
            for a, c in root.kids.items():
                assert val["child_w"][i, a] == float(c.w)
    assert any(float(a.w) != float(b.w) for a, b in zip(roots, narrow))


# ------------------------------------------------------------------------------------ tree reuse (option beyond the reference)
@pytest.mark.parametrize("n,sims,seed,extra", [(8, 30, 1, {}), (32, 100, 2, {}), (6, 3, 3, {}), (16, 60, 4, {"c_init": 1.4}),
                                               (12, 40, 5, {"dirichlet_alpha": 0.5, "dirichlet_eps": 0.25})])
def test_tree_reuse_option_bit_exact(n, sims, seed, extra):
    """AZSP_FLAG_TREE_REUSE: the chosen child's subtree (visit counts, value sums, priors, everything below) is
    the next ply's tree.  The reference rebuilds its trees (monte_carlo_tree_search.py:116-123), so this is pinned
    against the oracle's documented extension; slot restarts must start from a fresh tree, trees must survive
    across plies in place (node indices keep growing), and the run must differ from the non-reusing one."""
    import azsp
    from helpers import HashStub
    salt = 13
    rules = oz.C4Rules()
    okw = dict(extra)
    if "c_init" in okw:
        okw = {"c_of": oz.exploration_factor(okw["c_init"], 0.0)}
    np.random.seed(seed)
    tr = []
    ref = list(oz.generate_episodes(rules, oz.HashModel(rules, salt).predict, sims, n, reuse=True, trace=tr, **okw))
    after = float(np.random.random_sample())
    assert max(sum(t["counts"]) for t in tr) > sims            # reused roots carry visits over
    np.random.seed(seed)
    plain = list(oz.generate_episodes(rules, oz.HashModel(rules, salt).predict, sims, n, **okw))
    assert [[p.tolist() for _, p, _ in e] for e in plain] != [[p.tolist() for _, p, _ in e] for e in ref]
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                tree_reuse=True, **extra)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == after
    assert len(eps) == n
    for ep, r in zip(eps, ref):
        assert [s.state.tolist() for s in ep.samples] == [np.asarray(s).tolist() for s, _, _ in r]
        assert [s.policy.tolist() for s in ep.samples] == [np.asarray(p).tolist() for _, p, _ in r]
        assert [s.value for s in ep.samples] == [int(z) for _, _, z in r]
    # the same engine again: every slot starts from a fresh tree after reset()
    np.random.seed(seed)
    eps2 = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert [[s.policy.tolist() for s in e.samples] for e in eps2] == [[s.policy.tolist() for s in e.samples] for e in eps]


def test_gen1_dataset_tensors_equal_format_dataset_of_the_episodes(gen1):
    """gen1.generate_dataset (tensors written on the GPU) against the reference's own recipe applied to the yielded
    episodes: CNNModel._states_to_tensor (models/games/connect4/cnn.py:77-98), Connect4Model._policies_to_tensor
    (model.py:45-51), torch.FloatTensor(values) (models/base/model.py:80) -- restated here on the host."""
    import torch
    from azsp import connect_state as cs
    n, sims, seed = 12, 30, 8
    init = cs.Config().sample_initial_state()
    np.random.seed(seed)
    gen = gen1.EpisodeGenerator(model=gen1.HashModel(4), num_simulations=sims, num_episodes=n, game_initial_state=init)
    eps = list(gen.generate_episodes())
    samples = [s for ep in eps for s in ep.samples]
    planes = torch.zeros((len(samples), 3, 6, 7))
    pol = torch.zeros((len(samples), 7))
    for i, s in enumerate(samples):
        g = torch.tensor(np.asarray(s.state.grid))
        planes[i] = torch.stack([(g == -1).float(), (g == s.state.player).float(), (g == 1 - s.state.player).float()])
        for a, pr in s.policy.items():
            pol[i, a.column] = pr
    vals = torch.FloatTensor([s.value for s in samples])
    np.random.seed(seed)
    ds = gen1.generate_dataset(gen)
    got_planes, got_pol, got_vals = (t.cpu() for t in ds.tensors)
    assert torch.equal(got_planes, planes) and torch.equal(got_pol, pol) and torch.equal(got_vals, vals)
