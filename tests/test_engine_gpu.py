"""GPU parity tests: the HIP engine, called through the C ABI, against the oracle and the golden
fixtures generated from the reference.  Integer / index / float64 search results must be
BIT-EXACT; network outputs are compared with the tolerance stated in each test."""
import numpy as np
import pytest

from helpers import HashStub, HostOnlyModel, check_episode_invariants, load_json, random_midgame_boards
from oracle import az_oracle as oz
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def azsp():
    import azsp as pkg
    from azsp import _lib
    _lib.load()
    return pkg


def _engine(azsp, game, n, sims, **kw):
    from azsp.engine import Engine
    return Engine(game, n, sims, **kw)


# ------------------------------------------------------------------------------------ hash evaluator
def test_hash_evaluator_bit_exact(azsp):
    rng = np.random.RandomState(5)
    rules = oz.C4Rules()
    boards = random_midgame_boards(rules, rng, 300, 41)
    with _engine(azsp, "connect4", 128, 1, hash_salt=991) as e:
        e.use_hash_evaluator()
        p, v = e.predict(np.stack(boards))
    for b, pi, vi in zip(boards, p, v):
        po, vo = oz.hash_eval_bits(*rules.to_bits(b), rules.legal_mask_bits(b), 7, 991)
        assert np.array_equal(pi, po) and vi == vo


# ------------------------------------------------------------------------------------ search
def _check_run(out, run):
    A = out["counts"].shape[1]
    for i, g in enumerate(run["roots"]):
        assert out["root_n"][i] == g["n"]
        assert float(out["root_w"][i]) == g["w"]
        for a in range(A):
            k = g["kids"].get(str(a))
            if k is None:
                assert out["counts"][i, a] == 0 and out["prior"][i, a] == 0.0
            else:
                assert out["counts"][i, a] == k["n"], (i, a)
                assert float(out["child_w"][i, a]) == k["w"], (i, a)
                assert float(out["prior"][i, a]) == k["prior"], (i, a)


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
def test_run_batch_matches_reference_fixtures(azsp, cache):
    for run in load_json("search_hash.json")["runs"]:
        boards = np.array(run["boards"], dtype=np.int8)
        with _engine(azsp, run["game"], len(boards), run["sims"], hash_salt=run["salt"], eval_cache=cache) as e:
            e.use_hash_evaluator()
            _check_run(e.run_batch(boards), run)
            _check_run(e.run_batch(boards), run)        # again on the same engine (the cache now holds every position)


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
def test_mcts_facade_matches_reference_fixtures(azsp, cache):
    run = [r for r in load_json("search_hash.json")["runs"] if r["game"] == "connect4" and r["sims"] == 100][0]
    mcts = azsp.MCTS(azsp.Connect4Game(), HashStub("connect4", run["salt"]), run["sims"], eval_cache=cache)
    roots = mcts.run_batch([np.array(b) for b in run["boards"]], run["to_plays"])
    for root, g in zip(roots, run["roots"]):
        assert root.visit_count == g["n"] and float(root.value_sum) == g["w"]
        assert sorted(root.children) == sorted(int(a) for a in g["kids"])
        for a, k in g["kids"].items():
            c = root.children[int(a)]
            assert c.visit_count == k["n"] and float(c.value_sum) == k["w"] and float(c.prior) == k["prior"]
        assert root.select_action(0) == max(root.children, key=lambda a: (root.children[a].visit_count, -a))
    one = mcts.run(np.array(run["boards"][0]), 1)
    assert one.visit_count == run["roots"][0]["n"]
    single = mcts.run_single(np.array(run["boards"][3]), run["to_plays"][3])       # monte_carlo_tree_search.py:173-212
    g3 = run["roots"][3]
    assert single.visit_count == g3["n"] and float(single.value_sum) == g3["w"]
    assert {a: (c.visit_count, float(c.value_sum), float(c.prior)) for a, c in single.children.items()} == \
        {int(a): (k["n"], k["w"], k["prior"]) for a, k in g3["kids"].items()}


# ------------------------------------------------------------------------------------ episodes
def _episodes_equal(eps, golden):
    assert len(eps) == len(golden)
    for ep, g in zip(eps, golden):
        assert [s.state.tolist() for s in ep.samples] == g["states"]
        assert [s.policy.tolist() for s in ep.samples] == g["policies"]
        assert [s.value for s in ep.samples] == g["values"]
        assert all(type(s.value) is int for s in ep.samples)
        assert all(s.state.dtype == np.int64 and s.policy.dtype == np.float64 for s in ep.samples)


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
@pytest.mark.parametrize("model_cls", [HashStub, HostOnlyModel])
def test_generate_episodes_matches_reference_fixtures(azsp, model_cls, cache):
    """Drop-in EpisodeGenerator vs transcripts of the reference's generate_episodes: states, pi, z,
    yield order and the number of RNG draws, with the device evaluator and the host-callback path."""
    games = {"connect4": azsp.Connect4Game, "connect2": azsp.Connect2Game}
    for run in load_json("episodes_hash.json")["runs"]:
        if model_cls is HostOnlyModel and run["n"] * run["sims"] > 1000:
            continue            # python callback per round: keep the slow path small
        cfg = azsp.AlphaZeroConfig(batch_size=32, num_iterations=1, num_simulations=run["sims"],
                                   num_episodes=run["n"], num_iters_for_train_history=1, epochs=1)
        gen = azsp.EpisodeGenerator(game=games[run["game"]](), config=cfg, eval_cache=cache)     # (host callback: runs without it)
        np.random.seed(run["seed"])
        eps = list(gen.generate_episodes(model_cls(run["game"], run["salt"])))
        assert float(np.random.random_sample()) == run["rng_after"]
        _episodes_equal(eps, run["episodes"])
        st = gen.last_stats
        if cache and model_cls is HashStub:
            assert st["evals_executed"] < st["evals"] or st["evals"] < 4
        else:
            assert st["evals_executed"] == st["evals"]


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
def test_per_ply_root_counts_match_reference_fixtures(azsp, cache):
    run = [r for r in load_json("episodes_hash.json")["runs"] if r["game"] == "connect4" and r["n"] == 16][0]
    np.random.seed(run["seed"])
    with _engine(azsp, "connect4", run["n"], run["sims"], hash_salt=run["salt"], eval_cache=cache) as e:
        e.use_hash_evaluator()
        e.reset()
        for ply in run["plies"]:
            e.search()
            counts = e.root_counts()
            used = len(ply["actions"])
            assert counts[:used].tolist() == ply["counts"][:used]
            res = e.play(np.random.random_sample(run["n"]))
            assert res.uniforms_used == used
        assert res.finished and res.episodes_total == run["n"]


@pytest.mark.parametrize("n,sims,seed,salt,groups", [(512, 100, 11, 3, 1), (512, 100, 11, 3, 4), (2048, 30, 15, 2, 0),
                                                     (64, 400, 12, 8, 0), (1, 100, 13, 0, 0), (3, 1, 14, 1, 0)])
@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
def test_generate_matches_c_oracle_at_larger_sizes(azsp, n, sims, seed, salt, groups, cache):
    """Also covers the multi-stream slot groups: results must not depend on the grouping."""
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    cfg = azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=cfg, n_groups=groups, eval_cache=cache)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    assert len(eps) == n
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()
    st = gen.last_stats
    assert st["evals"] == ref["stats"][0] and st["terminal_leaves"] == ref["stats"][1]      # the reference's evaluation count
    assert st["path_nodes"] == ref["stats"][3]
    assert st["evals_executed"] == st["evals"] if not cache else st["evals_executed"] <= st["evals"]


# ------------------------------------------------------------------------------------ network
# Absolute bounds on |p - p_ref| and |v - v_ref| against a float32 torch-CPU forward of the same weights.  north_star
# asks for 1e-6 (NORTH_STAR_TOL): that is asserted for BASELINE's three networks with the random weights BASELINE
# names (default initialisation), test_baseline_networks_within_north_star_tolerance.  The kernels are otherwise held
# to what they measure (printed as [net-error] lines; round-2 run: <= 3.9e-7 everywhere but one case): NET_TOL for the
# reference's own vectors and every geometry the kernels are instantiated for, NET_TOL_WIDE only for the 8 x 128 stress
# net (randomised BatchNorm statistics let activations grow through 16 layers: 1.0e-6 on v, against 1.7e-7 for torch's
# own float32 forward relative to float64 -- the kernel sums each output's K = 9 * C products in one float32 chain).
NET_TOL = 5e-7
NET_TOL_WIDE = 2e-6
NORTH_STAR_TOL = 1e-6


def _torch_reference(net, rules, boards, double=False):
    """Masked softmax / tanh of the module's own PyTorch forward on the CPU (float32, or float64 as a yardstick)."""
    import copy
    import torch
    m = copy.deepcopy(net).double() if double else net
    with torch.no_grad():
        x = net._states_to_tensor(boards)
        logits, vl = m.forward(x.double() if double else x)
        mask = torch.tensor(np.array([rules.valid_moves(b) for b in boards]))
        pt = torch.softmax(logits.masked_fill(mask == 0, float("-inf")), dim=1).numpy()
        vt = torch.tanh(vl).numpy()
    return pt, vt, mask.numpy()


def _golden_net(tag):
    z = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "resnet_forward.npz"))
    sd = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(f"{tag}_sd/")}
    return z, sd


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
@pytest.mark.parametrize("tag", ["s", "m"])
def test_resnet_forward_matches_reference_vectors(azsp, tag, f16):
    """Fused MFMA tower vs ResNet.predict of the reference on torch-CPU.  fp32 everywhere, different
    summation order and BN folding -> tolerance NET_TOL (5e-7, below north_star's 1e-6) absolute on probabilities
    and tanh values; the measured error is printed."""
    from azsp.weights import resnet_blob
    z, sd = _golden_net(tag)
    blob, blocks, ch = resnet_blob({k: v for k, v in sd.items() if "num_batches_tracked" not in k})
    assert [blocks, ch] == z[f"{tag}_arch"].tolist()
    boards = z["boards"]
    with _engine(azsp, "connect4", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
        e.load_resnet(blob)
        p, v = e.predict(boards)
        p2, v2 = e.predict(boards[::-1].copy())            # batch-invariance: other order, other tiles
    assert np.array_equal(p2[::-1], p) and np.array_equal(v2[::-1], v)
    assert np.all(p[z[f"{tag}_p"] == 0] == 0)               # exact zeros on illegal columns
    err_p, err_v = np.max(np.abs(p - z[f"{tag}_p"])), np.max(np.abs(v - z[f"{tag}_v"]))
    print(f"[net-error] reference vectors {tag} ({blocks}x{ch}) f16x2={f16}: max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e}")
    assert err_p < NET_TOL and err_v < NET_TOL
    assert np.all(np.abs(p.sum(axis=1) - 1) < 1e-6)


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_resnet_self_play_bit_exact_vs_oracle_fed_by_device_net(azsp, f16, cache):
    """End-to-end with the real network: the C oracle runs the reference algorithm on the CPU but asks
    the DEVICE kernel for (P, V) through the C ABI; the engine runs everything on the GPU.  Both see
    identical floats, so visit counts, moves, pi and z must be bit-identical."""
    from azsp.weights import resnet_blob
    _, sd = _golden_net("m")
    blob, blocks, ch = resnet_blob({k: v for k, v in sd.items() if "num_batches_tracked" not in k})
    n, sims, seed = 32, 50, 21
    with _engine(azsp, "connect4", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (6, 7), 7)
        ref = co.generate("connect4", sims, n, seed=seed, predict_cb=cb, trace=False)

    class Net:
        def state_dict(self):
            return {k: v for k, v in sd.items() if "num_batches_tracked" not in k}

    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                f16_split=f16, eval_cache=cache)
    eps = list(gen.generate_episodes(Net()))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
@pytest.mark.parametrize("blocks,ch", [(2, 32), (1, 16), (3, 48), (2, 128), (8, 128), (1, 100), (4, 64), (2, 96), (1, 160),
                                       (2, 192), (2, 256)])
def test_resnet_model_predict_uses_device_kernel(azsp, blocks, ch, f16, monkeypatch):
    """azsp.ResNet.predict (HIP kernel) vs the same module's PyTorch forward on the CPU, for every
    channel-tile count the kernel is instantiated for (C padded to 32/64/128).  Tolerance NET_TOL for the BASELINE
    geometries, NET_TOL_WIDE otherwise; also printed: both against a float64 forward."""
    import torch
    monkeypatch.setenv("AZSP_F16_SPLIT", "1" if f16 else "0")
    torch.manual_seed(blocks * 1000 + ch)
    net = azsp.ResNet((6, 7), 7, blocks, ch)
    with torch.no_grad():                          # non-trivial BatchNorm statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    net.eval()
    rules = oz.C4Rules()
    boards = random_midgame_boards(rules, np.random.RandomState(2), 20, 40)
    p, v = net.predict(boards)
    pt, vt, _ = _torch_reference(net, rules, boards)
    p64, v64, _ = _torch_reference(net, rules, boards, double=True)
    err_p, err_v = np.max(np.abs(np.stack(p) - pt)), np.max(np.abs(np.array(v) - vt))
    print(f"[net-error] connect4 {blocks}x{ch} f16x2={f16}: max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e} vs torch f32; "
          f"vs float64: kernel {np.max(np.abs(np.stack(p) - p64)):.2e} / {np.max(np.abs(np.array(v) - v64)):.2e}, "
          f"torch f32 {np.max(np.abs(pt - p64)):.2e} / {np.max(np.abs(vt - v64)):.2e}")
    tol = NET_TOL_WIDE if (blocks, ch) == (8, 128) else NET_TOL     # the deep stress net, see the note on the bounds
    assert err_p < tol and err_v < tol
    clone = net.get_inference_clone()
    assert clone._azsp_engine is None and net._azsp_engine is not None
    with torch.no_grad():                          # weights change -> next predict re-uploads
        net.policy_head[4].bias[0] += 1.0         # (a uniform shift would leave the softmax unchanged)
    p2, _ = net.predict(boards)
    assert not np.allclose(np.stack(p2), np.stack(p))


@pytest.mark.parametrize("game,blocks,ch", [("connect4", 4, 64), ("connect4", 8, 128), ("gomoku9", 10, 256)])
def test_baseline_networks_within_north_star_tolerance(azsp, game, blocks, ch):
    """BASELINE.json's three networks with the weights its configs name ("random weights" = PyTorch's default
    initialisation, torch.manual_seed(0) as in bench.py): p and tanh(v) of the device kernel (exact fp32 and split
    fp16 where that kernel exists) within north_star's 1e-6 of the module's float32 PyTorch forward."""
    import torch
    torch.manual_seed(0)
    shape, A, rules = ((6, 7), 7, oz.C4Rules()) if game == "connect4" else ((9, 9), 81, oz.G9Rules())
    net = azsp.ResNet(shape, A, blocks, ch).eval()
    boards = random_midgame_boards(rules, np.random.RandomState(12), 48, 40)
    pt, vt, mask = _torch_reference(net, rules, boards)
    from azsp.weights import resnet_blob
    blob = resnet_blob(net)[0]
    for f16 in (False, True):
        with _engine(azsp, game, 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
            e.load_resnet(blob)
            p, v = e.predict(np.stack(boards))
        err_p, err_v = np.max(np.abs(p - pt)), np.max(np.abs(v - vt))
        print(f"[net-error] BASELINE net {game} {blocks}x{ch} f16x2={f16}: max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e}")
        assert err_p < NORTH_STAR_TOL and err_v < NORTH_STAR_TOL
        assert np.all(p[mask == 0] == 0)


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
@pytest.mark.parametrize("blocks,ch", [(2, 32), (2, 64), (2, 96), (3, 128), (1, 192), (2, 256)])
def test_gomoku_network_geometries_against_torch(azsp, blocks, ch, f16):
    """Every Gomoku kernel geometry -- three boards per workgroup up to 64 channels; from 96 on one board per workgroup on
    five 16-row MFMA tiles plus the corner cell on the vector ALU (chunks of 6 or 8 sixteen-channel groups), Linear layer in
    channel slices, value conv in channel slices, softmax on a wave; the split-fp16 kernel's one-board form on 96 GEMM rows
    with one to eight column tiles -- against the module's float32 PyTorch forward, with non-trivial BatchNorm statistics;
    batch sizes that leave workgroups partly filled and empty."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(100 * blocks + ch)
    net = azsp.ResNet((9, 9), 81, blocks, ch)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    net.eval()
    rules = oz.G9Rules()
    boards = random_midgame_boards(rules, np.random.RandomState(ch), 40, 60)
    boards.append(np.zeros((9, 9), np.int8))                      # the empty board: every tap of every cell multiplies zeros
    full = np.ones((9, 9), np.int8); full[::2, 1::2] = -1; full[8, 8] = 0
    boards.append(full)                                           # one legal move: the corner cell
    pt, vt, mask = _torch_reference(net, rules, boards)
    blob = resnet_blob(net)[0]
    with _engine(azsp, "gomoku9", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
        e.load_resnet(blob)
        p, v = e.predict(np.stack(boards))
        p1, v1 = e.predict(np.stack(boards[:7]))                  # bits do not depend on the batch
    err_p, err_v = np.max(np.abs(p - pt)), np.max(np.abs(v - vt))
    print(f"[net-error] gomoku9 {blocks}x{ch} f16x2={f16}: max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e} vs torch f32")
    assert err_p < NET_TOL_WIDE and err_v < NET_TOL_WIDE
    assert np.all(p[mask == 0] == 0) and p[-1][80] == 1.0
    assert np.array_equal(p[:7], p1) and np.array_equal(v[:7], v1)


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
@pytest.mark.parametrize("blocks,ch", [(4, 64), (2, 128), (2, 32), (1, 192), (2, 256)])
def test_resnet_every_batch_regime_returns_the_same_bits(azsp, blocks, ch, f16):
    """The engine picks a kernel variant by batch size (<= 768 boards: 3 boards per workgroup; above: 6 boards per
    workgroup with the rows in tap-skipping order, other wave grids).  A position's (P, V) must not depend on it:
    the same boards through a small batch, a large batch and a large batch in another order are compared bitwise
    (this is what makes the tap-skipping order and the MFMA value head testable: they are exact rearrangements).
    192 / 256 channels: the split-fp16 kernel's one-board form (64 rows, ring pipeline) against its three-board form."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(ch + blocks)
    net = azsp.ResNet((6, 7), 7, blocks, ch).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    blob = resnet_blob(net)[0]
    rules = oz.C4Rules()
    rng = np.random.RandomState(7)
    boards = np.stack(random_midgame_boards(rules, rng, 1500, 41))
    with _engine(azsp, "connect4", 1500, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
        e.load_resnet(blob)
        p_big, v_big = e.predict(boards)                      # one 1500-board launch: large-batch variant
        perm = rng.permutation(1500)
        p_perm, v_perm = e.predict(boards[perm])              # other tiles, other rows of the row order
        p_odd, v_odd = e.predict(boards[:1003])               # a partly filled last workgroup
    with _engine(azsp, "connect4", 100, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
        e.load_resnet(blob)
        p_small, v_small = e.predict(boards)                  # 100-board launches: one board per workgroup
    with _engine(azsp, "connect4", 700, 1, res_blocks=blocks, channels=ch, f16_split=f16) as e:
        e.load_resnet(blob)
        p_mid, v_mid = e.predict(boards)                      # 700-board launches: three boards per workgroup
    assert np.array_equal(p_mid, p_big) and np.array_equal(v_mid, v_big)
    assert np.array_equal(p_perm, p_big[perm]) and np.array_equal(v_perm, v_big[perm])
    assert np.array_equal(p_odd, p_big[:1003]) and np.array_equal(v_odd, v_big[:1003])
    assert np.array_equal(p_small, p_big) and np.array_equal(v_small, v_big)
    pt, vt, _ = _torch_reference(net, rules, list(boards[:64]))    # and all of it is the network (tolerance as elsewhere)
    err_p, err_v = np.max(np.abs(p_big[:64] - pt)), np.max(np.abs(v_big[:64] - vt))
    print(f"[net-error] connect4 {blocks}x{ch} f16x2={f16} (all batch regimes): max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e}")
    tol = NET_TOL if ch <= 128 else NET_TOL_WIDE
    assert err_p < tol and err_v < tol


# ------------------------------------------------------------------------------------ full size
@pytest.mark.parametrize("f16,cache", [(False, False), (True, False), (False, True), (True, True)],
                         ids=["f32", "f16x2", "f32-evalcache", "f16x2-evalcache"])
def test_full_size_config2_properties(azsp, f16, cache):
    """BASELINE.json configs[1]: Connect-4, 64-ch/4-block ResNet, 100 sims, 4096 concurrent games.
    Too large for the oracle; checked through size-independent invariants of every episode (both inference kernels,
    with and without the evaluation cache)."""
    import torch
    torch.manual_seed(0)
    net = azsp.ResNet((6, 7), 7, 4, 64)
    n, sims = 4096, 100
    np.random.seed(0)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16,
                                eval_cache=cache)
    eps = list(gen.generate_episodes(net))
    assert len(eps) == n
    rules = oz.C4Rules()
    for ep in eps[::7]:
        check_episode_invariants(rules, np.stack([s.state for s in ep.samples]),
                                 np.stack([s.policy for s in ep.samples]), [s.value for s in ep.samples], sims)
    st = gen.last_stats
    assert st["rounds"] == st["plies"] * (sims + 1)
    assert st["evals"] + st["terminal_leaves"] == st["plies"] * n * (sims + 1)


def _full_size_invariants(azsp, game, rules, net, n, sims, stride, **gen_kw):
    np.random.seed(0)
    gen = azsp.EpisodeGenerator(game=game, config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), **gen_kw)
    eps = list(gen.generate_episodes(net))
    assert len(eps) == n
    for ep in eps[::stride]:
        check_episode_invariants(rules, np.stack([s.state for s in ep.samples]),
                                 np.stack([s.policy for s in ep.samples]), [s.value for s in ep.samples], sims)
    st = gen.last_stats
    assert st["rounds"] == st["plies"] * (sims + 1)
    assert st["evals"] + st["terminal_leaves"] == st["plies"] * n * (sims + 1)
    return eps, st


@pytest.mark.parametrize("f16,cache", [(False, False), (True, False), (True, True)], ids=["f32", "f16x2", "f16x2-evalcache"])
def test_full_size_config3_properties(azsp, f16, cache):
    """BASELINE.json configs[2]: Connect-4, 64-ch/4-block ResNet, 400 sims, 8192 concurrent games on one GPU (the
    deep-tree backup stress: up to 2808 nodes per tree, 0.46 GB of trees), through the same invariants as config 2."""
    import torch
    torch.manual_seed(0)
    net = azsp.ResNet((6, 7), 7, 4, 64)
    eps, st = _full_size_invariants(azsp, azsp.Connect4Game(), oz.C4Rules(), net, 8192, 400, 23, f16_split=f16, eval_cache=cache)
    assert st["path_nodes"] / (st["evals"] + st["terminal_leaves"]) > 3.0      # deeper trees than at 100 sims


def test_full_size_config4_one_gpu_shard_properties(azsp):
    """BASELINE.json configs[3], the share of ONE of its 8 GPUs: Connect-4, 128-ch/8-block ResNet, 100 sims, 4096
    concurrent games (32768 games / 8; the shards are independent replicas)."""
    import torch
    torch.manual_seed(0)
    net = azsp.ResNet((6, 7), 7, 8, 128)
    _full_size_invariants(azsp, azsp.Connect4Game(), oz.C4Rules(), net, 4096, 100, 11)


@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_config4_network_self_play_bit_exact_vs_oracle_fed_by_device_net(azsp, f16, cache):
    """configs[3]'s network (8 blocks x 128 channels) in the loop, at a size the oracle finishes: 32 games x 100 sims;
    the C oracle asks the device kernel (fp32 or f16x2) for (P, V), so counts, moves, pi, z and the RNG position must be
    identical -- with and without the evaluation cache."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(3)
    net = azsp.ResNet((6, 7), 7, 8, 128).eval()
    blob, blocks, ch = resnet_blob(net)
    n, sims, seed = 32, 100, 5
    with _engine(azsp, "connect4", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (6, 7), 7)
        ref = co.generate("connect4", sims, n, seed=seed, predict_cb=cb, trace=False)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16,
                                eval_cache=cache)
    eps = list(gen.generate_episodes(net))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    _assert_equals_c_oracle(eps, ref, n)


def _assert_equals_c_oracle(eps, ref, n):
    assert len(eps) == n == ref["n_episodes"]
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_full_size_config5_one_gpu_shard_properties(azsp, f16):
    """BASELINE.json configs[4] on one GPU: 9x9 Gomoku through the Game plugin, 256-ch/10-block ResNet, 200 sims/move;
    192 concurrent games in two slot groups (a shard sized for the test's time budget), checked through the
    size-independent invariants (pi * sims integral, move chains, outcome / z consistency, evaluation accounting);
    both inference kernels."""
    import torch
    torch.manual_seed(0)
    net = azsp.ResNet((9, 9), 81, 10, 256)
    _full_size_invariants(azsp, azsp.GomokuGame(), oz.G9Rules(), net, 192, 200, 5, n_groups=2, f16_split=f16)


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_config5_network_self_play_bit_exact_vs_oracle_fed_by_device_net(azsp, f16):
    """configs[4]'s network (10 blocks x 256 channels, 9x9) in the loop at oracle size: 6 games x 24 sims (exact fp32 and
    the split-fp16 kernel: the oracle is fed by the same device kernel, so the search must agree bit for bit)."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(9)
    net = azsp.ResNet((9, 9), 81, 10, 256).eval()
    blob, blocks, ch = resnet_blob(net)
    n, sims, seed = 6, 24, 8
    with _engine(azsp, "gomoku9", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (9, 9), 81)
        ref = co.generate("gomoku9", sims, n, seed=seed, predict_cb=cb, trace=False)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16)
    eps = list(gen.generate_episodes(net))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    _assert_equals_c_oracle(eps, ref, n)


@pytest.mark.parametrize("blocks,ch", [(2, 32), (1, 128)])
def test_gomoku_device_network_with_several_slot_groups(azsp, blocks, ch):
    """Slot groups run on separate streams and share the evaluator's output array, one stripe of n_act rows of the
    game's padded action width (88 floats for Gomoku, 8 for Connect-4) per group.  128 games of Gomoku with the device
    network must come out identical with 1, 2 and 4 groups and equal to the C oracle fed by the same kernel."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(ch)
    net = azsp.ResNet((9, 9), 81, blocks, ch).eval()
    blob, _, _ = resnet_blob(net)
    n, sims, seed = 128, 12, 3
    with _engine(azsp, "gomoku9", 256, 1, res_blocks=blocks, channels=ch) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (9, 9), 81)
        ref = co.generate("gomoku9", sims, n, seed=seed, predict_cb=cb, trace=False)
    boards = np.stack(random_midgame_boards(oz.G9Rules(), np.random.RandomState(5), 128, 50))
    searched = []
    for groups in (1, 2, 4):
        np.random.seed(seed)
        gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), n_groups=groups)
        eps = list(gen.generate_episodes(net))
        assert float(np.random.random_sample()) == ref["next_uniform"], groups
        _assert_equals_c_oracle(eps, ref, n)
        with _engine(azsp, "gomoku9", 128, 40, res_blocks=blocks, channels=ch, n_groups=groups) as e:
            e.load_resnet(blob)
            searched.append(e.run_batch(boards))
    for other in searched[1:]:
        for k in ("counts", "child_w", "prior", "root_w", "root_n"):
            assert np.array_equal(searched[0][k], other[k]), k


@pytest.mark.parametrize("f16,cache", [(False, False), (True, False), (False, True), (True, True)],
                         ids=["f32", "f16x2", "f32-evalcache", "f16x2-evalcache"])
def test_real_network_transcript_of_the_reference_config1(azsp, f16, cache):
    """tests/golden/episodes_resnet.json: the REFERENCE's generate_episodes with its own torch-CPU ResNet(4, 64)
    (BASELINE.json configs[0]: 25 sims, one game, seeds 0 / 0; episode_generator.py:23-87 + resnet.py:130-157).  The
    engine plays the same game with the same weights on the device network: per-ply root visit counts, chosen moves,
    pi, z and the RNG position must be the reference's.  The two networks agree to ~1e-7, not bitwise, so a ply may
    legitimately differ only where the search passed through a near-tie: everything up to such a ply must be equal,
    and the PUCT margin there (oracle search fed by the device network) must be below 1e-6 -- it is printed."""
    from helpers import min_puct_margin
    from azsp.weights import resnet_blob
    run = load_json("episodes_resnet.json")
    z, sd = _golden_net("m")               # ResNet((6, 7), 7, 4, 64) after torch.manual_seed(0): the transcript's network
    assert z["m_arch"].tolist() == run["arch"] and run["torch_seed"] == 0
    blob, blocks, ch = resnet_blob({k: v for k, v in sd.items() if "num_batches_tracked" not in k})
    n, sims = run["n"], run["sims"]
    rules = oz.C4Rules()
    np.random.seed(run["seed"])
    states, pols = [], []
    with _engine(azsp, "connect4", n, sims, res_blocks=blocks, channels=ch, f16_split=f16, eval_cache=cache) as e:
        e.load_resnet(blob)
        e.reset()
        board, player = rules.init_board(), 1
        for k, ply in enumerate(run["plies"]):
            e.search()
            counts = e.root_counts()[0].tolist()
            if counts != ply["counts"][0]:
                with _engine(azsp, "connect4", 64, 1, res_blocks=blocks, channels=ch, f16_split=f16) as pe:
                    pe.load_resnet(blob)
                    gap = min_puct_margin(rules, lambda bs: tuple(map(list, pe.predict(np.stack(bs)))), sims,
                                          rules.canonical(board, player))
                print(f"[transcript] ply {k}: device-network counts {counts} vs reference {ply['counts'][0]}; "
                      f"smallest PUCT margin in this search = {gap:.3e}")
                assert gap < 1e-6, "the searches differ although no selection step was a near-tie"
                pytest.xfail(f"transcript equal up to ply {k}; ply {k} passes through a PUCT near-tie ({gap:.1e})")
            states.append(rules.canonical(board, player).tolist())
            pols.append((np.array(counts) / np.sum(counts)).tolist())
            res = e.play(np.random.random_sample(n))
            out = e.fetch_episodes(0, res.n_finished) if res.n_finished else None
            board, player = rules.next_state(board, player, ply["actions"][0])
        assert res.finished and res.episodes_total == 1 and len(run["plies"]) == len(states)
    g = run["episodes"][0]
    assert states == g["states"] and pols == g["policies"]
    assert out["states"].reshape(-1, 6, 7).tolist() == g["states"] and out["policies"].tolist() == g["policies"]
    assert out["values"].tolist() == g["values"]
    assert float(np.random.random_sample()) == run["rng_after"]
    # and through the drop-in generator, as the training loop would call it
    class Net:
        def state_dict(self):
            return {k: v for k, v in sd.items() if "num_batches_tracked" not in k}
    np.random.seed(run["seed"])
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16,
                                eval_cache=cache)
    _episodes_equal(list(gen.generate_episodes(Net())), run["episodes"])
    assert float(np.random.random_sample()) == run["rng_after"]


def test_device_dataset_and_json_writer_match_reference_output(azsp, tmp_path):
    """SURVEY 8f rank 1 and 2 pinned by the reference's own output: tests/golden/format_dataset.npz holds the tensors
    the reference's BaseModel.format_dataset / ResNet._states_to_tensor built (base_model.py:61-67, resnet.py:115-128)
    and tests/golden/episodes_wire.json the file its DataModule._save_episodes wrote (episode.py:15-24,64-66,
    datamodule.py:87-96) for one of its generator runs.  The engine repeats that run (hash evaluator, same seed):
    azsp_export_dataset_device must give those tensors and save_episodes that file, bit for bit / byte for byte."""
    import json
    import os
    import torch
    from helpers import GOLDEN
    from azsp.replay import generate_dataset, load_episodes, save_episodes
    z = np.load(os.path.join(GOLDEN, "format_dataset.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    text = open(os.path.join(GOLDEN, "episodes_wire.json")).read()
    cfg = azsp.AlphaZeroConfig(32, 1, meta["sims"], meta["n"], 1, 1)
    model = HashStub(meta["game"], meta["salt"])
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=cfg)
    np.random.seed(meta["seed"])
    ds = generate_dataset(gen, model)
    assert float(np.random.random_sample()) == meta["rng_after"]
    assert [str(t.dtype) for t in ds.tensors] == meta["dtypes"] and [list(t.shape) for t in ds.tensors] == meta["shapes"]
    for t, k in zip(ds.tensors, ("planes", "policies", "values")):
        assert t.is_cuda and torch.equal(t.cpu(), torch.from_numpy(z[k])), k
    np.random.seed(meta["seed"])
    eps = list(gen.generate_episodes(model))
    save_episodes(tmp_path / "episodes_iter1.json", eps)
    assert open(tmp_path / "episodes_iter1.json").read() == text
    assert load_episodes(os.path.join(GOLDEN, "episodes_wire.json")) == eps


def test_small_batches_run_fused_rounds_by_default(azsp, monkeypatch):
    """A few games (up to 32: one-board inference kernels) run ONE launch per simulation round unless the diagnostics
    switch AZSP_DIAG=no_fused_rounds is set: a round there is launch latency.  Same episodes, RNG position and evaluation accounting as the two-launch rounds, and
    bit-exact against the C oracle fed by the device network; large batches keep two launches per round by default."""
    import torch
    from azsp.weights import resnet_blob
    monkeypatch.delenv("AZSP_DIAG", raising=False)
    torch.manual_seed(5)
    net = azsp.ResNet((6, 7), 7, 2, 64).eval()
    blob, blocks, ch = resnet_blob(net)
    n, sims, seed = 12, 30, 21
    with _engine(azsp, "connect4", n, 1, res_blocks=blocks, channels=ch) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (6, 7), 7)
        ref = co.generate("connect4", sims, n, seed=seed, predict_cb=cb, trace=False)

    def run(env):
        if env is None:
            monkeypatch.delenv("AZSP_DIAG", raising=False)
        else:
            monkeypatch.setenv("AZSP_DIAG", env)
        np.random.seed(seed)
        g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1))
        out = list(g.generate_episodes(net))
        return out, g.last_stats, float(np.random.random_sample())
    a, sa, ra = run(None)
    b, sb, rb = run("no_fused_rounds")
    assert sa["select_launches"] == 0 and sa["infer_launches"] == sa["plies"] * (sims + 1)      # one launch per round
    assert sb["select_launches"] > 0
    assert ra == rb == ref["next_uniform"] and a == b
    _assert_equals_c_oracle(a, ref, n)
    assert (sa["evals"], sa["terminal_leaves"], sa["path_nodes"]) == (sb["evals"], sb["terminal_leaves"], sb["path_nodes"])
    assert sa["evals"] == ref["stats"][0] and sa["terminal_leaves"] == ref["stats"][1]
    monkeypatch.delenv("AZSP_DIAG", raising=False)
    np.random.seed(3)                                        # a large batch: two launches per round
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, 5, 3000, 1, 1))
    big = list(g.generate_episodes(net))
    assert g.last_stats["select_launches"] > 0
    # the other diagnostics switch: device evaluators on the host evaluator's four-launch slot-ordered rounds
    monkeypatch.setenv("AZSP_DIAG", "unfused_tree")
    np.random.seed(3)
    g4 = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, 5, 3000, 1, 1))
    assert list(g4.generate_episodes(net)) == big
    assert g4.last_stats["expand_launches"] > g.last_stats["expand_launches"]


def test_error_paths(azsp):
    from azsp._lib import AzspError
    with pytest.raises(AzspError):
        _engine(azsp, "connect4", 0, 10)
    with _engine(azsp, "connect4", 4, 5) as e:
        with pytest.raises(AzspError):
            e.play(np.zeros(4))                 # play before search
        with pytest.raises(AzspError):
            e.load_resnet(np.zeros(10, np.float32))
        with pytest.raises(ValueError):
            e.search() or e.play(np.zeros(3))
        boom = RuntimeError("predict failed")

        def bad(_):
            raise boom
        e.use_host_evaluator(bad)
        with pytest.raises(RuntimeError, match="predict failed"):
            e.search()
    with pytest.raises(NotImplementedError):
        class Other:
            pass
        _engine(azsp, Other(), 4, 5)


def test_device_record_export_matches_host_fetch(azsp):
    """azsp_export_records_device (the payload of the multi-GPU all-gather) == the episodes fetched to
    the host, record for record."""
    import torch
    from azsp.distributed import export_records, unpack_records
    n, sims = 256, 20
    np.random.seed(3)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1))
    eps = list(gen.generate_episodes(HashStub("connect4", 4)))
    rec = export_records(gen).cpu().numpy()
    s, p, v = unpack_records(rec, 42, 7)
    assert rec.shape == (sum(len(e) for e in eps), 104)
    k = 0
    for ep in eps:
        for smp in ep.samples:
            assert np.array_equal(s[k].reshape(6, 7), smp.state) and np.array_equal(p[k], smp.policy) and v[k] == smp.value
            k += 1
    assert torch.cuda.is_available()


def test_device_dataset_equals_format_dataset_of_yielded_episodes(azsp):
    """generate_dataset (tensors built on the GPU) == model.format_dataset applied, as the reference's
    DataModule does (datamodule.py:134-142), to the episodes the generator yields under the same seed."""
    import torch
    from azsp.replay import DeviceReplayBuffer, generate_dataset
    n, sims = 128, 30
    cfg = azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1)
    model = HashStub("connect4", 9)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=cfg)
    np.random.seed(4)
    eps = list(gen.generate_episodes(model))
    after = float(np.random.random_sample())
    samples = [s for e in eps for s in e.samples]
    host_ds = azsp.ResNet((6, 7), 7, 1, 32).format_dataset([s.state for s in samples], [s.policy for s in samples],
                                                           [s.value for s in samples])
    np.random.seed(4)
    dev_ds = generate_dataset(gen, model)
    assert float(np.random.random_sample()) == after
    for h, d in zip(host_ds.tensors, dev_ds.tensors):
        assert d.is_cuda and d.dtype == torch.float32 and torch.equal(h, d.cpu())
    buf = DeviceReplayBuffer(history=2)
    for _ in range(3):
        buf.extend(dev_ds)
    assert len(buf) == 2 * len(dev_ds) and buf.dataset().tensors[0].shape[0] == 2 * len(dev_ds)


@pytest.mark.parametrize("shards,n,sims", [(2, 96, 30), (3, 100, 25), (4, 64, 40)])
def test_sharded_exact_mode_equals_one_sequential_reference_run(azsp, shards, n, sims):
    """SURVEY 8e mode A: ONE generator run of n slots cut over several engines (here: several engines on the
    one GPU of the test box) with per-ply exchange of completion counts must be bit-identical -- episodes,
    their global order, pi, z and the RNG position -- to the sequential run of the same n slots (C oracle)."""
    seed, salt = 31 + shards, 6
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    gen = azsp.ShardedEpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                       devices=[0] * shards)
    np.random.seed(seed)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    gen.close()
    assert float(np.random.random_sample()) == ref["next_uniform"]
    assert len(eps) == n
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()


def _dist_worker(rank, world, port, n, sims, seed, salt, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "alphazero-implementation_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import azsp
    from helpers import HashStub as Stub
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = azsp.DistributedEpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1),
                                               dist=dist, device=0, coll_device=torch.device("cpu"))
        np.random.seed(seed)                       # identical RNG on every rank
        out = [(ply, slot, [s.state.tolist() for s in ep.samples], [s.policy.tolist() for s in ep.samples],
                [s.value for s in ep.samples]) for ep, ply, slot in gen.generate_episodes(Stub("connect4", salt))]
        q.put((rank, out, float(np.random.random_sample())))
    except BaseException as exc:       # noqa: BLE001  report instead of leaving the parent waiting
        q.put((rank, repr(exc), None))
        raise
    finally:
        dist.destroy_process_group()


def test_distributed_exact_mode_two_ranks_gloo_rehearsal(azsp):
    """DistributedEpisodeGenerator with 2 ranks (both on this box's one GPU, gloo for the tiny per-ply count
    exchange): the union of the ranks' episodes, merged by (ply, global slot), is the sequential reference run."""
    import socket
    import torch.multiprocessing as mp
    n, sims, seed, salt = 48, 20, 77, 2
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, n, sims, seed, salt, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(after is not None for _, _, after in got), got
    assert all(p.exitcode == 0 for p in procs)
    assert all(after == ref["next_uniform"] for _, _, after in got)      # every rank's RNG ends at the reference position
    merged = sorted((e for _, eps, _ in got for e in eps), key=lambda t: (t[0], t[1]))
    assert len(merged) == n
    for e, (ply, slot, states, pols, vals) in enumerate(merged):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert ply == ref["ep_ply"][e] and slot == ref["ep_slot"][e]
        assert np.array_equal(np.array(states).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.array(pols), ref["policies"][lo:hi]) and vals == ref["values"][lo:hi].tolist()


def test_bench_gpus2_launches_its_own_two_ranks_on_this_box(azsp):
    """`python bench.py --gpus 2` with no launcher: bench.py starts its own two rank processes (one torch.distributed.run
    child) and rank 0 prints ONE line with n_gpus 2 whose counts are both ranks'.  Here the two ranks share this box's one
    GPU and exchange over gloo (AZSP_BENCH_BACKEND=gloo; with nccl the same command refuses: fewer GPUs than ranks) --
    real engines, the real all-gather of exported records, the real rank combination."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--slots", "256", "--sims", "20", "--steps", "2", "--warmup", "1",
            "--single", "--no-clock-probe", "--no-eval-cache"]
    r = subprocess.run(args, env=dict(env, AZSP_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["episodes"] == 2 * 2 * 256 and len(out["ranks"]["ms_per_step"]) == 2
    assert out["config"]["launched_by"].startswith("bench.py") and out["value"] > 0 and 0 < out["roofline"]["frac"] <= 1.0
    import torch
    if torch.cuda.device_count() < 2:                          # RCCL needs one GPU per rank: refused, no line, non-zero exit
        r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_reference_mcts_tests_intent_on_the_engine(azsp):
    """The reference's own (stale) MCTSTests, restated: Connect2 + fixed-prior mock evaluator through the
    MCTS facade (host-callback evaluator); visit counts equal the oracle's and satisfy the asserted orderings."""
    from helpers import MCTS_INTENT_CASES, FixedPriorModel
    rules = oz.C2Rules()
    for priors, value, board, sims, check in MCTS_INTENT_CASES:
        model = FixedPriorModel(priors, value)
        root = azsp.MCTS(azsp.Connect2Game(), model, sims).run(np.array(board), to_play=1)
        counts = [root.children[a].visit_count if a in root.children else 0 for a in range(4)]
        ref = oz.run_batch(rules, model.predict, sims, [np.array(board)], [1])[0]
        assert counts == oz.root_counts(ref, 4) and check(counts)
        assert root.visit_count == sims and float(root.value_sum) == float(ref.w)


@pytest.mark.parametrize("seed,salt,cache", [(101, 11, False), (102, 12, False), (102, 12, True)])
def test_large_randomised_differential_run(azsp, seed, salt, cache):
    """A longer differential run (2048 games x 150 sims, ~9 M simulations per seed) against the C oracle: any
    rare divergence (ties, deep trees, slot restarts late in the run) would show as a mismatch."""
    n, sims = 2048, 150
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), eval_cache=cache)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    flat_s = np.concatenate([np.stack([s.state for s in ep.samples]).reshape(len(ep), -1) for ep in eps])
    flat_p = np.concatenate([np.stack([s.policy for s in ep.samples]) for ep in eps])
    flat_v = np.array([s.value for ep in eps for s in ep.samples])
    assert np.array_equal(flat_s, ref["states"]) and np.array_equal(flat_p, ref["policies"])
    assert np.array_equal(flat_v, ref["values"])
    assert [len(ep) for ep in eps] == np.diff(ref["ep_offset"][:n + 1]).tolist()


@pytest.mark.parametrize("opts", [{"sampling_plies": 6}, {"dirichlet_alpha": 0.8, "dirichlet_eps": 0.25},
                                  {"sampling_plies": 0}, {"sampling_plies": 4, "dirichlet_alpha": 0.3, "dirichlet_eps": 0.4}])
@pytest.mark.parametrize("cache", [False, True], ids=["nocache", "evalcache"])
def test_search_options_beyond_the_reference(azsp, opts, cache):
    """Temperature cut-off and root noise do not exist in the reference's generator, so the
    reference cannot pin them: they are pinned against the oracle's documented extension, bit for bit, and
    are off by default (every other test runs the reference's behaviour)."""
    n, sims, seed, salt = 12, 40, 9, 3
    rules = oz.C4Rules()
    np.random.seed(seed)
    ref = list(oz.generate_episodes(rules, oz.HashModel(rules, salt).predict, sims, n, **opts))
    after = float(np.random.random_sample())
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), eval_cache=cache, **opts)
    eps = list(gen.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == after
    assert len(eps) == len(ref) == n
    for ep, r in zip(eps, ref):
        assert [s.state.tolist() for s in ep.samples] == [np.asarray(s).tolist() for s, _, _ in r]
        assert [s.policy.tolist() for s in ep.samples] == [np.asarray(p).tolist() for _, p, _ in r]
        assert [s.value for s in ep.samples] == [int(z) for _, _, z in r]


# ------------------------------------------------------------------------------------ Gomoku 9x9 (new plugin game)
def test_gomoku_search_matches_reference_algorithm_fixtures(azsp):
    """81 actions per node (64 lanes x 2 per game), numpy's pairwise np.sum over the priors, 128-bit boards:
    bit-exact against the reference's MCTS.run_batch run with the Gomoku plugin (tests/golden/gomoku_hash.json)."""
    for run in load_json("gomoku_hash.json")["search"]:
        boards = np.array(run["boards"], dtype=np.int8)
        with _engine(azsp, "gomoku9", len(boards), run["sims"], hash_salt=run["salt"]) as e:
            e.use_hash_evaluator()
            _check_run(e.run_batch(boards), run)


@pytest.mark.parametrize("model_cls", [HashStub, HostOnlyModel])
def test_gomoku_generate_episodes_matches_reference_algorithm_fixtures(azsp, model_cls):
    for run in load_json("gomoku_hash.json")["episodes"]:
        cfg = azsp.AlphaZeroConfig(32, 1, run["sims"], run["n"], 1, 1)
        gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=cfg)
        np.random.seed(run["seed"])
        eps = list(gen.generate_episodes(model_cls("gomoku9", run["salt"])))
        assert float(np.random.random_sample()) == run["rng_after"]
        _episodes_equal(eps, run["episodes"])


@pytest.mark.parametrize("n,sims,seed,salt", [(96, 60, 5, 1), (8, 200, 6, 2)])
def test_gomoku_generate_matches_c_oracle_at_larger_sizes(azsp, n, sims, seed, salt):
    """BASELINE configs[4]'s search shape (200 sims, 81 actions) at oracle-sized slot counts."""
    ref = co.generate("gomoku9", sims, n, salt, seed=seed, trace=False)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1))
    eps = list(gen.generate_episodes(HashStub("gomoku9", salt)))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    assert len(eps) == n
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()


@pytest.mark.parametrize("blocks,ch", [(2, 32), (3, 64), (2, 96), (2, 128), (1, 192), (2, 256), (10, 256)])
def test_gomoku_resnet_predict_on_device(azsp, blocks, ch):
    """The same ResNet architecture on the 9x9 plugin game (policy Linear 32*81 -> 81, legal = empty cells).
    The reference's ResNet.predict cannot mask a 9x9 board, so the check is against this module's own PyTorch
    forward on the CPU (tolerance NET_TOL_WIDE; (10, 256) is BASELINE configs[4]'s network) -- parity is not pinnable
    by the reference here."""
    import torch
    torch.manual_seed(7 * blocks + ch)
    net = azsp.ResNet((9, 9), 81, blocks, ch)
    net.eval()
    rules = oz.G9Rules()
    boards = random_midgame_boards(rules, np.random.RandomState(3), 10, 60)
    p, v = net.predict(boards)
    pt, vt, mask = _torch_reference(net, rules, boards)
    assert np.stack(p).shape == (10, 81)
    err_p, err_v = np.max(np.abs(np.stack(p) - pt)), np.max(np.abs(np.array(v) - vt))
    print(f"[net-error] gomoku9 {blocks}x{ch}: max |dp| = {err_p:.2e}, max |dv| = {err_v:.2e}")
    assert err_p < NET_TOL and err_v < NET_TOL
    assert np.all(np.stack(p)[mask == 0] == 0)


def test_gomoku_self_play_with_device_network_bit_exact_vs_oracle(azsp):
    """End-to-end Gomoku self-play with the device network: the C oracle asks the same kernel for (P, V)."""
    import torch
    from azsp.engine import Engine
    from azsp.weights import resnet_blob
    torch.manual_seed(5)
    net = azsp.ResNet((9, 9), 81, 2, 32)
    blob, blocks, ch = resnet_blob(net)
    n, sims, seed = 6, 30, 4
    with Engine("gomoku9", 64, 1, res_blocks=blocks, channels=ch) as pe:
        pe.load_resnet(blob)
        cb = co.make_predict_cb(lambda bs: pe.predict(np.stack(bs)), (9, 9), 81)
        ref = co.generate("gomoku9", sims, n, seed=seed, predict_cb=cb, trace=False)
    np.random.seed(seed)
    gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1))
    eps = list(gen.generate_episodes(net))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    for e, ep in enumerate(eps):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])
        assert [s.value for s in ep.samples] == ref["values"][lo:hi].tolist()


def test_background_thread_generation_next_to_torch_training(azsp):
    """The reference can run generation on a daemon thread while Lightning trains
    (src/alphazero_less_simple/core/training/datamodule.py:19-47,111-117).  The engine uses private non-blocking
    HIP streams and ctypes releases the GIL, so generation on a second thread, concurrent with PyTorch work on the
    default stream, must give exactly the single-threaded result."""
    import threading
    import torch
    n, sims, seed, salt = 1024, 60, 17, 5
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    net = torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512)).cuda()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    net(torch.randn(8, 512, device="cuda")).sum().backward()       # warm the torch side up before the race
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1))
    out, err = [], []

    def worker():
        try:
            np.random.seed(seed)                    # the generator is the only consumer of the global RNG here
            out.extend(gen.generate_episodes(HashStub("connect4", salt)))
        except BaseException as exc:                # noqa: BLE001
            err.append(exc)

    th = threading.Thread(target=worker, daemon=True)
    th.start()
    steps = 0
    while th.is_alive() and steps < 20000:          # "training" on the default stream meanwhile
        x = torch.randn(256, 512, device="cuda")
        loss = net(x).square().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        steps += 1
    th.join(timeout=120)
    assert not err and not th.is_alive() and steps > 0
    assert len(out) == n
    for e, ep in enumerate(out):
        lo, hi = ref["ep_offset"][e], ref["ep_offset"][e + 1]
        assert np.array_equal(np.stack([s.state for s in ep.samples]).reshape(hi - lo, -1), ref["states"][lo:hi])
        assert np.array_equal(np.stack([s.policy for s in ep.samples]), ref["policies"][lo:hi])


def test_clock_probe_reports_a_plausible_in_kernel_clock(azsp):
    """azsp_probe_clock: the stamped diagnostic instantiation of the fp32 kernel returns the clock held inside the
    kernel; after a few hundred launches it must be a sane MI355X figure, and the probe must refuse engines it
    does not cover."""
    import torch
    from azsp.weights import resnet_blob
    torch.manual_seed(0)
    blob = resnet_blob(azsp.ResNet((6, 7), 7, 4, 64))[0]
    boards = np.stack(random_midgame_boards(oz.C4Rules(), np.random.RandomState(1), 1536, 41))
    with _engine(azsp, "connect4", 1536, 1, res_blocks=4, channels=64) as e:
        e.load_resnet(blob)
        p0, v0 = e.predict(boards)
        for _ in range(300):
            e.predict(boards)
        ghz, span = e.probe_clock(boards)
        p1, v1 = e.predict(boards)
    assert 1.5 < ghz < 2.6 and 50.0 < span < 5000.0
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)      # the probe leaves the engine as it was
    with _engine(azsp, "connect4", 1536, 1, res_blocks=4, channels=64, f16_split=True) as e:      # the split-fp16 kernel's stamped form
        e.load_resnet(blob)
        q0, w0 = e.predict(boards)
        for _ in range(300):
            e.predict(boards)
        ghz16, span16 = e.probe_clock(boards)
        q1, w1 = e.predict(boards)
    print(f"held clock: fp32 kernel {ghz:.3f} GHz, f16x2 kernel {ghz16:.3f} GHz")
    assert 1.2 < ghz16 < 2.6 and 20.0 < span16 < 5000.0
    assert np.array_equal(q0, q1) and np.array_equal(w0, w1)
    with _engine(azsp, "connect4", 64, 1, hash_salt=1) as e:
        e.use_hash_evaluator()
        with pytest.raises(Exception):
            e.probe_clock(boards[:8])


def test_c_abi_rejects_bad_arguments_with_messages(azsp):
    """Raw C-ABI calls (ctypes): wrong struct sizes, unknown ids, out-of-range slots, out-of-order calls and decided
    initial positions come back as negative codes with a message; nothing crosses the boundary as a crash."""
    import ctypes as C
    from azsp import _lib as L
    lib = L.load()
    h = C.c_void_p()
    good = L.Config(C.sizeof(L.Config), 0, 8, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0)

    def err():
        return lib.azsp_last_error().decode()

    bad = L.Config(C.sizeof(L.Config) - 4, 0, 8, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert lib.azsp_create(C.byref(bad), C.byref(h)) < 0 and "struct_size" in err()
    for field, value, word in (("game", 9, "game"), ("n_slots", -1, "n_slots"), ("n_sims", 70000, "n_sims"),
                               ("device", 99, "device"), ("search_rules", 7, "rule")):
        cfg = L.Config(*[getattr(good, f) for f, _ in L.Config._fields_])
        setattr(cfg, field, value)
        assert lib.azsp_create(C.byref(cfg), C.byref(h)) < 0 and word in err(), (field, err())
    cfg = L.Config(*[getattr(good, f) for f, _ in L.Config._fields_])
    cfg.search_rules, cfg.game = L.RULES_GEN1, L.GAME_CONNECT2
    assert lib.azsp_create(C.byref(cfg), C.byref(h)) < 0 and "Connect-4" in err()
    cfg = L.Config(*[getattr(good, f) for f, _ in L.Config._fields_])
    cfg.res_blocks, cfg.channels = 2, 999
    assert lib.azsp_create(C.byref(cfg), C.byref(h)) < 0 and "geometry" in err()
    assert lib.azsp_create(C.byref(good), C.byref(h)) == 0
    try:
        res = L.PlyResult()
        u = (C.c_double * 8)()
        assert lib.azsp_play_commit(h, -1, C.byref(res)) < 0 and "play_begin" in err()
        assert lib.azsp_root_counts(h, (C.c_int32 * 56)()) < 0 and "before" in err()
        assert lib.azsp_fetch_episodes(h, 0, 3, None, None, None, None, None, None, 0) < 0 and "not emitted" in err()
        opts = L.SearchOptions(C.sizeof(L.SearchOptions), -1, 2.0, 1.0, 0.0)
        assert lib.azsp_set_search_options(h, C.byref(opts)) < 0 and "noise_eps" in err()
        opts = L.SearchOptions(C.sizeof(L.SearchOptions), -1, 0.0, 1.0, -3.0)
        assert lib.azsp_set_search_options(h, C.byref(opts)) < 0 and "c_base" in err()
        won = np.zeros((6, 7), np.int8)
        won[5, :4] = -1                                   # the opponent already has four in a row
        assert lib.azsp_set_initial_board(h, won.ctypes.data_as(C.POINTER(C.c_int8))) < 0 and "decided" in err()
        boards = np.zeros((9, 42), np.int8)
        assert lib.azsp_run_batch(h, boards.ctypes.data_as(C.POINTER(C.c_int8)), 9, None, None, None, None, None) < 0 and "exceeds" in err()
        assert lib.azsp_set_evaluator(h, L.EVAL_RESNET, L.PREDICT_FN(), None) < 0 and "ResNet" in err()
        assert lib.azsp_set_evaluator(h, L.EVAL_HOST, L.PREDICT_FN(), None) < 0 and "callback" in err()
        assert lib.azsp_search(h) == 0 and lib.azsp_play(h, u, C.byref(res)) == 0 and res.ply == 0
    finally:
        lib.azsp_destroy(h)


# ------------------------------------------------------------------------------------ evaluation cache
def _run_gen(azsp, net, n, sims, seed, **kw):
    np.random.seed(seed)
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), **kw)
    eps = list(g.generate_episodes(net))
    return eps, g.last_stats, float(np.random.random_sample())


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_eval_cache_identical_episodes_at_2048x100_and_fewer_evaluations(azsp, f16):
    """AZSP_FLAG_EVAL_CACHE must be invisible in the results: 2048 games x 100 sims with the 4 x 64 device network, cache
    off and on -- the same episodes in the same order, the same RNG position, the same reference-equivalent accounting
    (evals, terminal leaves, path nodes) -- while the number of evaluations actually executed drops (the 2048 identical
    first plies, transpositions, last ply's subtree).  Twice on the same generator: the second call starts from an empty
    table again (a position is never carried over between generation runs)."""
    import torch
    torch.manual_seed(0)
    net = azsp.ResNet((6, 7), 7, 4, 64).eval()
    n, sims = 2048, 100
    off, s_off, r_off = _run_gen(azsp, net, n, sims, 7, f16_split=f16)
    np.random.seed(7)
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16, eval_cache=True)
    on = list(g.generate_episodes(net))
    s_on, r_on = g.last_stats, float(np.random.random_sample())
    assert on == off and r_on == r_off
    for k in ("plies", "rounds", "evals", "terminal_leaves", "path_nodes"):
        assert s_on[k] == s_off[k], k
    assert s_off["evals_executed"] == s_off["evals"] and s_off["eval_cache_entries"] == 0
    assert s_on["evals_executed"] < 0.6 * s_on["evals"] and s_on["eval_cache_entries"] >= 1024
    print(f"[eval-cache] 2048 x 100, f16x2={f16}: {s_on['evals']} reference evaluations, {s_on['evals_executed']} executed "
          f"({s_on['evals'] / s_on['evals_executed']:.2f}x fewer)")
    first = s_on["evals_executed"]
    eng = g.engine_for(net)
    eng.reset_stats()
    np.random.seed(7)
    assert list(g.generate_episodes(net)) == off
    again = g.last_stats["evals_executed"]
    assert abs(again - first) <= 0.01 * first      # (two slot groups meeting one new position in one search both evaluate it)


@pytest.mark.parametrize("f16", [False, True], ids=["f32", "f16x2"])
def test_eval_cache_identical_episodes_with_a_wide_network(azsp, f16):
    """The same at 192 channels, where the evaluation cache's launches (gated by the list's length on the device) switch
    between the one-board and the three-board forms of the wide nets: 600 games x 30 sims in one slot group, cache off / on."""
    import torch
    torch.manual_seed(3)
    net = azsp.ResNet((6, 7), 7, 1, 192).eval()
    n, sims = 600, 30
    off, s_off, r_off = _run_gen(azsp, net, n, sims, 11, f16_split=f16)
    np.random.seed(11)
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), f16_split=f16, eval_cache=True,
                              n_groups=1)
    on = list(g.generate_episodes(net))
    s_on, r_on = g.last_stats, float(np.random.random_sample())
    assert on == off and r_on == r_off
    assert s_on["evals"] == s_off["evals"] and s_on["evals_executed"] < 0.7 * s_on["evals"]


def test_eval_cache_edge_cases(azsp):
    """A table far too small for the run (fills up: later positions fall back to their slots' private entries), boards
    without a key (a floating stone: MCTS.run_batch accepts anything), tree reuse on top, new weights on a live
    engine, and the games it does not exist for."""
    from azsp._lib import AzspError
    # 1. table of 1024 entries for ~0.5 M evaluations
    n, sims, seed, salt = 512, 60, 31, 6
    ref = co.generate("connect4", sims, n, salt, seed=seed, trace=False)
    np.random.seed(seed)
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), eval_cache=True,
                              eval_cache_entries=1024)
    eps = list(g.generate_episodes(HashStub("connect4", salt)))
    assert float(np.random.random_sample()) == ref["next_uniform"]
    _assert_equals_c_oracle(eps, ref, n)
    st = g.last_stats
    assert st["eval_cache_entries"] == 1024 and st["evals"] == ref["stats"][0] and st["evals_executed"] > 0.5 * st["evals"]
    # 2. boards with floating stones have no key: searched from private entries, same results as without the cache
    rng = np.random.RandomState(3)
    rules = oz.C4Rules()
    boards = np.stack(random_midgame_boards(rules, rng, 24, 20)).astype(np.int8)
    for b in boards[::2]:
        col = int(rng.randint(0, 7))
        if b[0, col] == 0 and b[1, col] == 0:
            b[0, col] = 1                                  # a stone hanging in the top row
    outs = []
    for cache in (False, True):
        with _engine(azsp, "connect4", 24, 80, hash_salt=5, eval_cache=cache) as e:
            e.use_hash_evaluator()
            outs.append(e.run_batch(boards))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # 3. tree reuse (the trees persist, the cache persists: two kinds of memory, one result)
    a = _run_gen(azsp, HashStub("connect4", 9), 48, 40, 12, tree_reuse=True)
    b = _run_gen(azsp, HashStub("connect4", 9), 48, 40, 12, tree_reuse=True, eval_cache=True)
    assert a[0] == b[0] and a[2] == b[2]
    # 4. new weights on a live engine void every cached row
    import torch
    torch.manual_seed(1)
    net1 = azsp.ResNet((6, 7), 7, 2, 64).eval()
    torch.manual_seed(2)
    net2 = azsp.ResNet((6, 7), 7, 2, 64).eval()
    g = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, 30, 100, 1, 1), eval_cache=True)
    np.random.seed(4)
    list(g.generate_episodes(net1))
    np.random.seed(5)
    second = list(g.generate_episodes(net2))
    fresh, _, _ = _run_gen(azsp, net2, 100, 30, 5)
    assert second == fresh
    # 5. the search counter in the key word has 12 bits: after 4095 searches the table is cleared and the count restarts
    small = np.stack(random_midgame_boards(rules, np.random.RandomState(8), 6, 12)).astype(np.int8)
    with _engine(azsp, "connect4", 6, 3, hash_salt=2, eval_cache=True) as e:
        e.use_hash_evaluator()
        first = e.run_batch(small)
        for i in range(4200):
            out = e.run_batch(small)
            if i % 500 == 0 or i > 4080:
                assert all(np.array_equal(first[k], out[k]) for k in first), i
    # 6. no 64-bit key for 9x9 boards
    with pytest.raises(AzspError, match="EVAL_CACHE"):
        _engine(azsp, "gomoku9", 8, 10, eval_cache=True)


def test_generator_reuploads_weights_exactly_when_they_changed(azsp, monkeypatch):
    """generate_episodes exports and uploads the model's weights only when they changed since its last call (storage
    address + in-place version of every tensor, bound to the model object): an optimizer-style in-place update, a
    load_state_dict, another model object -- also one that reuses a freed model's memory -- and invalidate_device_weights()
    all trigger the upload; a second call on untouched weights does not; and the episodes are those of a fresh generator."""
    import copy
    import torch
    from azsp import mcts as M
    calls = []
    real = M.configure_engine_for_model
    monkeypatch.setattr("azsp.mcts.configure_engine_for_model", lambda e, m: (calls.append(id(e)), real(e, m))[1])
    torch.manual_seed(3)
    net = azsp.ResNet((6, 7), 7, 2, 32).eval()
    cfg = azsp.AlphaZeroConfig(32, 1, 10, 24, 1, 1)
    gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=cfg)

    def run(model, seed=5):
        np.random.seed(seed)
        return list(gen.generate_episodes(model))

    def fresh(model, seed=5):                                       # (its own engine: its upload is not counted)
        before = list(calls)
        np.random.seed(seed)
        out = list(azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=cfg).generate_episodes(model))
        calls[:] = before
        return out
    a = run(net)
    assert len(calls) == 1
    assert run(net) == a and len(calls) == 1                        # untouched weights: no export
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))                      # what an optimizer step does
    b = run(net)
    n_fresh = len(calls)
    assert n_fresh == 2 and b == fresh(net) and b != a
    calls.clear()
    net.load_state_dict(copy.deepcopy(net.state_dict()))
    run(net)
    assert len(calls) == 1
    clone = copy.deepcopy(net)                                      # the reference's get_inference_clone
    assert run(clone) == b and len(calls) == 2
    del clone
    torch.manual_seed(9)
    other = azsp.ResNet((6, 7), 7, 2, 32).eval()                    # may land on the freed clone's addresses, versions 0
    c = run(other)
    assert len(calls) == 3 and c == fresh(other)
    with torch.no_grad():
        other.value_head[4].bias.data.add_(0.5)                     # a write the version counters do not see
    gen.invalidate_device_weights()
    d = run(other)
    assert len(calls) == 4 and d == fresh(other)
