"""CPU tests of the host-side mirror: game plugins vs the reference fixtures, weight export,
episode JSON schema, and that the C-ABI library loads and exports every declared symbol."""
import os
import re

import numpy as np
import pytest

from helpers import load_json

import azsp
from azsp import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fname,game", [("rules_connect4.json", azsp.Connect4Game()),
                                        ("rules_connect2.json", azsp.Connect2Game())])
def test_host_games_match_reference(fname, game):
    for c in load_json(fname)["cases"]:
        b = np.array(c["board"], dtype=int)
        nb, npl = game.get_next_state(b, c["player"], c["action"])
        assert nb.tolist() == c["next"] and npl == c["next_player"]
        assert game.get_valid_moves(b) == c["valid"]
        assert game.has_legal_moves(b) == c["has_moves"]
        assert game.is_win(b, 1) == c["win_p"] and game.is_win(b, -1) == c["win_m"]
        assert game.get_reward_for_player(b, 1) == c["reward_p"]
        assert game.get_reward_for_player(b, -1) == c["reward_m"]
    assert game.get_action_size() == len(c["valid"])
    assert np.all(game.get_init_board() == 0)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "azsp.h")).read()
    declared = set(re.findall(r"\b(azsp_[a-z_0-9]+)\s*\(", header))
    declared -= {"azsp_predict_fn"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.azsp_abi_version() == _lib.ABI_VERSION == 3
    import ctypes as C
    r, c, a, mp = (C.c_int32() for _ in range(4))
    assert lib.azsp_game_info(0, C.byref(r), C.byref(c), C.byref(a), C.byref(mp)) == 0
    assert (r.value, c.value, a.value, mp.value) == (6, 7, 7, 42)
    assert lib.azsp_game_info(7, None, None, None, None) != 0
    assert b"unknown game" in lib.azsp_last_error()
    assert lib.azsp_record_bytes(0) == 48 + 56


def test_weight_blob_layout():
    import torch
    from azsp.weights import resnet_blob, resnet_key_order
    torch.manual_seed(1)
    net = azsp.ResNet((6, 7), 7, 3, 32)
    blob, blocks, ch = resnet_blob(net)
    assert (blocks, ch) == (3, 32)
    assert blob.dtype == np.float32 and blob.size == _lib.load().azsp_resnet_blob_floats(0, 3, 32)
    sd = net.state_dict()
    off = 0
    for k in resnet_key_order(3):
        t = sd[k].numpy().reshape(-1)
        assert np.array_equal(blob[off:off + t.size], t)
        off += t.size
    assert off == blob.size
    # same state_dict keys as the reference fixture (checkpoint interchange)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "resnet_forward.npz"))
    ref_keys = {k[5:] for k in z.files if k.startswith("m_sd/")}
    assert ref_keys == set(azsp.ResNet((6, 7), 7, 4, 64).state_dict().keys())


def test_resnet_forward_matches_reference_vectors_on_cpu_torch():
    """The host ResNet (training path, PyTorch) reproduces the reference's logits from its weights."""
    import torch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "resnet_forward.npz"))
    for tag in ("s", "m"):
        blocks, ch = z[f"{tag}_arch"].tolist()
        net = azsp.ResNet((6, 7), 7, blocks, ch)
        net.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_sd/")})
        net.eval()
        with torch.no_grad():
            logits, vl = net.forward(net._states_to_tensor(list(z["boards"].astype(np.int64))))
        assert np.max(np.abs(logits.numpy() - z[f"{tag}_logits"])) < 1e-5
        assert np.max(np.abs(vl.numpy() - z[f"{tag}_vlogit"])) < 1e-5


def test_episode_json_schema_roundtrip():
    ep = azsp.Episode()
    ep.add_sample(azsp.Sample(state=np.zeros((6, 7), dtype=np.int64), policy=np.full(7, 1 / 7), value=1))
    d = ep.to_dict()
    assert set(d) == {"samples"} and set(d["samples"][0]) == {"state", "policy", "value"}
    back = azsp.Episode.from_dict(d)
    assert len(back) == 1 and back.samples[0].value == 1 and np.array_equal(back.current_state, ep.current_state)


def test_format_dataset_accepts_generator_output():
    import torch
    net = azsp.ResNet((6, 7), 7, 1, 32)
    states = [np.zeros((6, 7), dtype=np.int64), -np.eye(6, 7, dtype=np.int64)]
    ds = net.format_dataset(states, [np.full(7, 1 / 7)] * 2, [1, -1])
    x, pi, z = ds.tensors
    assert x.shape == (2, 3, 6, 7) and pi.dtype == torch.float32 and z.tolist() == [1.0, -1.0]
    loss = net.training_step((x, pi, z), 0)
    assert torch.isfinite(loss)


def test_episode_json_wire_format_roundtrip(tmp_path):
    """save_episodes / load_episodes write and read the reference's file format
    (datamodule.py:87-103): a JSON list of {"samples": [{"state","policy","value"}]}."""
    import json
    from azsp.replay import load_episodes, save_episodes
    run = load_json("episodes_hash.json")["runs"][1]
    eps = [azsp.Episode(samples=[azsp.Sample(np.array(s), np.array(p), v)
                                 for s, p, v in zip(g["states"], g["policies"], g["values"])]) for g in run["episodes"]]
    path = tmp_path / "sub" / "episodes_iter1.json"
    save_episodes(path, eps)
    raw = json.load(open(path))
    assert isinstance(raw, list) and set(raw[0]) == {"samples"} and set(raw[0]["samples"][0]) == {"state", "policy", "value"}
    back = load_episodes(path)
    assert len(back) == len(eps)
    for a, b in zip(eps, back):
        assert len(a) == len(b)
        for x, y in zip(a.samples, b.samples):
            assert np.array_equal(x.state, y.state) and np.array_equal(x.policy, y.policy) and x.value == y.value


def _wire_fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "format_dataset.npz"))
    meta = __import__("json").loads(bytes(z["meta"]).decode())
    text = open(os.path.join(os.path.dirname(__file__), "golden", "episodes_wire.json")).read()
    return z, meta, text


def test_json_writer_and_loader_match_the_reference_written_file(tmp_path):
    """tests/golden/episodes_wire.json is the text the REFERENCE's DataModule._save_episodes wrote for one of its own
    generator runs (Episode.to_dict + json.dump, episode.py:15-24,64-66, datamodule.py:87-96).  The same run's
    episodes (fixture episodes_hash.json, n = 16) written by azsp.replay.save_episodes must give that file byte for
    byte, and load_episodes must parse the reference's file into what Episode.from_dict gives there
    (int64 boards, float64 policies, int values)."""
    from azsp.replay import load_episodes, save_episodes
    z, meta, text = _wire_fixture()
    run = [r for r in load_json("episodes_hash.json")["runs"]
           if (r["game"], r["n"], r["sims"], r["seed"], r["salt"]) == ("connect4", meta["n"], meta["sims"], meta["seed"], meta["salt"])][0]
    eps = [azsp.Episode(samples=[azsp.Sample(np.array(s, dtype=np.int64), np.array(p, dtype=np.float64), int(v))
                                 for s, p, v in zip(g["states"], g["policies"], g["values"])]) for g in run["episodes"]]
    path = tmp_path / "episodes_iter1.json"
    save_episodes(path, eps)
    assert open(path).read() == text
    back = load_episodes(os.path.join(os.path.dirname(__file__), "golden", "episodes_wire.json"))
    assert [len(e) for e in back] == meta["loaded"]["lengths"]
    s0 = back[0].samples[0]
    assert str(s0.state.dtype) == meta["loaded"]["state_dtype"] and str(s0.policy.dtype) == meta["loaded"]["policy_dtype"]
    assert type(s0.value).__name__ == meta["loaded"]["value_type"] and list(s0.state.shape) == meta["loaded"]["state_shape"]
    assert back == eps
    save_episodes(path, back)                     # load -> save is the identity on the reference's file
    assert open(path).read() == text


def test_host_format_dataset_matches_reference_tensors():
    """BaseModel.format_dataset + ResNet._states_to_tensor of the host mirror vs the tensors the reference's own
    methods produced from the same samples (base_model.py:61-67, resnet.py:115-128), bit for bit."""
    import torch
    z, meta, text = _wire_fixture()
    eps = [azsp.Episode.from_dict(d) for d in __import__("json").loads(text)]
    samples = [s for e in eps for s in e.samples]
    ds = azsp.ResNet((6, 7), 7, 1, 16).format_dataset([s.state for s in samples], [s.policy for s in samples],
                                                      [s.value for s in samples])
    assert [str(t.dtype) for t in ds.tensors] == meta["dtypes"] and [list(t.shape) for t in ds.tensors] == meta["shapes"]
    for t, k in zip(ds.tensors, ("planes", "policies", "values")):
        assert torch.equal(t, torch.from_numpy(z[k]))


def test_shard_allowances_reproduce_the_global_first_n_rule():
    """Brute-force check of the per-ply bookkeeping of the sharded exact mode against a flat slot scan."""
    from azsp.sharded import shard_allowances, shard_ranges
    rng = np.random.RandomState(0)
    for _ in range(500):
        g = int(rng.randint(1, 6))
        n = int(rng.randint(g, 40))
        ranges = shard_ranges(n, g)
        assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        done = rng.rand(n) < rng.rand()
        remaining = int(rng.randint(0, n + 3))
        allow, stop = shard_allowances([int(done[lo:hi].sum()) for lo, hi in ranges], remaining)
        counted = np.flatnonzero(done)[:max(remaining, 0)]
        assert allow == [int(((counted >= lo) & (counted < hi)).sum()) for lo, hi in ranges]
        if remaining > 0 and done.sum() >= remaining:
            last = counted[-1]
            assert stop == [r for r, (lo, hi) in enumerate(ranges) if lo <= last < hi][0]
        else:
            assert stop is None


@pytest.mark.parametrize("fname,gid,A", [("rules_connect4.json", 0, 7), ("rules_connect2.json", 1, 4)])
def test_device_game_traits_match_reference_on_host(fname, gid, A):
    """The bitboard traits the kernels run (csrc/azsp_games.h, shared host/device code) evaluated on the CPU
    through azsp_game_rules against every rule vector recorded from the reference -- including full columns,
    gravity-violating and over-full boards."""
    import ctypes as C
    lib = _lib.load()
    for c in load_json(fname)["cases"]:
        for player in (1, -1):
            canon = (player * np.array(c["board"], dtype=np.int64)).astype(np.int8).reshape(-1)
            mask, term = C.c_int32(), C.c_int32()
            nxt = np.zeros_like(canon)
            action = c["action"] if player == c["player"] else 0
            assert lib.azsp_game_rules(gid, canon.ctypes.data_as(C.POINTER(C.c_int8)), action,
                                       nxt.ctypes.data_as(C.POINTER(C.c_int8)), C.byref(mask), C.byref(term)) == 0
            assert [(mask.value >> a) & 1 for a in range(A)] == c["valid"]
            rew = c["reward_p"] if player == 1 else c["reward_m"]
            assert term.value == (2 if rew is None else rew)
            if player == c["player"]:
                # reference: next = get_next_state(board, player, action); canonical for the opponent = -player * next
                expect = (-player * np.array(c["next"], dtype=np.int64)).astype(np.int8).reshape(-1)
                assert np.array_equal(nxt, expect)


def test_gomoku_device_trait_matches_oracle_rules_on_host():
    """The Gomoku9 bit-set trait (two 64-bit words per side, 5-in-a-row by shift-and) evaluated on the CPU
    against the cell-array rules of the oracle, over random playouts and arbitrary boards."""
    import ctypes as C
    from oracle import az_oracle as oz
    lib = _lib.load()
    rules = oz.G9Rules()
    rng = np.random.RandomState(8)

    def check(board, action):
        canon = np.asarray(board, dtype=np.int8).reshape(-1)
        mask = (C.c_int32 * 3)()
        term = C.c_int32()
        nxt = np.zeros(81, np.int8)
        assert lib.azsp_game_rules(2, canon.ctypes.data_as(C.POINTER(C.c_int8)), int(action),
                                   nxt.ctypes.data_as(C.POINTER(C.c_int8)), mask, C.byref(term)) == 0
        assert [(mask[a >> 5] >> (a & 31)) & 1 for a in range(81)] == rules.valid_moves(board)
        rew = rules.reward(board, 1)
        assert term.value == (2 if rew is None else rew)
        nb, _ = rules.next_state(board, 1, action)
        assert np.array_equal(nxt.reshape(9, 9), -nb)

    for _ in range(40):
        b, p = rules.init_board(), 1
        for _ in range(81):
            legal = [i for i, v in enumerate(rules.valid_moves(b)) if v]
            a = int(rng.choice(legal))
            check(rules.canonical(b, p), a)
            b, p = rules.next_state(b, p, a)
            if rules.reward(b, p) is not None:
                check(rules.canonical(b, p), int(rng.randint(81)))
                break
    for _ in range(200):
        check(rng.randint(-1, 2, size=(9, 9)), int(rng.randint(81)))
    lines = np.zeros((9, 9), dtype=int)
    lines[4, 2:7] = 1                                   # exactly five across the 64-bit word boundary rows
    check(lines, 0)
    lines = np.zeros((9, 9), dtype=int)
    for i in range(5):
        lines[4 + i, 8 - i] = -1                        # anti-diagonal reaching the last row
    check(lines, 0)


# ------------------------------------------------------------------------------------ gen-1 surface (host side)
def test_connect_state_agrees_with_the_gen1_oracle_rules():
    """azsp.connect_state (the stand-in for `simulator.game.connect`) against the oracle's independent rules, and
    against the engine's canonical encoding through the reference-pinned Connect-4 rules: random playouts."""
    from azsp import connect_state as cs
    from azsp.gen1 import canonical_cells, grid_from_cells
    from oracle import gen1_oracle as g1
    from oracle import az_oracle as oz
    rng = np.random.RandomState(0)
    rules = oz.C4Rules()
    ended = 0
    for _ in range(300):
        s, o = cs.Config(6, 7, 4).sample_initial_state(), g1.GConfig().sample_initial_state()
        board, player = rules.init_board(), 1
        while True:
            assert np.array_equal(s.grid, o.grid) and s.player == o.player and s.has_ended == o.has_ended
            assert s.reward.tolist() == o.reward.tolist()
            assert [a.column for a in s.actions] == [a.column for a in o.actions]
            cells = canonical_cells(s)
            assert np.array_equal(cells, rules.canonical(board, player))        # same position, engine encoding
            assert np.array_equal(grid_from_cells(cells, s.player), s.grid)
            if s.has_ended:
                # reference rules: reward for the player to move (-1 lost / 0 draw) <-> +1 for the previous mover
                rew = rules.reward(board, player)
                assert rew is not None and s.reward[1 - s.player] == -rew
                ended += 1
                break
            assert rules.reward(board, player) is None
            assert [a.column for a in s.actions] == [c for c, v in enumerate(rules.valid_moves(board)) if v]
            k = int(rng.randint(len(s.actions)))
            col = s.actions[k].column
            s, o = s.actions[k].sample_next_state(), o.actions[k].sample_next_state()
            board, player = rules.next_state(board, player, col)
    assert ended == 300


def test_gen1_records_json_and_model_surface():
    from azsp import connect_state as cs
    from azsp import gen1
    s = cs.Config().sample_initial_state()
    s = s.actions[3].sample_next_state()
    assert cs.State.from_json(s.to_json()) == s
    a = s.actions[2]
    assert cs.Action.from_json(a.to_json()) == a and hash(cs.Action.from_json(a.to_json())) == hash(a)
    ep = gen1.Episode()
    ep.add_sample(gen1.Sample(state=s, policy={x: 1 / 7 for x in s.actions}, value=[0.0, 0.0]))
    ep.backpropagate_outcome([1.0, -1.0])
    d = ep.to_dict()
    assert d["samples"][0]["value"] == [1.0, -1.0] and len(d["samples"][0]["policy"]) == 7
    assert d["samples"][0]["state"] == s.to_json() and ep.current_state is s
    with pytest.raises(TypeError):
        gen1.Model()                       # abstract: predict must be provided  (models/base/model.py:54)
    with pytest.raises(NotImplementedError):
        gen1._check_connect4(cs.Config(5, 5, 3).sample_initial_state())


def test_ctypes_structures_match_the_header(tmp_path):
    """The Python binding restates azsp_config / azsp_stats / azsp_ply_result / azsp_search_options field by field: their
    sizes and the offsets of their last fields must be what a C compiler makes of include/azsp.h (ABI drift would
    otherwise only show on a GPU box, as a struct_size refusal or, for the stats, as silently shifted numbers)."""
    import ctypes as C
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "azsp.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(azsp_config), sizeof(azsp_stats), sizeof(azsp_ply_result), sizeof(azsp_search_options),
           offsetof(azsp_config, timing_sample), offsetof(azsp_stats, infer_exec_flops), offsetof(azsp_stats, evals_executed));
    return 0;
}
''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(_lib.Config), C.sizeof(_lib.Stats), C.sizeof(_lib.PlyResult), C.sizeof(_lib.SearchOptions),
            _lib.Config.timing_sample.offset, _lib.Stats.infer_exec_flops.offset, _lib.Stats.evals_executed.offset]
    assert got == want, (got, want)
