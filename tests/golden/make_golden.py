#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference's hot path imports `lightning` (src/alphazero_simple/base_model.py:4), which is
not installed; an in-memory stub module (harness side only, nothing is written into the
reference tree) supplies `LightningModule = nn.Module + save_hyperparameters/log/device`.

    python tests/golden/make_golden.py            # regenerate everything

Outputs (data only -- inputs and expected outputs, no reference source text):
  rules_connect4.json / rules_connect2.json   board -> legal mask / next board / rewards
  search_hash.json      MCTS.run_batch transcripts under the hash evaluator
  episodes_hash.json    EpisodeGenerator.generate_episodes transcripts under the hash evaluator
  resnet_forward.npz    ResNet.predict vectors (seeded weights + boards -> p, v)
  episodes_resnet.json  generate_episodes with the real torch-CPU ResNet (BASELINE config 1)
  episodes_wire.json    the file DataModule._save_episodes writes (Episode.to_dict + json.dump) for one reference run
  format_dataset.npz    BaseModel.format_dataset / ResNet._states_to_tensor applied to that run's samples
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _install_lightning_stub():
    L = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    class Trainer:
        def __init__(self, *a, **k):
            pass

    L.LightningModule, L.LightningDataModule, L.Trainer = LightningModule, LightningDataModule, Trainer
    sys.modules["lightning"] = L
    for name, attrs in (("lightning.pytorch", {}),
                        ("lightning.pytorch.callbacks", {"ModelCheckpoint": object}),
                        ("lightning.pytorch.loggers", {"TensorBoardLogger": object})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


_install_lightning_stub()
sys.path.insert(0, "/root/reference/src")

from alphazero_simple.config import AlphaZeroConfig  # noqa: E402
from alphazero_simple.connect2_game import Connect2Game  # noqa: E402
from alphazero_simple.connect4_game import Connect4Game  # noqa: E402
from alphazero_simple import monte_carlo_tree_search as ref_mcts  # noqa: E402
from alphazero_simple.resnet import ResNet  # noqa: E402
from alphazero_less_simple.core.training.episode_generator import EpisodeGenerator  # noqa: E402

from oracle.az_oracle import C2Rules, C4Rules, G9Rules, HashModel  # noqa: E402  (hash definition only)


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rew(x):
    return None if x is None else int(x)


# ---------------------------------------------------------------------------------------------
def rules_connect4():
    g = Connect4Game()
    rng = np.random.RandomState(1234)
    cases = []

    def record(board, player, action):
        nb, npl = g.get_next_state(board, player, action)
        cases.append({
            "board": board.tolist(), "player": int(player), "action": int(action),
            "next": nb.tolist(), "next_player": int(npl),
            "valid": [int(v) for v in g.get_valid_moves(board)],
            "has_moves": bool(g.has_legal_moves(board)),
            "win_p": bool(g.is_win(board, 1)), "win_m": bool(g.is_win(board, -1)),
            "reward_p": rew(g.get_reward_for_player(board, 1)),
            "reward_m": rew(g.get_reward_for_player(board, -1)),
            "canon_m": g.get_canonical_board(board, -1).tolist(),
        })
        return nb, npl

    for _ in range(40):                       # random legal playouts to the end
        b, p = g.get_init_board(), 1
        while True:
            valid = g.get_valid_moves(b)
            if g.get_reward_for_player(b, p) is not None:
                record(b, p, int(rng.randint(7)))      # terminal board: still a vector
                break
            a = int(rng.choice([i for i, v in enumerate(valid) if v]))
            b, p = record(b, p, a)
    for _ in range(60):                       # arbitrary (gravity-violating, over-full) boards
        b = rng.randint(-1, 2, size=(6, 7))
        record(b, int(rng.choice([-1, 1])), int(rng.randint(7)))
    full = np.ones((6, 7), dtype=int)
    full[::2, ::2] = -1
    record(full, 1, 3)                        # full column: get_next_state is a no-op
    dump("rules_connect4.json", {"cases": cases})


def rules_connect2():
    g = Connect2Game()
    cases = []
    for code in range(3 ** 4):
        b = np.array([(code // 3 ** i) % 3 - 1 for i in range(4)], dtype=int)
        for player in (1, -1):
            for action in range(4):
                nb, npl = g.get_next_state(b, player, action)
                cases.append({
                    "board": b.tolist(), "player": player, "action": action,
                    "next": nb.tolist(), "next_player": int(npl),
                    "valid": [int(v) for v in g.get_valid_moves(b)],
                    "has_moves": bool(g.has_legal_moves(b)),
                    "win_p": bool(g.is_win(b, 1)), "win_m": bool(g.is_win(b, -1)),
                    "reward_p": rew(g.get_reward_for_player(b, 1)),
                    "reward_m": rew(g.get_reward_for_player(b, -1)),
                })
    dump("rules_connect2.json", {"cases": cases})


# ---------------------------------------------------------------------------------------------
def node_dump(root, A):
    kids = {}
    for a, c in root.children.items():
        kids[str(a)] = {"n": int(c.visit_count), "w": float(c.value_sum), "prior": float(c.prior),
                        "w_is_f32": isinstance(c.value_sum, np.float32),
                        "expanded": c.expanded()}
    size = 0
    depth = 0
    stack = [(root, 0)]
    while stack:
        n, d = stack.pop()
        size += 1
        depth = max(depth, d)
        stack.extend((c, d + 1) for c in n.children.values())
    return {"n": int(root.visit_count), "w": float(root.value_sum),
            "w_is_f32": isinstance(root.value_sum, np.float32), "kids": kids,
            "nodes": size, "depth": depth}


def midgame_boards(game, rng, n, max_plies):
    out = []
    while len(out) < n:
        b, p = game.get_init_board(), 1
        plies = int(rng.randint(0, max_plies))
        ok = True
        for _ in range(plies):
            valid = game.get_valid_moves(b)
            a = int(rng.choice([i for i, v in enumerate(valid) if v]))
            b, p = game.get_next_state(b, p, a)
            if game.get_reward_for_player(b, p) is not None:
                ok = False
                break
        if ok:
            out.append((game.get_canonical_board(b, p), p))
    return out


def search_hash():
    runs = []
    rng = np.random.RandomState(7)
    for gname, Game, Rules, sims_list, nb, mp in (
            ("connect4", Connect4Game, C4Rules, (1, 2, 25, 100, 400), 12, 30),
            ("connect2", Connect2Game, C2Rules, (1, 10, 50), 4, 2)):
        game, rules = Game(), Rules()
        boards = [(game.get_init_board(), 1)] + midgame_boards(game, rng, nb, mp)
        for salt in (0, 12345):
            model = HashModel(rules, salt)
            for sims in sims_list:
                m = ref_mcts.MCTS(game, model, sims)
                roots = m.run_batch([b for b, _ in boards], [p for _, p in boards])
                runs.append({
                    "game": gname, "salt": salt, "sims": sims,
                    "boards": [b.tolist() for b, _ in boards],
                    "to_plays": [int(p) for _, p in boards],
                    "roots": [node_dump(r, game.get_action_size()) for r in roots],
                })
    dump("search_hash.json", {"runs": runs})


# ---------------------------------------------------------------------------------------------
def run_reference_generator(game, model, sims, n, seed):
    """Run the reference EpisodeGenerator, recording per-ply root visit counts and chosen moves
    by wrapping (not editing) MCTS.run_batch and Node.select_action."""
    plies = []
    orig_run_batch = ref_mcts.MCTS.run_batch
    orig_select = ref_mcts.Node.select_action

    def run_batch_wrap(self, states, to_plays):
        roots = orig_run_batch(self, states, to_plays)
        A = self.game.get_action_size()
        counts = []
        for r in roots:
            c = [0] * A
            for a, ch in r.children.items():
                c[a] = int(ch.visit_count)
            counts.append(c)
        plies.append({"counts": counts, "root_w": [float(r.value_sum) for r in roots],
                      "actions": []})
        return roots

    def select_wrap(self, temperature):
        a = orig_select(self, temperature)
        plies[-1]["actions"].append(int(a))
        return a

    ref_mcts.MCTS.run_batch = run_batch_wrap
    ref_mcts.Node.select_action = select_wrap
    try:
        np.random.seed(seed)
        cfg = AlphaZeroConfig(batch_size=32, num_iterations=1, num_simulations=sims,
                              num_episodes=n, num_iters_for_train_history=1, epochs=1)
        gen = EpisodeGenerator(game=game, config=cfg)
        episodes = []
        for ep in gen.generate_episodes(model):
            episodes.append({
                "states": [np.asarray(s.state).tolist() for s in ep.samples],
                "policies": [np.asarray(s.policy, dtype=np.float64).tolist() for s in ep.samples],
                "values": [int(s.value) for s in ep.samples],
                "value_types": sorted({type(s.value).__name__ for s in ep.samples}),
                "ply": len(plies) - 1,
            })
        rng_after = float(np.random.random_sample())
    finally:
        ref_mcts.MCTS.run_batch = orig_run_batch
        ref_mcts.Node.select_action = orig_select
    return {"episodes": episodes, "plies": plies, "rng_after": rng_after}


def episodes_hash():
    runs = []
    for gname, Game, Rules, cfgs in (
            ("connect4", Connect4Game, C4Rules,
             [(1, 25, 0, 0), (4, 25, 1, 0), (4, 100, 2, 99), (16, 50, 3, 5), (64, 100, 4, 0),
              (2, 400, 5, 7)]),
            ("connect2", Connect2Game, C2Rules, [(1, 10, 0, 0), (8, 25, 1, 3)])):
        game, rules = Game(), Rules()
        for n, sims, seed, salt in cfgs:
            out = run_reference_generator(game, HashModel(rules, salt), sims, n, seed)
            out.update({"game": gname, "n": n, "sims": sims, "seed": seed, "salt": salt})
            runs.append(out)
            print(f"  episodes_hash {gname} n={n} sims={sims}: {len(out['episodes'])} episodes, "
                  f"{len(out['plies'])} plies")
    dump("episodes_hash.json", {"runs": runs})


# ---------------------------------------------------------------------------------------------
def randomise_bn(model, gen):
    """Non-trivial eval-mode BatchNorm statistics so that BN folding is actually exercised."""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) * 1.5 + 0.25)
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)


def resnet_forward():
    game = Connect4Game()
    rng = np.random.RandomState(11)
    boards = [b for b, _ in ([(game.get_init_board(), 1)] + midgame_boards(game, rng, 47, 41))]
    out = {"boards": np.stack(boards).astype(np.int8)}
    for tag, blocks, ch, seed in (("s", 2, 16, 3), ("m", 4, 64, 0)):
        torch.manual_seed(seed)
        net = ResNet((6, 7), 7, blocks, ch)
        if tag == "s":
            randomise_bn(net, torch.Generator().manual_seed(5))
        net.eval()
        torch.set_num_threads(1)
        p, v = net.predict(boards)
        with torch.no_grad():
            logits, vlogit = net.forward(net._states_to_tensor(boards))
        out[f"{tag}_p"] = np.stack(p).astype(np.float32)
        out[f"{tag}_v"] = np.array(v, dtype=np.float32)
        out[f"{tag}_logits"] = logits.numpy().astype(np.float32)
        out[f"{tag}_vlogit"] = vlogit.numpy().astype(np.float32)
        out[f"{tag}_arch"] = np.array([blocks, ch], dtype=np.int32)
        for k, t in net.state_dict().items():
            out[f"{tag}_sd/{k}"] = t.numpy()
    path = os.path.join(HERE, "resnet_forward.npz")
    np.savez(path, **out)
    print(f"wrote resnet_forward.npz: {os.path.getsize(path) / 1024:.1f} KiB")


def episodes_resnet():
    """BASELINE.json configs[0]: Connect-4, random-weight ResNet(4 blocks, 64 ch), 25 sims, N=1."""
    torch.manual_seed(0)
    torch.set_num_threads(1)
    net = ResNet((6, 7), 7, 4, 64)
    out = run_reference_generator(Connect4Game(), net, 25, 1, 0)
    out.update({"game": "connect4", "n": 1, "sims": 25, "seed": 0, "torch_seed": 0,
                "arch": [4, 64], "torch": torch.__version__, "numpy": np.__version__})
    dump("episodes_resnet.json", out)


def dataset_and_wire():
    """What sits right after the hot path, produced by the reference's own code on the
    episodes of one reference generator run (Connect-4, hash evaluator):
      * the JSON text DataModule._save_episodes writes: json.dump([episode.to_dict() ...]) (datamodule.py:87-96,
        episode.py:15-24,64-66), and what Episode.from_dict (the loader datamodule.py:98-103 uses) makes of it;
      * the TensorDataset BaseModel.format_dataset builds from the flattened samples exactly as
        DataModule.train_dataloader does (datamodule.py:134-142, base_model.py:61-67, resnet.py:115-128)."""
    from alphazero_less_simple.core.training.episode import Episode
    n, sims, seed, salt = 16, 50, 3, 5
    game = Connect4Game()
    np.random.seed(seed)
    cfg = AlphaZeroConfig(batch_size=32, num_iterations=1, num_simulations=sims, num_episodes=n,
                          num_iters_for_train_history=1, epochs=1)
    episodes = list(EpisodeGenerator(game=game, config=cfg).generate_episodes(HashModel(C4Rules(), salt)))
    rng_after = float(np.random.random_sample())
    text = json.dumps([ep.to_dict() for ep in episodes])          # json.dump(obj, f) writes exactly this text
    with open(os.path.join(HERE, "episodes_wire.json"), "w") as f:
        json.dump([ep.to_dict() for ep in episodes], f)
    with open(os.path.join(HERE, "episodes_wire.json")) as f:
        assert f.read() == text
    back = [Episode.from_dict(d) for d in json.loads(text)]
    loaded = {"state_dtype": str(back[0].samples[0].state.dtype), "policy_dtype": str(back[0].samples[0].policy.dtype),
              "value_type": type(back[0].samples[0].value).__name__,
              "state_shape": list(back[0].samples[0].state.shape), "lengths": [len(e) for e in back]}
    samples = [s for ep in episodes for s in ep.samples]
    torch.manual_seed(0)
    ds = ResNet((6, 7), 7, 1, 16).format_dataset([s.state for s in samples], [s.policy for s in samples],
                                                 [s.value for s in samples])
    planes, pol, val = (t.numpy() for t in ds.tensors)
    meta = {"game": "connect4", "n": n, "sims": sims, "seed": seed, "salt": salt, "rng_after": rng_after,
            "loaded": loaded, "dtypes": [str(t.dtype) for t in ds.tensors], "shapes": [list(t.shape) for t in ds.tensors]}
    path = os.path.join(HERE, "format_dataset.npz")
    np.savez_compressed(path, planes=planes, policies=pol, values=val,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(f"wrote episodes_wire.json: {len(text) / 1024:.1f} KiB; format_dataset.npz: {os.path.getsize(path) / 1024:.1f} KiB, "
          f"{len(samples)} samples, dtypes {meta['dtypes']}")


def gomoku_hash():
    """Gomoku 9x9 is not in the reference; its MCTS and EpisodeGenerator are game-agnostic, so they are run
    here with the NEW GomokuGame plugin (alphazero-implementation_amd/azsp/games.py) and the hash evaluator:
    the search / generator logic with 81 actions (numpy pairwise np.sum over 81 priors, etc.) is the reference's."""
    sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
    from azsp.games import GomokuGame
    game, rules = GomokuGame(), G9Rules()
    rng = np.random.RandomState(21)
    boards = [(game.get_init_board(), 1)] + midgame_boards(game, rng, 5, 40)
    runs = []
    for salt, sims in ((0, 1), (0, 30), (4242, 120)):
        m = ref_mcts.MCTS(game, HashModel(rules, salt), sims)
        roots = m.run_batch([b for b, _ in boards], [p for _, p in boards])
        runs.append({"game": "gomoku9", "salt": salt, "sims": sims, "boards": [b.tolist() for b, _ in boards],
                     "to_plays": [int(p) for _, p in boards], "roots": [node_dump(r, 81) for r in roots]})
    eps = []
    for n, sims, seed, salt in ((1, 10, 0, 1), (3, 24, 1, 2)):
        out = run_reference_generator(game, HashModel(rules, salt), sims, n, seed)
        out.update({"game": "gomoku9", "n": n, "sims": sims, "seed": seed, "salt": salt})
        eps.append(out)
        print(f"  gomoku episodes n={n} sims={sims}: {len(out['episodes'])} episodes, {len(out['plies'])} plies")
    dump("gomoku_hash.json", {"search": runs, "episodes": eps})


if __name__ == "__main__":
    if "--gomoku" in sys.argv:
        gomoku_hash()
        sys.exit(0)
    if "--dataset" in sys.argv:
        dataset_and_wire()
        sys.exit(0)
    rules_connect4()
    rules_connect2()
    search_hash()
    episodes_hash()
    resnet_forward()
    episodes_resnet()
    dataset_and_wire()
    gomoku_hash()
