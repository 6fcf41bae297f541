"""First-generation surface of the reference (`src/alphazero_implementation`) on the device engine.

Mirrors, name for name (paths relative to src/alphazero_implementation):
  Model                 models/base/model.py:15-96         predict(states) -> (list[dict[Action, float]], list[list[float]])
  Sample, Episode       core/training/episode.py:9-71      state: State, policy: dict[Action, float], value: list[float]
  Node                  core/search/mcts/node.py:7-73      (a view: the root and its first-level children)
  AlphaZeroSearch       core/search/mcts/search.py:10-91   run(root) / run_simulations(nodes) / update_inference_model
  EpisodeGenerator      core/training/episode_generator.py:12-81
  Player, AlphaZeroPlayer  ui/cli/player.py:10-76, and a head-less `play_match` for ui/cli/arena.py:42-57

The trees, the PUCT descent, the backups and the move sampling run in libazsp under AZSP_RULES_GEN1 (the
arithmetic of search.py / node.py, include/azsp.h); this module only converts between `State` / `Action` objects
and the engine's canonical boards.  States are duck-typed: anything with the attributes listed in
azsp/connect_state.py works (`simulator.game.connect` itself, where installed).  Leaf positions handed to a host
`Model.predict` are built with `state_factory(grid, player)`; it defaults to azsp.connect_state.State.

`simulator` is not installable in the build environment, so nothing here could be compared with the reference
running: the rule set is tested against a restatement written from the source text (oracle/gen1_oracle.py,
"parity unpinned")."""
from __future__ import annotations

import copy
import random
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Any, Callable, Generator

import numpy as np

from . import _lib as L
from . import connect_state as cs
from .engine import Engine
from .mcts import default_device
from .weights import is_resnet_state_dict, resnet_blob

ActionPolicy = dict        # dict[Action, float]   (models/base/prediction_types.py:6)
Value = list               # list[float], one entry per player (prediction_types.py:13)


# ------------------------------------------------------------------------------------------------
# State <-> canonical board
# ------------------------------------------------------------------------------------------------
def canonical_cells(state) -> np.ndarray:
    """gen-1 grid (row 0 = bottom, -1 empty, 0/1 = player) -> int8 cells, row 0 = top, +1 = side to move."""
    g = np.asarray(state.grid)[::-1]
    out = np.zeros(g.shape, np.int8)
    out[g == state.player] = 1
    out[g == 1 - state.player] = -1
    return out


def grid_from_cells(cells: np.ndarray, player: int) -> np.ndarray:
    g = np.full(cells.shape, -1, np.int8)
    g[cells == 1] = player
    g[cells == -1] = 1 - player
    return g[::-1].copy()


def _check_connect4(state) -> None:
    c = state.config
    if (c.height, c.width, getattr(c, "count", 4), getattr(c, "num_players", 2)) != (6, 7, 4, 2):
        raise NotImplementedError("the device trait behind the gen-1 surface is 6x7 Connect-4 for two players "
                                  "(the reference trains on Config(6, 7, 4), scripts/train.py:12); no CPU fallback")


def _action_for(state, column: int):
    for a in state.actions:
        if a.column == column:
            return a
    raise ValueError(f"column {column} is not legal in {state!r}")


# ------------------------------------------------------------------------------------------------
# records (core/training/episode.py:9-71)
# ------------------------------------------------------------------------------------------------
@dataclass
class Sample:
    state: Any
    policy: ActionPolicy
    value: Value

    def to_dict(self) -> dict[str, Any]:
        return {"state": self.state.to_json(),
                "policy": {str(a.to_json()): p for a, p in self.policy.items()},
                "value": self.value}


@dataclass
class Episode:
    samples: list[Sample] = field(default_factory=list)

    def __len__(self) -> int:
        return len(self.samples)

    def add_sample(self, sample: Sample) -> None:
        self.samples.append(sample)

    def backpropagate_outcome(self, value: Value) -> None:
        for s in self.samples:
            s.value = value

    @property
    def current_state(self):
        return self.samples[-1].state

    def to_dict(self) -> dict[str, Any]:
        return {"samples": [s.to_dict() for s in self.samples]}


# ------------------------------------------------------------------------------------------------
# model plugin (models/base/model.py:15-96, models/games/connect4/model.py:9-51)
# ------------------------------------------------------------------------------------------------
class Model(ABC):
    """The abstract method set of the reference's `Model`; training members are left to subclasses (the
    reference mixes in LightningModule, which the self-play path never touches)."""

    @abstractmethod
    def predict(self, states: list) -> tuple[list[ActionPolicy], list[Value]]: ...

    def get_inference_clone(self):
        clone = copy.deepcopy(self)
        if hasattr(clone, "eval"):
            clone.eval()
        return clone


class HashModel(Model):
    """Deterministic (P, V) from the engine's integer-hash evaluator; parity tests only."""

    def __init__(self, salt: int = 0):
        self.azsp_hash_salt = int(salt)

    def predict(self, states):
        raise RuntimeError("HashModel is evaluated on the device (AZSP_EVAL_HASH); it has no host predict")


class ResNetModel(Model):
    """A gen-1 `Model` around the reference's ResNet (src/alphazero_simple/resnet.py:30-157) evaluated by the
    fused HIP kernel: policy = softmax over the legal columns as a dict over `state.actions`
    (models/games/connect4/model.py:27-40), value = [v, -v] seen from the side to move, so that
    `value[state.player]` (search.py:91) is the network's v for the player to move."""

    def __init__(self, net):
        self.net = net          # azsp.model.ResNet (or any module with its state_dict layout and .predict)

    def state_dict(self):
        return self.net.state_dict()

    def load_state_dict(self, sd):
        return self.net.load_state_dict(sd)

    def eval(self):
        self.net.eval()
        return self

    def predict(self, states):
        p, v = self.net.predict([canonical_cells(s).astype(np.int64) for s in states])
        pols = [{a: float(p[i][a.column]) for a in s.actions} for i, s in enumerate(states)]
        vals = [[float(x), -float(x)] if s.player == 0 else [-float(x), float(x)] for x, s in zip(v, states)]
        return pols, vals


# ------------------------------------------------------------------------------------------------
# search (core/search/mcts/node.py, search.py)
# ------------------------------------------------------------------------------------------------
class Node:
    def __init__(self, state, parent: "Node | None" = None, prior: float = 0.0):
        self.state, self.parent, self.prior = state, parent, prior
        self.children: dict = {}
        self.visit_count = 0
        self.value_sum = 0.0

    @property
    def raw_policy(self) -> ActionPolicy:
        return {a: c.prior for a, c in self.children.items()}

    @property
    def improved_policy(self) -> ActionPolicy:
        return {a: c.visit_count / (self.visit_count - 1) for a, c in self.children.items()}

    def select_next_node(self) -> "Node":
        pol = self.improved_policy
        index = np.random.choice(len(pol), p=list(pol.values()))
        action = list(pol.keys())[index]
        return Node(state=action.sample_next_state(), parent=self, prior=pol[action])

    @property
    def value(self) -> float:
        return 0 if self.visit_count == 0 else self.value_sum / self.visit_count

    @property
    def is_expanded(self) -> bool:
        return len(self.children) > 0

    @property
    def is_terminal(self) -> bool:
        return self.state.has_ended

    @property
    def utility_values(self) -> list[float]:
        return self.state.reward.tolist()

    @property
    def is_root(self) -> bool:
        return self.parent is None


class AlphaZeroSearch:
    def __init__(self, *, model, num_simulations: int, exploration_weight: float = 1.0, device: int | None = None,
                 state_factory: Callable | None = None, timing: bool = False):
        self.num_simulations = int(num_simulations)
        self.exploration_weight = float(exploration_weight)
        self.device, self.timing, self.state_factory = device, timing, state_factory
        self._engine: Engine | None = None
        self._engine_key = None
        self._p0 = 0
        self._base_pieces = 0
        self.update_inference_model(model)

    # ---- model ----------------------------------------------------------------------------------
    def update_inference_model(self, model) -> None:
        """search.py:22-25.  Device networks are re-uploaded from the model's current weights; any other model is
        cloned the way the reference clones it."""
        self._weights = None
        if hasattr(model, "azsp_hash_salt"):
            self.inference_model = model
        elif hasattr(model, "state_dict") and is_resnet_state_dict(model.state_dict()):
            self.inference_model = model
            self._weights = resnet_blob(model.state_dict())          # (blob, blocks, channels) snapshot
        else:
            self.inference_model = model.get_inference_clone() if hasattr(model, "get_inference_clone") else model
        self._dirty = True

    def _geometry(self):
        if hasattr(self.inference_model, "azsp_hash_salt"):
            return 0, 0, int(self.inference_model.azsp_hash_salt)
        if self._weights is not None and 1 <= (self._weights[2] + 31) // 32 <= 8:
            return self._weights[1], self._weights[2], 0
        return 0, 0, 0

    def engine_for(self, n: int, exact: bool = False) -> Engine:
        blocks, ch, salt = self._geometry()
        key = (blocks, ch, salt)
        e = self._engine
        if e is None or key != self._engine_key or (e.n_slots != n if exact else e.n_slots < n):
            if e is not None:
                e.close()
            dev = default_device() if self.device is None else self.device
            e = self._engine = Engine("connect4", n if exact else max(n, 64), self.num_simulations, device=dev,
                                      res_blocks=blocks, channels=ch, hash_salt=salt, timing=self.timing,
                                      search_rules=L.RULES_GEN1)
            self._engine_key = key
            self._dirty = True
        if self._dirty:
            if salt or hasattr(self.inference_model, "azsp_hash_salt"):
                e.use_hash_evaluator()
            elif blocks:
                e.load_resnet(self._weights[0])
            else:
                e.use_host_evaluator(self._host_predict)
            self._dirty = False
        e.set_search_options(c_init=self.exploration_weight)
        return e

    def _host_predict(self, boards: list[np.ndarray]):
        """Engine leaves (canonical cells) -> State objects -> Model.predict -> (float32[A], value for the side to move)."""
        make = self.state_factory or self._default_factory
        states = []
        for b in boards:
            pieces = int(np.count_nonzero(b))
            player = (self._p0 + pieces - self._base_pieces) % 2
            states.append(make(grid_from_cells(np.asarray(b, np.int8), player), player))
        pols, vals = self.inference_model.predict(states)
        p = np.zeros((len(states), 7), np.float32)
        v = np.empty(len(states), np.float32)
        for i, (s, pol, val) in enumerate(zip(states, pols, vals)):
            for a, pr in pol.items():
                p[i, a.column] = pr
            v[i] = val[s.player]                                    # search.py:91
        return p, v

    _config = None

    def _default_factory(self, grid, player):
        return cs.State(self._config or cs.Config(), grid, player)

    def _anchor(self, state) -> None:
        """Remember whose turn it is for a given number of pieces (the host evaluator rebuilds `player` from it)."""
        _check_connect4(state)
        self._p0 = int(state.player)
        self._base_pieces = int(np.count_nonzero(np.asarray(state.grid) != -1))
        c = state.config
        self._config = cs.Config(c.height, c.width, getattr(c, "count", 4), getattr(c, "num_players", 2))

    # ---- search.py:60-91 --------------------------------------------------------------------------
    def run_simulations(self, current_nodes: list[Node]) -> None:
        if not current_nodes:
            return
        if any(n.state.has_ended for n in current_nodes):
            raise ValueError("cannot search from a finished game")
        if any(n.visit_count or n.children for n in current_nodes):
            # search.py:66-91 would descend the node's existing tree and add num_simulations more visits to it; the
            # device search always starts from an unexpanded root, so a second run on the same Node is refused rather
            # than silently restarted (improved_policy = n / (N - 1) would no longer sum to 1)
            raise ValueError("run_simulations needs fresh nodes (visit_count == 0, no children): the device search "
                             "does not continue an existing host-side tree; wrap the state in a new Node")
        parity = {(int(n.state.player) - int(np.count_nonzero(np.asarray(n.state.grid) != -1))) % 2 for n in current_nodes}
        if len(parity) != 1:
            raise ValueError("states of one batch must agree on which player moves at an even number of pieces")
        self._anchor(current_nodes[0].state)
        eng = self.engine_for(len(current_nodes))
        out = eng.run_batch(np.stack([canonical_cells(n.state) for n in current_nodes]))
        val = eng.root_values(len(current_nodes))
        for i, node in enumerate(current_nodes):
            node.visit_count = int(val["root_n"][i])
            node.value_sum = float(val["root_w"][i])
            node.children = {}
            for a in node.state.actions:
                child = Node(None, node, float(out["prior"][i, a.column]))
                child.visit_count = int(out["counts"][i, a.column])
                child.value_sum = float(val["child_w"][i, a.column])
                node.children[a] = child

    def run(self, root: Node) -> tuple[ActionPolicy, float]:
        self.run_simulations([root])
        return root.improved_policy, root.value


# ------------------------------------------------------------------------------------------------
# self-play (core/training/episode_generator.py:12-81)
# ------------------------------------------------------------------------------------------------
class EpisodeGenerator:
    def __init__(self, *, model, num_simulations: int, num_episodes: int, game_initial_state,
                 exploration_weight: float = 1.0, device: int | None = None, state_factory: Callable | None = None,
                 timing: bool = False):
        self.search = AlphaZeroSearch(model=model, num_simulations=num_simulations,
                                      exploration_weight=exploration_weight, device=device,
                                      state_factory=state_factory, timing=timing)
        self.num_episodes = int(num_episodes)
        self.game_initial_state = game_initial_state
        self.num_players = game_initial_state.config.num_players
        self.last_stats: dict | None = None

    def update_inference_model(self, model) -> None:
        self.search.update_inference_model(model)

    def generate_episodes(self, initial_state=None) -> Generator[Episode, None, None]:
        """Yields `Episode`s in the reference's (ply, slot) completion order and stops at the num_episodes-th one;
        consumes the global numpy RNG like the reference's np.random.choice calls (one double per slot per ply)."""
        if initial_state is None:
            initial_state = self.game_initial_state
        if initial_state.has_ended:
            raise ValueError("the initial state is terminal")
        n = self.num_episodes
        self.search._anchor(initial_state)
        eng = self.search.engine_for(n, exact=True)
        init_cells = canonical_cells(initial_state)
        eng.set_initial_board(init_cells if init_cells.any() else None)
        eng.reset()
        emitted = 0
        eng.search()
        while True:
            rng_state = np.random.get_state()
            res = eng.play(np.random.random_sample(n))
            if res.uniforms_used < n:            # the reference returns inside its slot loop (:80-81)
                np.random.set_state(rng_state)
                np.random.random_sample(res.uniforms_used)
            if not res.finished:
                eng.search()
            if res.n_finished:
                out = eng.fetch_episodes(emitted, res.n_finished)
                off = out["ep_offset"]
                moves = eng.fetch_moves(emitted, res.n_finished, int(off[-1]))
                for e in range(res.n_finished):
                    yield self._assemble(initial_state, out["policies"][off[e]:off[e + 1]], moves[off[e]:off[e + 1]])
                emitted += res.n_finished
            if res.finished:
                self.last_stats = eng.stats()
                return

    @staticmethod
    def _assemble(initial_state, policies: np.ndarray, moves: np.ndarray) -> Episode:
        """Rebuild the episode's State objects by replaying its moves through the State API, exactly the objects
        the reference's nodes would hold (node.py:37-41); the outcome vector is the terminal state's reward
        (episode_generator.py:66, episode.py:52-54)."""
        ep = Episode()
        state = initial_state
        for pol, col in zip(policies, moves):
            ep.add_sample(Sample(state=state, policy={a: float(pol[a.column]) for a in state.actions}, value=[]))
            state = _action_for(state, int(col)).sample_next_state()
        if not state.has_ended:
            raise RuntimeError("engine and State API disagree on the end of the game")
        ep.backpropagate_outcome(state.reward.tolist())
        return ep


def generate_dataset(gen: EpisodeGenerator, initial_state=None, device=None):
    """One generation of `gen` without materialising State / Episode objects: the three training tensors that
    `Model.format_dataset` (models/base/model.py:76-82) builds for the reference's CNNModel
    (`_states_to_tensor` models/games/connect4/cnn.py:77-98: planes [empty, side to move, opponent] in the grid's own
    row order, row 0 = bottom; `_policies_to_tensor` models/games/connect4/model.py:45-51; values = the outcome
    vector per player), written on the GPU by the engine's export kernel and re-oriented with torch ops:
    TensorDataset(planes float32[S,3,6,7], policies float32[S,7], values float32[S,2])."""
    import torch
    from torch.utils.data import TensorDataset
    if initial_state is None:
        initial_state = gen.game_initial_state
    n = gen.num_episodes
    gen.search._anchor(initial_state)
    eng = gen.search.engine_for(n, exact=True)
    init_cells = canonical_cells(initial_state)
    eng.set_initial_board(init_cells if init_cells.any() else None)
    eng.reset()
    eng.search()
    while True:
        rng_state = np.random.get_state()
        res = eng.play(np.random.random_sample(n))
        if res.uniforms_used < n:
            np.random.set_state(rng_state)
            np.random.random_sample(res.uniforms_used)
        if res.finished:
            break
        eng.search()
    gen.last_stats = eng.stats()
    device = eng.torch_device(device)          # the export kernel writes on the engine's own GPU
    s = int(res.samples_total)
    planes = torch.empty((s, 3, eng.rows, eng.cols), dtype=torch.float32, device=device)
    pol = torch.empty((s, eng.n_actions), dtype=torch.float32, device=device)
    z = torch.empty((s,), dtype=torch.float32, device=device)
    eng.sync_torch_stream()
    got = eng.export_dataset_device(0, n, planes.data_ptr(), pol.data_ptr(), z.data_ptr(), s)
    assert got == s
    # the engine's z is the outcome for the sample's side to move; gen-1 stores the vector over absolute players
    off = eng.fetch_episodes(0, n, arrays=False)["ep_offset"]
    ply = np.concatenate([np.arange(off[e + 1] - off[e]) for e in range(n)])
    mover0 = torch.as_tensor(((int(initial_state.player) + ply) % 2) == 0, device=device)
    values = torch.stack([torch.where(mover0, z, -z), torch.where(mover0, -z, z)], dim=1)
    return TensorDataset(torch.flip(planes, dims=[2]), pol, values)


# ------------------------------------------------------------------------------------------------
# evaluation players (ui/cli/player.py:10-76) and a head-less arena (ui/cli/arena.py:42-57)
# ------------------------------------------------------------------------------------------------
class Player(ABC):
    @abstractmethod
    def play(self, state): ...


class RandomPlayer(Player):
    def play(self, state):
        return random.choice(state.actions)


class AlphaZeroPlayer(Player):
    """player.py:18-76.  `play` searches one state; `play_many` searches a batch of states in one engine call
    (the arena's games are independent, so they share launches)."""

    def __init__(self, model, *, mcts_simulation: int = 100, temperature: float = 1.0, device: int | None = None,
                 state_factory: Callable | None = None) -> None:
        self.model = model
        self.mcts_simulation = mcts_simulation
        self.temperature = temperature
        self._search = AlphaZeroSearch(model=model, num_simulations=mcts_simulation, device=device,
                                       state_factory=state_factory)

    def _choose(self, policy: ActionPolicy):
        if self.temperature == 0:
            return max(policy.items(), key=lambda x: x[1])[0]
        probs = [p ** (1 / self.temperature) for p in policy.values()]
        total = sum(probs)
        return random.choices(list(policy.keys()), weights=[p / total for p in probs])[0]

    def play(self, state):
        if self.temperature == float("inf"):
            return random.choice(state.actions)
        policy, _ = self._search.run(Node(state))
        return self._choose(policy)

    def play_many(self, states: list) -> list:
        if self.temperature == float("inf"):
            return [random.choice(s.actions) for s in states]
        nodes = [Node(s) for s in states]
        self._search.run_simulations(nodes)
        return [self._choose(n.improved_policy) for n in nodes]


def play_match(player1: Player, player2: Player, initial_state, num_games: int) -> list:
    """arena.py:42-57 without the TUI: `num_games` games from `initial_state`, player1 moving when
    state.player == 0; all unfinished games advance together, one batched search per side per ply.  Returns the
    final states (winner = state.reward.argmax(), arena.py:56)."""
    states = [initial_state for _ in range(num_games)]
    while True:
        live = [i for i, s in enumerate(states) if not s.has_ended]
        if not live:
            return states
        for who, player in ((0, player1), (1, player2)):
            idx = [i for i in live if states[i].player == who and not states[i].has_ended]
            if not idx:
                continue
            batch = [states[i] for i in idx]
            acts = player.play_many(batch) if hasattr(player, "play_many") else [player.play(s) for s in batch]
            for i, a in zip(idx, acts):
                states[i] = a.sample_next_state()
