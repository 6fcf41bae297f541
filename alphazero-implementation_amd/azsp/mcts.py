"""Search facade (reference: src/alphazero_simple/monte_carlo_tree_search.py:25-171).

`MCTS(game, model, num_simulations).run / run_batch` run the whole search on the device and return
light `Node` views: the root and its first-level children, which is what the reference's callers
read (`root.children[a].visit_count`, `root.value()`, `root.select_action(T)`)."""
from __future__ import annotations

import os

import numpy as np

from .engine import Engine, game_id_of
import weakref

from .weights import is_resnet_state_dict, resnet_blob, resnet_geometry, weights_fingerprint


class Node:
    def __init__(self, prior: float, to_play: int, parent: "Node | None" = None):
        self.visit_count: int = 0
        self.to_play: int = to_play
        self.prior = prior
        self.value_sum = 0.0
        self.children: dict[int, Node] = {}
        self.state: np.ndarray | None = None
        self.parent = parent

    def expanded(self) -> bool:
        return len(self.children) > 0

    def value(self):
        return 0 if self.visit_count == 0 else self.value_sum / self.visit_count

    def select_action(self, temperature: float) -> int:
        """Visit-count sampling with the reference's three temperature regimes; consumes the global
        numpy RNG exactly like the reference (one np.random.choice call unless temperature == 0)."""
        counts = np.array([c.visit_count for c in self.children.values()])
        actions = list(self.children.keys())
        if temperature == 0:
            return actions[int(np.argmax(counts))]
        if temperature == float("inf"):
            return np.random.choice(actions)
        dist = counts ** (1 / temperature)
        dist = dist / sum(dist)
        return np.random.choice(actions, p=dist)

    def __repr__(self):
        return f"Node(prior={float(self.prior):.2f}, count={self.visit_count}, value={self.value()})"


def configure_engine_for_model(engine: Engine, model) -> None:
    """Pick the evaluator that stands behind `model.predict`:
    ResNet-layout state_dict -> on-device MFMA inference; `azsp_hash_salt` attribute -> hash evaluator
    (parity tests); anything else with .predict -> host callback."""
    if hasattr(model, "azsp_hash_salt"):
        engine.use_hash_evaluator()
        return
    sd = model.state_dict() if hasattr(model, "state_dict") else None
    if sd is not None and is_resnet_state_dict(sd) and engine.channels > 0:
        engine.load_resnet(resnet_blob(sd)[0])
        return
    engine.use_host_evaluator(model.predict)


class WeightsOnDevice:
    """Remembers which model's weights an engine holds, so that they are exported (22 ms for 4 x 64: 142 tensors) and
    uploaded again only when they changed: storage address + in-place version counter of every tensor
    (`weights_fingerprint`), bound to the model object by a weak reference (a freed model's addresses get reused).  Models
    that are not torch modules are configured on every call."""

    def __init__(self):
        self._fp = None
        self._of = None
        self._engine_id = None

    def invalidate(self) -> None:
        self._fp = None

    def ensure(self, engine: Engine, model) -> None:
        fp = weights_fingerprint(model)
        same = self._of is not None and self._of() is model and self._engine_id == id(engine)
        if fp is None or not same or fp != self._fp:
            configure_engine_for_model(engine, model)
            self._fp, self._engine_id = fp, id(engine)
            try:
                self._of = weakref.ref(model)
            except TypeError:               # (objects without weak references: configured every time)
                self._of = None


def engine_geometry_for_model(model, game=None) -> tuple[int, int, int]:
    """(res_blocks, channels, hash_salt) to create an engine for this model.  The device network exists for
    Connect-4 (the reference's ResNet, resnet.py:30-157) and, with the same architecture, for the 9x9 Gomoku plugin;
    other games evaluate through the host callback."""
    if hasattr(model, "azsp_hash_salt"):
        return 0, 0, int(model.azsp_hash_salt)
    sd = model.state_dict() if hasattr(model, "state_dict") else None
    if game is not None and game_id_of(game) not in (0, 2):
        return 0, 0, 0
    if sd is not None and is_resnet_state_dict(sd):
        blocks, ch = resnet_geometry(sd)
        if 1 <= (ch + 31) // 32 <= 8:
            return blocks, ch, 0
    return 0, 0, 0


def default_device() -> int:
    return int(os.environ.get("AZSP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


class MCTS:
    def __init__(self, game, model, num_simulations: int, *, eval_cache: bool = False):
        self.game = game
        self.model = model
        self.num_simulations = num_simulations
        # optional (not in the reference's signature): keep every evaluated position's (p, v) on the device across run_batch
        # calls on the same weights -- same roots, fewer evaluations (AZSP_FLAG_EVAL_CACHE; device evaluators only)
        self.eval_cache = eval_cache
        self._engine: Engine | None = None
        self._weights = WeightsOnDevice()

    def invalidate_device_weights(self) -> None:
        """After weight updates that by-pass the tensors' version counters (writes through `.data`, `set_()`)."""
        self._weights.invalidate()

    def _engine_for(self, n: int) -> Engine:
        if self._engine is None or self._engine.n_slots < n:
            if self._engine is not None:
                self._engine.close()
            blocks, ch, salt = engine_geometry_for_model(self.model, self.game)
            self._engine = Engine(self.game, max(n, 64), self.num_simulations, device=default_device(),
                                  res_blocks=blocks, channels=ch, hash_salt=salt, eval_cache=self.eval_cache)
            self._weights.invalidate()          # a new engine holds no weights
        self._weights.ensure(self._engine, self.model)
        return self._engine

    def run(self, state: np.ndarray, to_play: int) -> Node:
        return self.run_batch([state], [to_play])[0]

    def run_batch(self, states: list[np.ndarray], to_plays: list[int]) -> list[Node]:
        eng = self._engine_for(len(states))
        out = eng.run_batch(np.stack([np.asarray(s) for s in states]))
        roots = []
        for i, (st, tp) in enumerate(zip(states, to_plays)):
            root = Node(0, tp)
            root.state = st
            root.visit_count = int(out["root_n"][i])
            root.value_sum = out["root_w"][i]
            for a in range(eng.n_actions):
                if out["prior"][i, a] != 0:
                    c = Node(out["prior"][i, a], -tp, root)
                    c.visit_count = int(out["counts"][i, a])
                    c.value_sum = out["child_w"][i, a]
                    root.children[a] = c
            roots.append(root)
        return roots

    def run_single(self, state: np.ndarray, to_play: int) -> Node:
        """The reference's unbatched search (monte_carlo_tree_search.py:173-212): one tree, one `predict` call per
        simulation.  It builds the same tree as `run_batch` on a one-element list (SURVEY.md appendix A.11 checked
        identical visit counts), so it is the same device search."""
        return self.run_batch([state], [to_play])[0]
