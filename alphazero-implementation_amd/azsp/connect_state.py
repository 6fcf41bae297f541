"""Stand-alone Connect-k states with the attribute surface the reference's first-generation package reads
from `simulator.game.connect` (board-game-simulator-python@0.0.4, un-vendored: requirements.txt:9).

Call sites that define the surface (paths relative to src/alphazero_implementation):
  State.grid / .player / .config.{height,width}      models/games/connect4/cnn.py:77-98, basic.py:40-47
  State.actions, Action.column                      models/games/connect4/model.py:27-40,45-51
  Action.sample_next_state()                        core/search/mcts/search.py:89, node.py:38
  State.has_ended, State.reward (ndarray, .tolist) core/search/mcts/node.py:61-68, ui/cli/arena.py:56
  State.config.num_players                          core/training/episode_generator.py:29
  to_json / from_json                               core/training/episode.py:20-36
  Config(h, w, k).sample_initial_state()            scripts/train.py:12-13

It lets the gen-1 adapter (azsp/gen1.py) run where `simulator` is not installed, and it is what the adapter's
`state_factory` defaults to.  Conventions (grid -1 empty / 0 / 1, row 0 = bottom, reward +1 / -1 / 0) follow the
comments in notebooks/policy_comparison.ipynb cell 6; `simulator` itself is absent, so they are assumptions and the
gen-1 surface stays "parity unpinned" (DESIGN.md)."""
from __future__ import annotations

import numpy as np


class Config:
    def __init__(self, height: int = 6, width: int = 7, count: int = 4, num_players: int = 2):
        self.height, self.width, self.count, self.num_players = int(height), int(width), int(count), int(num_players)

    def sample_initial_state(self) -> "State":
        return State(self, np.full((self.height, self.width), -1, dtype=np.int8), 0)

    def to_json(self) -> dict:
        return {"height": self.height, "width": self.width, "count": self.count, "num_players": self.num_players}

    @classmethod
    def from_json(cls, data: dict) -> "Config":
        return cls(data["height"], data["width"], data["count"], data.get("num_players", 2))

    def __eq__(self, other):
        return isinstance(other, Config) and self.to_json() == other.to_json()

    def __hash__(self):
        return hash((self.height, self.width, self.count, self.num_players))


class Action:
    def __init__(self, state: "State", column: int):
        self.state, self.column = state, int(column)

    def sample_next_state(self) -> "State":
        s = self.state
        h = s.config.height
        free = int(np.count_nonzero(s.grid[:, self.column] != -1))      # row 0 = bottom: lowest free row
        if free >= h:
            raise ValueError(f"column {self.column} is full")
        grid = s.grid.copy()
        grid[free, self.column] = s.player
        bits = list(s._bits)
        bits[s.player] |= 1 << (self.column * (h + 1) + free)
        return State(s.config, grid, 1 - s.player, _bits=tuple(bits))

    def to_json(self) -> dict:
        return {"state": self.state.to_json(), "column": self.column}

    @classmethod
    def from_json(cls, data: dict) -> "Action":
        return cls(State.from_json(data["state"]), data["column"])

    def __eq__(self, other):
        return isinstance(other, Action) and self.column == other.column and self.state == other.state

    def __hash__(self):
        return hash((self.column, self.state))

    def __repr__(self):
        return f"Action(column={self.column})"


class State:
    def __init__(self, config: Config, grid, player: int, _bits=None):
        self.config = config
        self.grid = np.asarray(grid, dtype=np.int8).reshape(config.height, config.width)
        self.player = int(player)
        # one bit set per player: cell (row r from the bottom, column c) -> bit c * (height + 1) + r
        self._bits = _bits if _bits is not None else _to_bits(self.grid)
        winner = _winner(self._bits, config.height, config.count)
        full = not (self.grid[config.height - 1] == -1).any()
        self.has_ended = winner is not None or full
        self.reward = np.zeros(config.num_players, dtype=np.float32)
        if winner is not None:
            self.reward[:] = -1.0
            self.reward[winner] = 1.0
        self._actions: list[Action] | None = None

    @property
    def actions(self) -> list[Action]:
        if self._actions is None:
            top = self.grid[self.config.height - 1]
            self._actions = [] if self.has_ended else [Action(self, c) for c in range(self.config.width) if top[c] == -1]
        return self._actions

    def to_json(self) -> dict:
        return {"config": self.config.to_json(), "grid": self.grid.tolist(), "player": self.player}

    @classmethod
    def from_json(cls, data: dict) -> "State":
        return cls(Config.from_json(data["config"]), data["grid"], data["player"])

    def __eq__(self, other):
        return (isinstance(other, State) and self.player == other.player and self.config == other.config
                and np.array_equal(self.grid, other.grid))

    def __hash__(self):
        return hash((self.player, self.grid.tobytes()))

    def __repr__(self):
        rows = ["".join(".XO"[v + 1] for v in row) for row in self.grid[::-1]]
        return "State(player=%d)\n%s" % (self.player, "\n".join(rows))


def _to_bits(grid: np.ndarray) -> tuple:
    h = grid.shape[0]
    out = []
    for who in (0, 1):
        r, c = np.nonzero(grid == who)
        out.append(int(sum(1 << (int(cc) * (h + 1) + int(rr)) for rr, cc in zip(r, c))))
    return tuple(out)


def _winner(bits: tuple, h: int, k: int):
    """k in a row on bit sets with one spare bit per column: vertical 1, horizontal h + 1, diagonals h and h + 2."""
    for who in (0, 1):
        m = bits[who]
        for step in (1, h + 1, h, h + 2):
            run = m
            for i in range(1, k):
                run &= m >> (i * step)
            if run:
                return who
    return None
