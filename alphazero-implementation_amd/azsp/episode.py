"""Output records of self-play (reference: src/alphazero_less_simple/core/training/episode.py:7-75).
Same field names, same JSON schema: {"samples": [{"state": [[..]], "policy": [..], "value": v}]}."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np


@dataclass
class Sample:
    state: np.ndarray        # canonical board, np.int64, side to move = +1
    policy: np.ndarray       # MCTS visit distribution, np.float64[A]
    value: float             # game outcome from this position's side to move (python int: -1/0/+1)

    def to_dict(self) -> dict[str, Any]:
        return {"state": self.state.tolist(), "policy": self.policy.tolist(), "value": self.value}

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "Sample":
        return cls(state=np.array(data["state"]), policy=np.array(data["policy"]), value=data["value"])


@dataclass
class Episode:
    samples: list[Sample] = field(default_factory=list)

    def __len__(self) -> int:
        return len(self.samples)

    def add_sample(self, sample: Sample) -> None:
        self.samples.append(sample)

    def backpropagate_outcome(self, value: float) -> None:
        for s in self.samples:
            s.value = value

    @property
    def current_state(self) -> np.ndarray:
        return self.samples[-1].state

    def to_dict(self) -> dict[str, Any]:
        return {"samples": [s.to_dict() for s in self.samples]}

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "Episode":
        return cls(samples=[Sample.from_dict(d) for d in data["samples"]])
