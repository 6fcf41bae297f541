"""Output records of self-play.  Field names, constructor keywords and the JSON schema
({"samples": [{"state": [[..]], "policy": [..], "value": v}]}) are those of the reference's records
(src/alphazero_less_simple/core/training/episode.py:7-75), so that files written by either side load in the other;
the classes themselves are plain slot objects filled from the engine's arena."""
from __future__ import annotations

from typing import Any, Iterable

import numpy as np


class Sample:
    """One training position: canonical board (np.int64, side to move = +1), MCTS visit distribution
    (np.float64[A]) and the game outcome for the side to move (python int -1 / 0 / +1)."""

    __slots__ = ("state", "policy", "value")

    def __init__(self, state: np.ndarray, policy: np.ndarray, value: float):
        self.state, self.policy, self.value = state, policy, value

    def __repr__(self) -> str:
        return f"Sample(value={self.value}, policy={np.asarray(self.policy).round(3).tolist()})"

    def __eq__(self, other) -> bool:
        return (isinstance(other, Sample) and self.value == other.value and np.array_equal(self.state, other.state)
                and np.array_equal(self.policy, other.policy))

    __hash__ = None

    # JSON wire format (episode.py:15-33)
    def to_dict(self) -> dict[str, Any]:
        return dict(state=np.asarray(self.state).tolist(), policy=np.asarray(self.policy).tolist(), value=self.value)

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "Sample":
        return cls(np.array(data["state"]), np.array(data["policy"]), data["value"])


class Episode:
    """The positions of one finished game, in playing order."""

    __slots__ = ("samples",)

    def __init__(self, samples: Iterable[Sample] | None = None):
        self.samples: list[Sample] = list(samples) if samples is not None else []

    def __len__(self) -> int:
        return len(self.samples)

    def __iter__(self):
        return iter(self.samples)

    def __repr__(self) -> str:
        return f"Episode({len(self.samples)} samples)"

    def __eq__(self, other) -> bool:
        return isinstance(other, Episode) and self.samples == other.samples

    __hash__ = None

    def add_sample(self, sample: Sample) -> None:
        self.samples.append(sample)

    def backpropagate_outcome(self, value: float) -> None:
        """Label every stored position with the same outcome (episode.py:52-54)."""
        for smp in self.samples:
            smp.value = value

    @property
    def current_state(self) -> np.ndarray:
        return self.samples[-1].state

    # JSON wire format (episode.py:64-75)
    def to_dict(self) -> dict[str, Any]:
        return {"samples": [smp.to_dict() for smp in self.samples]}

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "Episode":
        return cls(Sample.from_dict(entry) for entry in data["samples"])
