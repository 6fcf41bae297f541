"""What sits right after the hot path in the reference:

* `generate_dataset` / `DeviceReplayBuffer`: the reference flattens episodes into Python lists and calls
  `model.format_dataset(boards, policies, values)` (src/alphazero_less_simple/core/training/datamodule.py:134-142,
  src/alphazero_simple/base_model.py:61-67) -- device -> host -> lists -> np.stack -> device.  Here the
  generator's arena is turned into the same three training tensors by one kernel, on the GPU.
* `save_episodes` / `load_episodes`: the reference's JSON wire format (datamodule.py:87-103,
  episode.py:15-33,64-75), including the loader the reference defines but never calls."""
from __future__ import annotations

import json
from collections import deque
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import TensorDataset

from .episode import Episode


def generate_dataset(gen, model, device: torch.device | None = None) -> TensorDataset:
    """Run one generation (`gen` is an azsp.EpisodeGenerator) without materialising Episode objects and
    return TensorDataset(planes float32[S,3,rows,cols], policies float32[S,A], values float32[S]) on the
    GPU -- element for element what `model.format_dataset` builds from the yielded episodes."""
    eng = gen.engine_for(model)
    n = gen.config.num_episodes
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
    rows, cols = (eng.rows, eng.cols)
    planes = torch.empty((s, 3, rows, cols), dtype=torch.float32, device=device)
    pol = torch.empty((s, eng.n_actions), dtype=torch.float32, device=device)
    val = torch.empty((s,), dtype=torch.float32, device=device)
    eng.sync_torch_stream()
    got = eng.export_dataset_device(0, n, planes.data_ptr(), pol.data_ptr(), val.data_ptr(), s)
    assert got == s
    return TensorDataset(planes, pol, val)


class DeviceReplayBuffer:
    """Tensor-backed stand-in for the reference's `deque[Episode]` replay window
    (datamodule.py:67-69: maxlen = num_iters_for_train_history * num_episodes): keeps the datasets of
    the last `history` generation calls on the device and serves their concatenation."""

    def __init__(self, history: int):
        self._parts: deque[TensorDataset] = deque(maxlen=max(int(history), 1))

    def extend(self, dataset: TensorDataset) -> None:
        self._parts.append(dataset)

    def __len__(self) -> int:
        return sum(len(p) for p in self._parts)

    def dataset(self) -> TensorDataset:
        cols = list(zip(*(p.tensors for p in self._parts)))
        return TensorDataset(*(torch.cat(c) for c in cols))


def save_episodes(path, episodes) -> None:
    """Same file the reference's DataModule._save_episodes writes (datamodule.py:87-96)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        json.dump([ep.to_dict() for ep in episodes], f)


def load_episodes(path) -> list[Episode]:
    """DataModule._load_episodes (datamodule.py:98-103)."""
    with open(path) as f:
        return [Episode.from_dict(d) for d in json.load(f)]
