"""One reference generator run, sharded over several engines / GPUs, with the reference's EXACT semantics
("mode A"): the N slots of `EpisodeGenerator.generate_episodes`
(src/alphazero_less_simple/core/training/episode_generator.py:23-87) are cut into contiguous ranges, one
engine per range; searches run independently (no traffic); per ply the shards only exchange how many of
their games ended, so that the global "first N completions in (ply, slot) order" stop rule and the global
RNG position come out exactly as in the sequential reference.

  ShardedEpisodeGenerator      all shards in this process (one engine per listed device; several shards may
                               share a GPU) -- yields episodes in the reference's global order
  DistributedEpisodeGenerator  one shard per torch.distributed rank; every rank draws the same uniforms from
                               an identically seeded numpy RNG and yields its own episodes, tagged with
                               (ply, global slot) so that any rank can merge them into the reference order
"""
from __future__ import annotations

from typing import Generator

import numpy as np

from .config import AlphaZeroConfig
from .engine import Engine
from .episode import Episode, Sample
from .mcts import configure_engine_for_model, engine_geometry_for_model


def shard_ranges(n_slots: int, n_shards: int) -> list[tuple[int, int]]:
    return [(n_slots * r // n_shards, n_slots * (r + 1) // n_shards) for r in range(n_shards)]


def shard_allowances(done_counts: list[int], remaining: int) -> tuple[list[int], int | None]:
    """Given how many games ended in each shard this ply (shards in slot order) and how many episodes the
    generator may still emit, return (allow[r], stop_rank): allow[r] completions of shard r are within the
    first `remaining`; stop_rank is the shard holding the `remaining`-th one (None if the ply does not stop)."""
    allow, left, stop = [], remaining, None
    for r, d in enumerate(done_counts):
        a = min(d, max(left, 0))
        allow.append(a)
        if stop is None and d >= left > 0:
            stop = r
        left -= d
    if remaining <= 0:
        stop = None
    return allow, stop


def _episodes_from(eng: Engine, first: int, n: int, shape) -> list[tuple[Episode, int, int]]:
    out = eng.fetch_episodes(first, n)
    states = out["states"].astype(np.int64).reshape((-1,) + shape)
    off = out["ep_offset"]
    res = []
    for e in range(n):
        ep = Episode(samples=[Sample(state=states[k], policy=out["policies"][k], value=int(out["values"][k]))
                              for k in range(off[e], off[e + 1])])
        res.append((ep, int(out["ep_ply"][e]), int(out["ep_slot"][e])))
    return res


class ShardedEpisodeGenerator:
    def __init__(self, *, game, config: AlphaZeroConfig, devices: list[int], f16_split: bool = False, n_groups: int = 0):
        self.game, self.config, self.devices = game, config, list(devices)
        self.f16_split, self.n_groups = f16_split, n_groups
        self.ranges = shard_ranges(config.num_episodes, len(self.devices))
        self._engines: list[Engine] | None = None
        self._key = None

    def _engines_for(self, model) -> list[Engine]:
        blocks, ch, salt = engine_geometry_for_model(model, self.game)
        key = (blocks, ch, salt)
        if self._engines is None or key != self._key:
            for e in self._engines or []:
                e.close()
            n = self.config.num_episodes
            self._engines = [Engine(self.game, hi - lo, self.config.num_simulations, device=dev, res_blocks=blocks,
                                    channels=ch, hash_salt=salt, f16_split=self.f16_split, n_groups=self.n_groups,
                                    arena_episodes=n)
                             for dev, (lo, hi) in zip(self.devices, self.ranges)]
            self._key = key
        for e in self._engines:
            configure_engine_for_model(e, model)
        return self._engines

    def generate_episodes(self, model) -> Generator[Episode, None, None]:
        engines = self._engines_for(model)
        n = self.config.num_episodes
        shape = engines[0].board_shape
        for e in engines:
            e.reset()
        emitted = [0] * len(engines)
        total = 0
        for e in engines:
            e.search()                               # asynchronous: all shards search concurrently
        while True:
            rng_state = np.random.get_state()
            u = np.random.random_sample(n)           # the reference's draws: one per slot, slot order
            done = [e.play_begin(u[lo:hi]) for e, (lo, hi) in zip(engines, self.ranges)]
            allow, stop = shard_allowances(done, n - total)
            results = [e.play_commit(a) for e, a in zip(engines, allow)]
            if stop is not None:                     # the N-th completion: later slots never drew
                used = self.ranges[stop][0] + results[stop].last_counted_slot + 1
                np.random.set_state(rng_state)
                np.random.random_sample(used)
            else:
                for e in engines:
                    e.search()
            for r, (e, a) in enumerate(zip(engines, allow)):
                if a:
                    for ep, _ply, _slot in _episodes_from(e, emitted[r], a, shape):
                        yield ep
                    emitted[r] += a
                    total += a
            if stop is not None:
                return

    def close(self) -> None:
        for e in self._engines or []:
            e.close()
        self._engines = None


class DistributedEpisodeGenerator:
    """One shard per rank.  `dist` is torch.distributed (initialised); every rank must hold the same numpy
    global RNG state when generate_episodes is called (e.g. np.random.seed(s) on all ranks)."""

    def __init__(self, *, game, config: AlphaZeroConfig, dist, device: int, coll_device=None, f16_split: bool = False):
        import torch
        self.torch = torch
        self.game, self.config, self.dist = game, config, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ranges = shard_ranges(config.num_episodes, self.world)
        self.device, self.f16_split = device, f16_split
        self.coll_device = coll_device or torch.device("cuda", device)
        self._engine: Engine | None = None
        self._key = None

    def _all_gather_ints(self, *vals: int) -> list[list[int]]:
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.coll_device)
        out = self.torch.empty(self.world * len(vals), dtype=self.torch.int64, device=self.coll_device)
        self.dist.all_gather_into_tensor(out, t)
        return out.view(self.world, len(vals)).cpu().tolist()

    def generate_episodes(self, model) -> Generator[tuple[Episode, int, int], None, None]:
        """Yields (episode, ply, global_slot) for the episodes of this rank's shard."""
        blocks, ch, salt = engine_geometry_for_model(model, self.game)
        lo, hi = self.ranges[self.rank]
        n = self.config.num_episodes
        key = (blocks, ch, salt)
        if self._engine is None or key != self._key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self.game, hi - lo, self.config.num_simulations, device=self.device, res_blocks=blocks,
                                  channels=ch, hash_salt=salt, f16_split=self.f16_split, arena_episodes=n)
            self._key = key
        eng = self._engine
        configure_engine_for_model(eng, model)
        eng.reset()
        emitted = total = 0
        eng.search()
        while True:
            rng_state = np.random.get_state()
            u = np.random.random_sample(n)
            done = [d[0] for d in self._all_gather_ints(eng.play_begin(u[lo:hi]))]
            allow, stop = shard_allowances(done, n - total)
            res = eng.play_commit(allow[self.rank])
            if stop is not None:
                last = [x[0] for x in self._all_gather_ints(res.last_counted_slot)]
                np.random.set_state(rng_state)
                np.random.random_sample(self.ranges[stop][0] + last[stop] + 1)
            else:
                eng.search()
            a = allow[self.rank]
            if a:
                for ep, ply, slot in _episodes_from(eng, emitted, a, eng.board_shape):
                    yield ep, ply, lo + slot
                emitted += a
            total += sum(allow)
            if stop is not None:
                return
