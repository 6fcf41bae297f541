"""Drop-in for the reference's self-play driver
(reference: src/alphazero_less_simple/core/training/episode_generator.py:13-87).

Same constructor (`EpisodeGenerator(*, game, config)`) and the same generator protocol
(`generate_episodes(model)` yields `Episode`s in (ply, slot) completion order and stops at the
`config.num_episodes`-th completion), but all `num_episodes` games and their search trees live on
the GPU: per ply the host only draws the uniforms the reference would draw from the global numpy
RNG (one per slot, slot order) and collects the episodes that finished."""
from __future__ import annotations

from typing import Generator

import numpy as np

from .config import AlphaZeroConfig
from .engine import Engine
from .episode import Episode, Sample
from .mcts import WeightsOnDevice, default_device, engine_geometry_for_model


class EpisodeGenerator:
    def __init__(self, *, game, config: AlphaZeroConfig, device: int | None = None, timing: bool = False,
                 n_groups: int = 0, f16_split: bool = False, sampling_plies: int | None = None,
                 dirichlet_alpha: float | None = None, dirichlet_eps: float = 0.25, c_init: float = 1.0,
                 c_base: float = 0.0, numpy1_value_sums: bool = False, tree_reuse: bool = False, eval_cache: bool = False,
                 eval_cache_entries: int = 0, timing_sample: int = 0):
        self.game = game
        self.config = config
        self.device = device
        self.timing = timing
        self.timing_sample = timing_sample    # timing=True: time only 32 of every 32 * timing_sample simulation rounds (0 / 1: all)
        # search options beyond the reference (None/off = reference behaviour): arg-max moves after
        # `sampling_plies` plies; gamma(alpha) root noise mixed in with weight `dirichlet_eps`, drawn per ply from
        # the global numpy RNG BEFORE the ply's uniforms (n_slots x n_actions values)
        self.sampling_plies, self.dirichlet_alpha, self.dirichlet_eps = sampling_plies, dirichlet_alpha, dirichlet_eps
        # exploration factor c(N) in front of P*sqrt(N)/(n+1): constant c_init, or log((N+c_base+1)/c_base)+c_init
        self.c_init, self.c_base = c_init, c_base
        # True: float64 value sums, the arithmetic the reference's source has under numpy 1.x (AZSP_RULES_SIMPLE_NUMPY1)
        self.numpy1_value_sums = numpy1_value_sums
        # True: keep the chosen child's subtree (and its statistics) as the next ply's tree (AZSP_FLAG_TREE_REUSE)
        self.tree_reuse = tree_reuse
        # True: every position is evaluated at most once per generate_episodes call (AZSP_FLAG_EVAL_CACHE): a position ->
        # (p, v) table in HBM; identical episodes, fewer network evaluations (Connect-4 / Connect2, device evaluators)
        self.eval_cache, self.eval_cache_entries = eval_cache, eval_cache_entries
        self.f16_split = f16_split    # device ResNet on the fp16 matrix pipe with split operands (fp32-class accuracy)
        self.n_groups = n_groups      # concurrent slot groups (HIP streams); 0 = automatic; never changes results
        self._engine: Engine | None = None
        self._engine_key = None
        self._weights = WeightsOnDevice()     # which model's weights the engine holds (re-exported only when they changed)
        self.last_stats: dict | None = None

    def invalidate_device_weights(self) -> None:
        """Force the next generate_episodes call to export and upload the model's weights again (needed only after
        updates that by-pass the tensors' version counters: writes through `.data`, `set_()`)."""
        self._weights.invalidate()

    def engine_for(self, model) -> Engine:
        blocks, ch, salt = engine_geometry_for_model(model, self.game)
        key = (self.config.num_episodes, self.config.num_simulations, blocks, ch, salt, self.f16_split, self.numpy1_value_sums, self.tree_reuse,
               self.eval_cache, self.eval_cache_entries)
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            dev = default_device() if self.device is None else self.device
            self._engine = Engine(self.game, self.config.num_episodes, self.config.num_simulations, device=dev,
                                  res_blocks=blocks, channels=ch, timing=self.timing, hash_salt=salt,
                                  n_groups=self.n_groups, f16_split=self.f16_split,
                                  search_rules=2 if self.numpy1_value_sums else 0, tree_reuse=self.tree_reuse,
                                  eval_cache=self.eval_cache, eval_cache_entries=self.eval_cache_entries,
                                  timing_sample=self.timing_sample)
            self._engine_key = key
            self._weights.invalidate()          # a new engine holds no weights (and may reuse the old one's id)
        # The weights are exported and uploaded again only when the model changed since the last call -- every training
        # iteration does, a second call on the same weights does not.
        self._weights.ensure(self._engine, model)
        return self._engine

    def generate_episodes(self, model) -> Generator[Episode, None, None]:
        eng = self.engine_for(model)
        n = self.config.num_episodes
        shape = eng.board_shape
        eng.reset()
        eng.set_search_options(-1 if self.sampling_plies is None else self.sampling_plies,
                               self.dirichlet_eps if self.dirichlet_alpha else 0.0, self.c_init, self.c_base)

        def search():
            if self.dirichlet_alpha:
                eng.search_with_noise(np.random.gamma(self.dirichlet_alpha, 1.0, (n, eng.n_actions)))
            else:
                eng.search()

        emitted = 0
        search()
        while True:
            rng_state = np.random.get_state()
            res = eng.play(np.random.random_sample(n))
            if res.uniforms_used < n:
                # the reference returns inside its slot loop at the N-th completion: later slots
                # of that ply never draw.  Rewind so the global RNG ends where the reference leaves it.
                np.random.set_state(rng_state)
                np.random.random_sample(res.uniforms_used)
            if not res.finished:
                search()            # enqueue the next ply's search now: it runs on the GPU while the
                                    # host copies and materialises the episodes that just finished
            if res.n_finished:
                out = eng.fetch_episodes(emitted, res.n_finished)
                states = out["states"].astype(np.int64).reshape((-1,) + shape)
                off = out["ep_offset"]
                for e in range(res.n_finished):
                    yield Episode(samples=[Sample(state=states[k], policy=out["policies"][k], value=int(out["values"][k]))
                                           for k in range(off[e], off[e + 1])])
                emitted += res.n_finished
            if res.finished:
                self.last_stats = eng.stats()
                return
