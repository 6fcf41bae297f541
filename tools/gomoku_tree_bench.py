#!/usr/bin/env python3
"""Tree-kernel throughput on 9x9 Gomoku (81 actions, BASELINE.json configs[4]'s search shape) with the hash
evaluator: measures the HBM-bound selection / expansion / backup kernels without any network."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alphazero-implementation_amd")]
import numpy as np  # noqa: E402
import azsp  # noqa: E402


class HashNet:
    azsp_hash_salt = 3


for n, sims in ((1024, 200), (4096, 200)):
    gen = azsp.EpisodeGenerator(game=azsp.GomokuGame(), config=azsp.AlphaZeroConfig(32, 1, sims, n, 1, 1), timing=True)
    np.random.seed(0)
    t = time.time()
    eps = list(gen.generate_episodes(HashNet()))
    dt = time.time() - t
    st = gen.last_stats
    n_sims = st["evals"] + st["terminal_leaves"]
    print(f"gomoku9 hash evaluator, {n} games x {sims} sims: {len(eps) / dt:.1f} episodes/s in {dt:.1f} s, "
          f"{st['plies']} plies, select {st['select_ms']:.0f} ms, expand {st['expand_ms']:.0f} ms, {n_sims / dt / 1e6:.2f} M sims/s, "
          f"mean path {st['path_nodes'] / max(n_sims, 1):.2f} nodes")
