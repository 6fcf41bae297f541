"""cProfile of the Python side of one generate_episodes step (what the host spends while the GPU searches)."""
import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
import torch, azsp
torch.manual_seed(0)
net = azsp.ResNet((6, 7), 7, 4, 64).eval()
gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, 100, 4096, 1, 1), f16_split=True, eval_cache=True)
np.random.seed(1); n = sum(1 for _ in gen.generate_episodes(net))
pr = cProfile.Profile()
np.random.seed(2)
t0 = time.perf_counter()
pr.enable()
eps = list(gen.generate_episodes(net))
pr.disable()
print("step", time.perf_counter() - t0, "s, episodes", len(eps))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
