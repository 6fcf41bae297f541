"""Throughput of the first-generation surface (azsp.gen1: search + episode generation on the connect package's states)."""
import sys, time
sys.path.insert(0, "alphazero-implementation_amd")
import numpy as np, torch
from azsp import gen1, connect_state as cs
from azsp.model import ResNet
torch.manual_seed(0)
net = ResNet((6, 7), 7, 4, 64).eval()
model = gen1.ResNetModel(net)
for n in (1024, 4096):
    gen = gen1.EpisodeGenerator(model=model, num_simulations=100, num_episodes=n, game_initial_state=cs.Config().sample_initial_state())
    np.random.seed(0)
    list(gen.generate_episodes())
    t = time.time(); eps = list(gen.generate_episodes()); dt = time.time() - t
    print("gen1 episodes objects:", n, round(n / dt, 1), "episodes/s", flush=True)
    t = time.time(); ds = gen1.generate_dataset(gen); dt = time.time() - t
    print("gen1 device dataset  :", n, round(n / dt, 1), "episodes/s", len(ds), flush=True)
