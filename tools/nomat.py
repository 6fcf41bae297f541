"""The engine loop alone (search / play, no episode fetch, no Python objects) against the full generator: is the host the limit?
    python tools/nomat.py [--cache] [--f16]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
import torch, azsp
f16 = "--f16" in sys.argv; cache = "--cache" in sys.argv
torch.manual_seed(0)
net = azsp.ResNet((6, 7), 7, 4, 64).eval()
gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, 100, 4096, 1, 1), f16_split=f16, eval_cache=cache)
np.random.seed(1); n = sum(1 for _ in gen.generate_episodes(net))
eng = gen.engine_for(net)
def step(seed, fetch):
    np.random.seed(seed)
    eng.reset(); eng.set_search_options()
    eng.search(); emitted = 0
    while True:
        res = eng.play(np.random.random_sample(4096))
        if not res.finished: eng.search()
        if fetch and res.n_finished:
            out = eng.fetch_episodes(emitted, res.n_finished)
        emitted += res.n_finished
        if res.finished: return
for fetch in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4): step(10 * k, fetch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print({"f16": f16, "cache": cache, "fetch": fetch, "ms_per_step": round(dt * 1e3, 1), "eps_per_s": round(4096 / dt, 1)})
t0 = time.perf_counter()
for k in range(4):
    np.random.seed(10 * k); n = sum(1 for _ in gen.generate_episodes(net))
dt = (time.perf_counter() - t0) / 4
print({"full generator": True, "ms_per_step": round(dt * 1e3, 1), "eps_per_s": round(4096 / dt, 1)})
