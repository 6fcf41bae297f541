"""Headline workload without the timing flag (no HIP events around the kernels): what bench.py's event records cost.
    python tools/notiming.py [--cache] [--f16] [--timing] [--slots 4096] [--sims 100] [--groups 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
import torch                                   # noqa: E402
import azsp                                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cache", action="store_true")
ap.add_argument("--f16", action="store_true")
ap.add_argument("--timing", action="store_true")
ap.add_argument("--timing-sample", type=int, default=0)
ap.add_argument("--slots", type=int, default=4096)
ap.add_argument("--sims", type=int, default=100)
ap.add_argument("--groups", type=int, default=0)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--channels", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--gc", choices=["default", "off", "freeze"], default="default")
a = ap.parse_args()
torch.manual_seed(0)
net = azsp.ResNet((6, 7), 7, a.blocks, a.channels).eval()
gen = azsp.EpisodeGenerator(game=azsp.Connect4Game(), config=azsp.AlphaZeroConfig(32, 1, a.sims, a.slots, 1, 1), timing=a.timing,
                            n_groups=a.groups, f16_split=a.f16, eval_cache=a.cache, timing_sample=a.timing_sample)
import gc                                      # noqa: E402
if a.gc == "off":
    gc.disable()
elif a.gc == "freeze":
    gc.collect(); gc.freeze()
np.random.seed(1000)
n = sum(1 for _ in gen.generate_episodes(net))
torch.cuda.synchronize()
gen.engine_for(net).reset_stats()
t0 = time.perf_counter()
n = 0
for k in range(a.steps):
    np.random.seed(10 * k)
    n += sum(1 for _ in gen.generate_episodes(net))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = gen.last_stats
print({"gc": a.gc, "cache": a.cache, "f16": a.f16, "timing": a.timing, "slots": a.slots, "groups": a.groups, "episodes_per_s": round(n / dt, 1), "ms_per_step": round(1e3 * dt / a.steps, 1),
       "rounds": int(st["rounds"]), "us_per_round": round(1e6 * dt / max(int(st["rounds"]), 1), 1)})
