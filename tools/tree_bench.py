"""Tree kernels alone (hash evaluator, no inference kernel on the machine): time per fused expand+select launch.
    python tools/tree_bench.py [--game connect4|gomoku9] [--slots 4096] [--sims 100] [--groups 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
import azsp                                   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--game", default="connect4")
ap.add_argument("--slots", type=int, default=4096)
ap.add_argument("--sims", type=int, default=100)
ap.add_argument("--groups", type=int, default=0)
a = ap.parse_args()


class Hash:
    azsp_hash_salt = 3


game = azsp.Connect4Game() if a.game == "connect4" else azsp.GomokuGame()
gen = azsp.EpisodeGenerator(game=game, config=azsp.AlphaZeroConfig(32, 1, a.sims, a.slots, 1, 1), timing=True, n_groups=a.groups)
for rep in range(2):
    np.random.seed(rep)
    t0 = time.perf_counter()
    n = sum(1 for _ in gen.generate_episodes(Hash()))
    dt = time.perf_counter() - t0
st = gen.last_stats
sims = st["evals"] + st["terminal_leaves"]
print({"game": a.game, "slots": a.slots, "sims": a.sims, "episodes_per_s": round(n / dt, 1), "sims_per_s": round(sims / dt),
       "us_per_expand_select_launch": round(1e3 * st["select_ms"] / max(st["select_launches"], 1), 2),
       "select_launches": int(st["select_launches"]), "mean_levels": round(st["path_nodes"] / sims - 1, 2)})
