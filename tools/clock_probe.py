"""In-kernel clock of the fp32 inference kernel (azsp_probe_clock) after W untraced launches in the SAME process:
    python tools/clock_probe.py --boards 3072 --warm 0 100 1000 4000"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
from azsp.engine import Engine            # noqa: E402
from azsp.weights import resnet_blob      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boards", type=int, default=3072)
ap.add_argument("--warm", type=int, nargs="+", default=[0, 100, 1000, 4000])
a = ap.parse_args()
import torch                               # noqa: E402
from azsp.model import ResNet              # noqa: E402
torch.manual_seed(0)
blob = resnet_blob(ResNet((6, 7), 7, 4, 64))[0]
rng = np.random.RandomState(0)
pb = np.zeros((a.boards, 6, 7), np.int8)
for b in pb:
    for c in range(7):
        h = rng.randint(0, 6)
        b[6 - h:, c] = rng.choice([-1, 1], size=h)
with Engine("connect4", a.boards, 1, res_blocks=4, channels=64) as e:
    e.load_resnet(blob)
    for w in a.warm:
        t = time.time()
        for _ in range(w):
            e.predict(pb)
        dt = time.time() - t
        ghz, span = e.probe_clock(pb)
        print({"boards": a.boards, "untraced_launches_before": w, "seconds": round(dt, 2), "clock_ghz": round(ghz, 3), "span_us": round(span, 1)}, flush=True)
