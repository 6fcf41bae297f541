"""Phase trace (AZSP_RESNET_TRACE stamps) and in-kernel clock of both inference kernels at one batch size:
    python tools/phase_trace.py [boards]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
os.environ["AZSP_PROBE_VERBOSE"] = "1"
from azsp.engine import Engine
from azsp.weights import resnet_blob
import torch
from azsp.model import ResNet
torch.manual_seed(0)
blob = resnet_blob(ResNet((6, 7), 7, 4, 64))[0]
rng = np.random.RandomState(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
pb = np.zeros((B, 6, 7), np.int8)
for b in pb:
    for c in range(7):
        h = rng.randint(0, 6)
        b[6 - h:, c] = rng.choice([-1, 1], size=h)
for f16 in (True, False):
    with Engine("connect4", B, 1, res_blocks=4, channels=64, f16_split=f16) as e:
        e.load_resnet(blob)
        for _ in range(1500):
            e.predict(pb)
        print("f16" if f16 else "f32", e.probe_clock(pb), flush=True)
