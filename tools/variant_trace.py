"""Phase trace of ONE inference-kernel variant (mean microseconds per phase over the workgroups, in-kernel clock):
    python tools/variant_trace.py <boards> <variant name, e.g. c4_64_mid | c4_64_one | c4_64_big> [f16]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-implementation_amd"))
os.environ["AZSP_PROBE_VERBOSE"] = "1"
os.environ["AZSP_RESNET_VARIANT"] = sys.argv[2] if len(sys.argv) > 2 else "c4_64_mid"
from azsp.engine import Engine
from azsp.weights import resnet_blob
import torch
from azsp.model import ResNet
torch.manual_seed(0)
blob = resnet_blob(ResNet((6, 7), 7, 4, 64))[0]
rng = np.random.RandomState(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
pb = np.zeros((B, 6, 7), np.int8)
for b in pb:
    for c in range(7):
        h = rng.randint(0, 6)
        b[6 - h:, c] = rng.choice([-1, 1], size=h)
F16 = len(sys.argv) > 3 and sys.argv[3] == "f16"        # phase trace of the split-fp16 kernel (six-board form)
with Engine("connect4", B, 1, res_blocks=4, channels=64, f16_split=F16) as e:
    e.load_resnet(blob)
    for _ in range(1500):
        e.predict(pb)
    print(e.probe_clock(pb), flush=True)
