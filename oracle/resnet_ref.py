"""CPU ORACLE for the network (test infrastructure only): a functional torch-CPU restatement of
the reference ResNet's inference path, driven directly by a state_dict in the reference layout.

  forward   /root/reference/src/alphazero_simple/resnet.py:86-96 (+ ResBlock.forward :21-27)
  planes    resnet.py:115-128 (_states_to_tensor: [empty, mine, theirs] float32)
  predict   resnet.py:130-157 (eval-mode BN, illegal columns -> -inf, softmax dim 1, tanh)

Pinned by tests/test_oracle_golden.py::test_resnet_ref_matches_reference_vectors against
tests/golden/resnet_forward.npz (outputs of the reference's own ResNet.predict)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


class ResNetRef:
    def __init__(self, state_dict: dict):
        self.sd = {k: _t(v).float() for k, v in state_dict.items() if "num_batches_tracked" not in k}
        self.blocks = 0
        while f"residual_blocks.{self.blocks}.conv1.weight" in self.sd:
            self.blocks += 1
        self.calls = 0
        self.boards_seen = 0

    def _conv_bn(self, x, conv, bn, pad):
        s = self.sd
        x = F.conv2d(x, s[f"{conv}.weight"], s[f"{conv}.bias"], padding=pad)
        return F.batch_norm(x, s[f"{bn}.running_mean"], s[f"{bn}.running_var"], s[f"{bn}.weight"], s[f"{bn}.bias"],
                            training=False, eps=EPS)

    @staticmethod
    def planes(boards) -> torch.Tensor:
        b = np.stack(boards)
        return torch.from_numpy(np.stack([(b == 0), (b == 1), (b == -1)], axis=1).astype(np.float32))

    def forward(self, x: torch.Tensor):
        s = self.sd
        x = F.relu(self._conv_bn(x, "input_conv.0", "input_conv.1", 1))
        for i in range(self.blocks):
            r = x
            x = F.relu(self._conv_bn(x, f"residual_blocks.{i}.conv1", f"residual_blocks.{i}.bn1", 1))
            x = self._conv_bn(x, f"residual_blocks.{i}.conv2", f"residual_blocks.{i}.bn2", 1)
            x = F.relu(x + r)
        p = F.relu(self._conv_bn(x, "policy_head.0", "policy_head.1", 0)).flatten(1)
        p = F.linear(p, s["policy_head.4.weight"], s["policy_head.4.bias"])
        v = F.relu(self._conv_bn(x, "value_head.0", "value_head.1", 1)).flatten(1)
        v = F.linear(v, s["value_head.4.weight"], s["value_head.4.bias"])
        return p, v.view(-1)

    def predict(self, boards):
        self.calls += 1
        self.boards_seen += len(boards)
        with torch.no_grad():
            logits, vl = self.forward(self.planes(boards))
            legal = torch.from_numpy(np.stack([np.asarray(b)[0] == 0 for b in boards]))
            probs = F.softmax(logits.masked_fill(~legal, float("-inf")), dim=1)
            val = torch.tanh(vl)
        return list(probs.numpy()), list(val.numpy().flatten())
