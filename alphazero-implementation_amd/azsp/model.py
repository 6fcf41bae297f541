"""NeuralNet plugin surface (reference: src/alphazero_simple/base_model.py:14-81 and
src/alphazero_simple/resnet.py:13-157).

`BaseModel` keeps the reference's method set (forward / predict / _states_to_tensor abstract;
format_dataset, get_inference_clone, training_step, configure_optimizers concrete).  `ResNet` has
the reference's constructor and state_dict layout, so checkpoints interchange; its `predict` runs
the fused HIP inference kernel through the C ABI (training stays in PyTorch)."""
from __future__ import annotations

import copy
import os
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn
from torch.utils.data import TensorDataset

try:                                    # the reference derives from LightningModule; optional here
    import lightning as _L
    _ModuleBase = _L.LightningModule
except Exception:                       # noqa: BLE001
    class _ModuleBase(nn.Module):
        """nn.Module with the three LightningModule members the plugin surface touches."""

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")


class BaseModel(ABC, _ModuleBase):
    def __init__(self, learning_rate: float = 1e-3):
        super().__init__()
        self.save_hyperparameters()
        self.learning_rate = learning_rate
        self.model_name = type(self).__name__
        self.save_hyperparameters({"model_name": self.model_name})

    @abstractmethod
    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]: ...

    @abstractmethod
    def predict(self, boards: list[np.ndarray]) -> tuple[list[np.ndarray], list[float]]: ...

    @abstractmethod
    def _states_to_tensor(self, boards: list[np.ndarray]) -> Tensor: ...

    def training_step(self, batch, batch_idx: int) -> Tensor:      # cross-entropy + MSE
        x, pi, z = batch
        logits, v = self.forward(x)
        policy_loss = F.cross_entropy(logits, pi)
        value_loss = F.mse_loss(v, z)
        loss = policy_loss + value_loss
        self.log("train_loss", loss)
        self.log("policy_loss", policy_loss)
        self.log("value_loss", value_loss)
        return loss

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.learning_rate, weight_decay=1e-4)

    def format_dataset(self, states, policies, values) -> TensorDataset:
        return TensorDataset(self._states_to_tensor(states), torch.FloatTensor(np.array(policies)),
                             torch.FloatTensor(np.array(values).astype(np.float64)))

    def get_inference_clone(self):
        clone = copy.deepcopy(self)
        clone.eval()
        return clone


class ResBlock(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channels, num_channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_channels)
        self.conv2 = nn.Conv2d(num_channels, num_channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(num_channels)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class ResNet(BaseModel):
    POLICY_CHANNELS, VALUE_CHANNELS = 32, 3

    def __init__(self, board_size: tuple[int, int], action_size: int, num_res_blocks: int, num_channels: int):
        super().__init__()
        self.rows, self.cols = board_size
        self.action_size = action_size
        self.num_res_blocks, self.num_channels = num_res_blocks, num_channels
        cells = self.rows * self.cols
        self.input_conv = nn.Sequential(nn.Conv2d(3, num_channels, kernel_size=3, padding=1),
                                        nn.BatchNorm2d(num_channels), nn.ReLU())
        self.residual_blocks = nn.ModuleList(ResBlock(num_channels) for _ in range(num_res_blocks))
        self.policy_head = nn.Sequential(nn.Conv2d(num_channels, self.POLICY_CHANNELS, kernel_size=1),
                                         nn.BatchNorm2d(self.POLICY_CHANNELS), nn.ReLU(), nn.Flatten(),
                                         nn.Linear(self.POLICY_CHANNELS * cells, action_size))
        self.value_head = nn.Sequential(nn.Conv2d(num_channels, self.VALUE_CHANNELS, kernel_size=3, padding=1),
                                        nn.BatchNorm2d(self.VALUE_CHANNELS), nn.ReLU(), nn.Flatten(),
                                        nn.Linear(self.VALUE_CHANNELS * cells, 1))
        self._azsp_engine = None
        self._azsp_fingerprint = None

    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]:
        x = self.input_conv(x)
        for blk in self.residual_blocks:
            x = blk(x)
        return self.policy_head(x), self.value_head(x).view(-1)

    def _states_to_tensor(self, boards) -> Tensor:
        b = np.stack(boards)
        planes = np.stack([(b == 0), (b == 1), (b == -1)], axis=1).astype(np.float32)
        return torch.from_numpy(planes)

    # ---- inference on the HIP engine ------------------------------------------------------------
    def _fingerprint(self):
        """What decides whether the device copy of the weights is current: storage address and in-place version of
        every tensor.  `_version` does not see writes through `.data` (p.data.copy_(), m.weight.data.normal_()) or
        `set_()`: after such an update call `invalidate_device_weights()` (load_state_dict does it by itself)."""
        return tuple((int(t.data_ptr()), int(t._version)) for t in list(self.parameters()) + list(self.buffers()))

    def invalidate_device_weights(self) -> None:
        """Force the next predict() to re-upload the weights to the device engine."""
        self._azsp_fingerprint = None

    def load_state_dict(self, *args, **kwargs):
        self._azsp_fingerprint = None
        return super().load_state_dict(*args, **kwargs)

    def predict(self, boards: list[np.ndarray]) -> tuple[list[np.ndarray], list[float]]:
        """boards -> (list of float32[A] move probabilities, list of float32 values); masked softmax
        over the game's legal moves (Connect-4: columns whose top cell is empty, as the reference; Gomoku: empty
        cells) and tanh, computed by the fused gfx950 kernel."""
        from .engine import Engine           # late import: training-only users never load the .so
        from .weights import resnet_blob
        if self._azsp_engine is None:
            dev = int(os.environ.get("AZSP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            trait = {(6, 7, 7): "connect4", (9, 9, 81): "gomoku9"}.get((self.rows, self.cols, self.action_size))
            if trait is None:
                raise NotImplementedError(f"no device network for a {self.rows}x{self.cols} board with {self.action_size} actions")
            self._azsp_engine = Engine(trait, 4096, 1, device=dev, res_blocks=self.num_res_blocks,
                                       channels=self.num_channels,
                                       f16_split=os.environ.get("AZSP_F16_SPLIT", "0") == "1")
        fp = self._fingerprint()
        if fp != self._azsp_fingerprint:
            self._azsp_engine.load_resnet(resnet_blob(self)[0])
            self._azsp_fingerprint = fp
        p, v = self._azsp_engine.predict(np.stack(boards))
        return list(p), list(v)

    def __deepcopy__(self, memo):            # engines are per-instance and not copyable
        eng, fp = self._azsp_engine, self._azsp_fingerprint
        self._azsp_engine = self._azsp_fingerprint = None
        try:
            cls = type(self)
            clone = cls.__new__(cls)
            memo[id(self)] = clone
            for k, val in self.__dict__.items():
                clone.__dict__[k] = copy.deepcopy(val, memo)
            return clone
        finally:
            self._azsp_engine, self._azsp_fingerprint = eng, fp
