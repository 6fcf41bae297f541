"""Weight export: ResNet.state_dict() (layout of reference src/alphazero_simple/resnet.py:44-72)
-> the flat float32 blob azsp_load_resnet expects (order documented in include/azsp.h)."""
from __future__ import annotations

import numpy as np

_BN = ("weight", "bias", "running_mean", "running_var")


def _conv_bn(prefix_conv: str, prefix_bn: str) -> list[str]:
    return [f"{prefix_conv}.weight", f"{prefix_conv}.bias"] + [f"{prefix_bn}.{k}" for k in _BN]


def resnet_key_order(num_res_blocks: int) -> list[str]:
    keys = _conv_bn("input_conv.0", "input_conv.1")
    for i in range(num_res_blocks):
        keys += _conv_bn(f"residual_blocks.{i}.conv1", f"residual_blocks.{i}.bn1")
        keys += _conv_bn(f"residual_blocks.{i}.conv2", f"residual_blocks.{i}.bn2")
    keys += _conv_bn("policy_head.0", "policy_head.1") + ["policy_head.4.weight", "policy_head.4.bias"]
    keys += _conv_bn("value_head.0", "value_head.1") + ["value_head.4.weight", "value_head.4.bias"]
    return keys


def is_resnet_state_dict(sd) -> bool:
    return all(k in sd for k in ("input_conv.0.weight", "policy_head.4.weight", "value_head.4.weight"))


def resnet_geometry(sd) -> tuple[int, int]:
    """(num_res_blocks, num_channels) read off the state_dict's keys and shapes (no tensor is copied)."""
    blocks = 0
    while f"residual_blocks.{blocks}.conv1.weight" in sd:
        blocks += 1
    return blocks, int(sd["input_conv.0.weight"].shape[0])


def weights_fingerprint(model):
    """What decides whether the engine's copy of a torch module's weights is current: storage address and in-place version
    counter of every parameter and buffer (optimizer steps, BatchNorm's running statistics and load_state_dict all bump
    the counters; a deep copy has new storage).  Writes through `.data` or `set_()` are invisible to it: call the
    generator's `invalidate_device_weights()` after such an update.  None for models that are not torch modules (their
    weights are exported on every call)."""
    params, buffers = getattr(model, "parameters", None), getattr(model, "buffers", None)
    if not callable(params) or not callable(buffers):
        return None
    try:
        return tuple((int(t.data_ptr()), int(t._version)) for t in list(params()) + list(buffers()))
    except (AttributeError, RuntimeError, TypeError):
        return None


def resnet_blob(model_or_state_dict) -> tuple[np.ndarray, int, int]:
    """Returns (blob float32[n], num_res_blocks, num_channels)."""
    sd = model_or_state_dict if isinstance(model_or_state_dict, dict) else model_or_state_dict.state_dict()
    if not is_resnet_state_dict(sd):
        raise ValueError("state_dict does not have the ResNet layout (input_conv / residual_blocks / policy_head / value_head)")
    blocks = 0
    while f"residual_blocks.{blocks}.conv1.weight" in sd:
        blocks += 1
    channels = int(np.asarray(_np(sd["input_conv.0.weight"])).shape[0])
    parts = [np.ascontiguousarray(_np(sd[k]), dtype=np.float32).reshape(-1) for k in resnet_key_order(blocks)]
    return np.concatenate(parts), blocks, channels


def _np(t):
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().numpy()
