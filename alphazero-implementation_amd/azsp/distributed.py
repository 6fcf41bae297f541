"""Multi-GPU side of self-play: one process per GPU, games sharded over ranks with no traffic during
search; the only collective is one variable-length all-gather of packed (s, pi, z) records at the
end of a generation call (RCCL over xGMI when the tensors are on GPUs; gloo in the CPU tests).

Record layout (csrc k_export_records, azsp_record_bytes):
    int8 state[cells] | int8 value | zero pad to a multiple of 8 | float64 policy[A]
The reference has no distributed path; this is new surface."""
from __future__ import annotations

import numpy as np
import torch


def record_bytes(cells: int, n_actions: int) -> int:
    return (cells + 1 + 7) // 8 * 8 + 8 * n_actions


def pack_records(states: np.ndarray, policies: np.ndarray, values: np.ndarray) -> np.ndarray:
    """Host-side packer with the device layout (used by tests and by CPU-only tooling)."""
    n = states.shape[0]
    cells, A = int(np.prod(states.shape[1:])), policies.shape[1]
    head = (cells + 1 + 7) // 8 * 8
    rec = np.zeros((n, head + 8 * A), np.uint8)
    rec[:, :cells] = states.reshape(n, cells).astype(np.int8).view(np.uint8)
    rec[:, cells] = values.astype(np.int8).view(np.uint8)
    rec[:, head:] = np.ascontiguousarray(policies, dtype=np.float64).view(np.uint8).reshape(n, 8 * A)
    return rec


def unpack_records(rec: np.ndarray, cells: int, n_actions: int):
    """records uint8[n][record_bytes] -> (states int8[n][cells], policies float64[n][A], values int32[n])."""
    rec = np.ascontiguousarray(rec)
    head = (cells + 1 + 7) // 8 * 8
    states = rec[:, :cells].view(np.int8).copy()
    values = rec[:, cells].view(np.int8).astype(np.int32)
    policies = np.ascontiguousarray(rec[:, head:head + 8 * n_actions]).view(np.float64).reshape(-1, n_actions)
    return states, policies, values


def all_gather_variable(local: torch.Tensor, dist, world: int) -> list[torch.Tensor]:
    """All-gather of 2-D tensors whose first dimension differs per rank: counts first, then one
    padded all_gather_into_tensor (a single large collective: per-link xGMI bandwidth bound, not
    latency bound).  Returns the per-rank tensors trimmed to their true lengths."""
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n_local)
    counts_h = counts.cpu().tolist()
    width = local.shape[1]
    cap = max(max(counts_h), 1)
    padded = torch.zeros((cap, width), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * cap, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return [out[r * cap: r * cap + counts_h[r]] for r in range(world)]


def export_records(gen, device: torch.device | None = None) -> torch.Tensor:
    """Pack every episode of the generator's last run into a CUDA uint8 tensor, device to device."""
    eng = gen._engine
    st = gen.last_stats
    if eng is None or st is None:
        raise RuntimeError("export_records needs a finished generate_episodes run")
    n_eps = eng.n_slots
    device = eng.torch_device(device)          # the export kernel writes on the engine's own GPU
    cap = n_eps * eng.max_plies
    buf = torch.empty((cap, eng.record_bytes), dtype=torch.uint8, device=device)
    eng.sync_torch_stream()
    n = eng.export_records_device(0, n_eps, buf.data_ptr(), cap)
    return buf[:n]


def all_gather_records(gen, dist, world: int, coll_device: torch.device | None = None) -> list[torch.Tensor]:
    """Every rank ends up with every rank's records.  `coll_device` = where the collective runs
    (the GPU for RCCL; CPU for the gloo rehearsal mode)."""
    rec = export_records(gen)
    if coll_device is not None and coll_device.type != rec.device.type:
        rec = rec.to(coll_device)
    return all_gather_variable(rec, dist, world)
