"""Python handle on one libazsp engine (one HIP device, one private stream).  Thin: every method is
one C-ABI call plus numpy buffer plumbing."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Sequence

import numpy as np

from . import _lib as L

GAME_IDS = {"connect4": L.GAME_CONNECT4, "connect2": L.GAME_CONNECT2, "gomoku9": L.GAME_GOMOKU9}


def game_id_of(game) -> int:
    """Map a BaseGame instance (or trait name) to its device trait id."""
    name = game if isinstance(game, str) else getattr(game, "device_trait", None)
    if name is None:      # duck-typed reference game objects: go by class name
        name = {"Connect4Game": "connect4", "Connect2Game": "connect2", "GomokuGame": "gomoku9"}.get(type(game).__name__)
    if name not in GAME_IDS:
        raise NotImplementedError(
            f"{type(game).__name__} has no device trait in libazsp (csrc/azsp_games.h); the engine has no CPU fallback")
    return GAME_IDS[name]


def _ptr(a: np.ndarray | None, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class Engine:
    def __init__(self, game, n_slots: int, n_sims: int, *, device: int = 0, res_blocks: int = 0, channels: int = 0,
                 timing: bool = False, hash_salt: int = 0, n_groups: int = 0, f16_split: bool = False, arena_episodes: int = 0,
                 search_rules: int = L.RULES_SIMPLE, tree_reuse: bool = False, eval_cache: bool = False,
                 eval_cache_entries: int = 0, timing_sample: int = 0):
        self.lib = L.load()
        self.game_id = game_id_of(game)
        r, c, a, mp = (C.c_int32() for _ in range(4))
        L.check(self.lib.azsp_game_info(self.game_id, C.byref(r), C.byref(c), C.byref(a), C.byref(mp)))
        self.rows, self.cols, self.n_actions, self.max_plies = r.value, c.value, a.value, mp.value
        self.cells = self.rows * self.cols
        self.board_shape = (self.rows, self.cols) if self.rows > 1 else (self.cols,)
        self.n_slots, self.n_sims = int(n_slots), int(n_sims)
        self.device = int(device)
        self.res_blocks, self.channels = int(res_blocks), int(channels)
        cfg = L.Config(C.sizeof(L.Config), self.game_id, self.n_slots, self.n_sims, int(device), self.res_blocks,
                       self.channels,
                       (L.FLAG_TIMING if timing else 0) | (L.FLAG_F16_SPLIT if f16_split else 0)
                       | (L.FLAG_TREE_REUSE if tree_reuse else 0) | (L.FLAG_EVAL_CACHE if eval_cache else 0),
                       int(hash_salt) & (2**64 - 1), int(n_groups), int(arena_episodes), int(search_rules), int(eval_cache_entries),
                       int(timing_sample), 0)
        self.search_rules = int(search_rules)
        h = C.c_void_p()
        L.check(self.lib.azsp_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._cb = None           # keep the ctypes trampoline alive
        self._cb_error: BaseException | None = None

    # ---- lifecycle ----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.azsp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- evaluator ----------------------------------------------------------------------------
    def load_resnet(self, blob: np.ndarray) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        L.check(self.lib.azsp_load_resnet(self._h, _ptr(blob, C.c_float), blob.size))
        L.check(self.lib.azsp_set_evaluator(self._h, L.EVAL_RESNET, L.PREDICT_FN(), None))

    def use_hash_evaluator(self) -> None:
        L.check(self.lib.azsp_set_evaluator(self._h, L.EVAL_HASH, L.PREDICT_FN(), None))

    def use_host_evaluator(self, predict: Callable[[list[np.ndarray]], tuple[Sequence, Sequence]]) -> None:
        """`predict` has the BaseModel.predict signature: list of int boards -> (list p, list v)."""
        cells, A, shape = self.cells, self.n_actions, self.board_shape

        def tramp(_user, boards, n, p_out, v_out):
            try:
                b = np.ctypeslib.as_array(boards, shape=(n, cells)).astype(np.int64)
                ps, vs = predict([x.reshape(shape) for x in b])
                np.ctypeslib.as_array(p_out, shape=(n, A))[:] = np.asarray(ps, dtype=np.float32).reshape(n, A)
                np.ctypeslib.as_array(v_out, shape=(n,))[:] = np.asarray(vs, dtype=np.float32).reshape(n)
                return 0
            except BaseException as exc:    # noqa: BLE001  never unwind through C
                self._cb_error = exc
                return 1

        self._cb = L.PREDICT_FN(tramp)
        L.check(self.lib.azsp_set_evaluator(self._h, L.EVAL_HOST, self._cb, None))

    def _check(self, rc: int) -> None:
        if rc != 0 and self._cb_error is not None:
            exc, self._cb_error = self._cb_error, None
            raise exc
        L.check(rc)

    def _boards_i8(self, boards) -> np.ndarray:
        b = np.ascontiguousarray(np.asarray(boards), dtype=np.int8)
        return b.reshape(-1, self.cells)

    def predict(self, boards) -> tuple[np.ndarray, np.ndarray]:
        b = self._boards_i8(boards)
        n = b.shape[0]
        p = np.empty((n, self.n_actions), np.float32)
        v = np.empty(n, np.float32)
        self._check(self.lib.azsp_predict(self._h, _ptr(b, C.c_int8), n, _ptr(p, C.c_float), _ptr(v, C.c_float)))
        return p, v

    def probe_clock(self, boards) -> tuple[float, float]:
        """(GHz the chip held inside one traced inference launch over `boards` -- fp32 or f16x2 kernel, 64 channels --, span of that launch in us)."""
        b = self._boards_i8(boards)
        ghz, span = C.c_double(), C.c_double()
        self._check(self.lib.azsp_probe_clock(self._h, _ptr(b, C.c_int8), b.shape[0], C.byref(ghz), C.byref(span)))
        return ghz.value, span.value

    # ---- search -------------------------------------------------------------------------------
    def run_batch(self, boards) -> dict[str, np.ndarray]:
        b = self._boards_i8(boards)
        n, A = b.shape[0], self.n_actions
        out = {"counts": np.empty((n, A), np.int32), "child_w": np.empty((n, A), np.float32),
               "prior": np.empty((n, A), np.float64), "root_w": np.empty(n, np.float32),
               "root_n": np.empty(n, np.int32)}
        self._check(self.lib.azsp_run_batch(self._h, _ptr(b, C.c_int8), n, _ptr(out["counts"], C.c_int32),
                                            _ptr(out["child_w"], C.c_float), _ptr(out["prior"], C.c_double),
                                            _ptr(out["root_w"], C.c_float), _ptr(out["root_n"], C.c_int32)))
        return out

    # ---- episode generation ---------------------------------------------------------------------
    def reset(self) -> None:
        self._check(self.lib.azsp_reset(self._h))

    def search(self) -> None:
        self._check(self.lib.azsp_search(self._h))

    def set_search_options(self, sampling_plies: int = -1, noise_eps: float = 0.0, c_init: float = 1.0,
                           c_base: float = 0.0) -> None:
        o = L.SearchOptions(C.sizeof(L.SearchOptions), int(sampling_plies), float(noise_eps), float(c_init), float(c_base))
        self._check(self.lib.azsp_set_search_options(self._h, C.byref(o)))

    def root_values(self, n: int | None = None) -> dict[str, np.ndarray]:
        """value_sum (float64) and visit_count of the first n trees of the search just run."""
        n = self.n_slots if n is None else int(n)
        out = {"child_w": np.empty((n, self.n_actions), np.float64), "root_w": np.empty(n, np.float64),
               "root_n": np.empty(n, np.int32)}
        self._check(self.lib.azsp_root_values(self._h, n, _ptr(out["child_w"], C.c_double), _ptr(out["root_w"], C.c_double),
                                              _ptr(out["root_n"], C.c_int32)))
        return out

    def set_initial_board(self, board=None) -> None:
        """Canonical position every game starts from (None = empty board); applies from the next reset()."""
        b = None if board is None else self._boards_i8([board])
        self._check(self.lib.azsp_set_initial_board(self._h, _ptr(b, C.c_int8)))

    def fetch_moves(self, first: int, n: int, n_samples: int) -> np.ndarray:
        out = np.empty(max(int(n_samples), 1), np.int32)
        self._check(self.lib.azsp_fetch_moves(self._h, int(first), int(n), _ptr(out, C.c_int32), out.size))
        return out[:n_samples]

    def search_with_noise(self, noise: np.ndarray) -> None:
        z = np.ascontiguousarray(noise, dtype=np.float64)
        if z.shape != (self.n_slots, self.n_actions):
            raise ValueError(f"noise must be [{self.n_slots}, {self.n_actions}]")
        self._check(self.lib.azsp_search_with_noise(self._h, _ptr(z, C.c_double)))

    def root_counts(self) -> np.ndarray:
        out = np.empty((self.n_slots, self.n_actions), np.int32)
        self._check(self.lib.azsp_root_counts(self._h, _ptr(out, C.c_int32)))
        return out

    def play(self, uniforms: np.ndarray) -> L.PlyResult:
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        if u.size != self.n_slots:
            raise ValueError(f"need {self.n_slots} uniforms, got {u.size}")
        res = L.PlyResult()
        self._check(self.lib.azsp_play(self._h, _ptr(u, C.c_double), C.byref(res)))
        return res

    def play_begin(self, uniforms: np.ndarray) -> int:
        """Phase 1 of a sharded ply: sample + apply moves; returns how many games of this shard ended."""
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        if u.size != self.n_slots:
            raise ValueError(f"need {self.n_slots} uniforms, got {u.size}")
        n_done = C.c_int32(0)
        self._check(self.lib.azsp_play_begin(self._h, _ptr(u, C.c_double), C.byref(n_done)))
        return int(n_done.value)

    def play_commit(self, allow: int) -> L.PlyResult:
        """Phase 2: emit the first `allow` completions of this ply (slot order), restart finished slots."""
        res = L.PlyResult()
        self._check(self.lib.azsp_play_commit(self._h, int(allow), C.byref(res)))
        return res

    def fetch_episodes(self, first: int, n: int, arrays: bool = True) -> dict[str, np.ndarray]:
        """Episodes [first, first+n) as SoA numpy buffers; `arrays=False` copies only the episode table."""
        out = {"ep_offset": np.zeros(n + 1, np.int64), "ep_slot": np.zeros(n, np.int32), "ep_ply": np.zeros(n, np.int32)}
        cap = n * self.max_plies if arrays else 0
        if arrays:
            out.update(states=np.empty((cap, self.cells), np.int8), policies=np.empty((cap, self.n_actions), np.float64),
                       values=np.empty(cap, np.int32))
        self._check(self.lib.azsp_fetch_episodes(self._h, first, n, _ptr(out["ep_offset"], C.c_int64),
                                                 _ptr(out["ep_slot"], C.c_int32), _ptr(out["ep_ply"], C.c_int32),
                                                 _ptr(out.get("states"), C.c_int8), _ptr(out.get("policies"), C.c_double),
                                                 _ptr(out.get("values"), C.c_int32), cap))
        ns = int(out["ep_offset"][n])
        for k in ("states", "policies", "values"):
            if k in out:
                out[k] = out[k][:ns]
        return out

    def export_records_device(self, first: int, n: int, dev_ptr: int, cap_records: int) -> int:
        got = self.lib.azsp_export_records_device(self._h, first, n, C.c_void_p(dev_ptr), cap_records)
        if got < 0:
            L.check(int(got))
        return int(got)

    def export_dataset_device(self, first: int, n: int, planes_ptr: int, policies_ptr: int, values_ptr: int,
                              cap_samples: int) -> int:
        got = self.lib.azsp_export_dataset_device(self._h, first, n, C.c_void_p(planes_ptr), C.c_void_p(policies_ptr),
                                                  C.c_void_p(values_ptr), cap_samples)
        if got < 0:
            L.check(int(got))
        return int(got)

    def torch_device(self, requested=None):
        """torch.device of this engine's GPU: where the azsp_export_*_device destinations must live (the engine's kernel
        writes through raw pointers on its own device and stream).  A different `requested` device is refused."""
        import torch
        dev = torch.device("cuda", self.device)
        if requested is not None:
            req = torch.device(requested)
            if req.type != "cuda" or (req.index is not None and req.index != self.device):
                raise ValueError(f"the engine lives on {dev}; it cannot export into memory on {req}")
        return dev

    def sync_torch_stream(self) -> None:
        """Wait for PyTorch's current stream on the engine's device: a block torch.empty hands out may still have work
        pending there, and the engine's private stream does not order itself behind it."""
        import torch
        torch.cuda.current_stream(self.device).synchronize()

    @property
    def record_bytes(self) -> int:
        return int(self.lib.azsp_record_bytes(self.game_id))

    # ---- stats --------------------------------------------------------------------------------
    def stats(self) -> dict[str, float]:
        s = L.Stats()
        self._check(self.lib.azsp_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.Stats._fields_}

    def reset_stats(self) -> None:
        self._check(self.lib.azsp_reset_stats(self._h))

    def sync(self) -> None:
        self._check(self.lib.azsp_sync(self._h))
