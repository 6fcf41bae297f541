"""ctypes wrapper of oracle/az_oracle.c (CPU ORACLE -- test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libazoracle.so")
GAME_ID = {"connect4": 0, "connect2": 1, "gomoku9": 2}
GEOM = {0: (6, 7, 7), 1: (1, 4, 4), 2: (9, 9, 81)}
MAX_PLIES = {0: 42, 1: 4, 2: 81}

PREDICT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int8), C.c_int32, C.POINTER(C.c_float),
                         C.POINTER(C.c_float))
UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "az_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.azo_run_batch.restype = C.c_int
        _lib.azo_generate.restype = C.c_int
    return _lib


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def make_predict_cb(predict, cells_shape, n_actions):
    """Wrap a python BaseModel.predict-style callable (list of int boards -> (list p, list v))."""
    cells = int(np.prod(cells_shape))

    def cb(_user, boards, n, p_out, v_out):
        try:
            b = np.ctypeslib.as_array(boards, shape=(n, cells)).astype(np.int64)
            ps, vs = predict([x.reshape(cells_shape) for x in b])
            np.ctypeslib.as_array(p_out, shape=(n, n_actions))[:] = np.asarray(ps, dtype=np.float32)
            np.ctypeslib.as_array(v_out, shape=(n,))[:] = np.asarray(vs, dtype=np.float32)
            return 0
        except Exception:           # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1
    return PREDICT_FN(cb)


def run_batch(game: str, sims: int, boards: np.ndarray, to_plays=None, salt: int = 0, predict_cb=None):
    gid = GAME_ID[game]
    rows, cols, A = GEOM[gid]
    boards = np.ascontiguousarray(np.asarray(boards).reshape(len(boards), rows * cols), dtype=np.int8)
    n = boards.shape[0]
    tp = None if to_plays is None else np.ascontiguousarray(to_plays, dtype=np.int32)
    out = {"counts": np.zeros((n, A), np.int32), "child_w": np.zeros((n, A), np.float32),
           "prior": np.zeros((n, A), np.float64), "root_w": np.zeros(n, np.float32),
           "root_n": np.zeros(n, np.int32), "n_nodes": np.zeros(n, np.int32),
           "stats": np.zeros(7, np.int64)}
    rc = lib().azo_run_batch(gid, sims, _ptr(boards, C.c_int8), _ptr(tp, C.c_int32), n,
                             predict_cb if predict_cb is not None else PREDICT_FN(), None,
                             C.c_uint64(salt), _ptr(out["counts"], C.c_int32), _ptr(out["child_w"], C.c_float),
                             _ptr(out["prior"], C.c_double), _ptr(out["root_w"], C.c_float),
                             _ptr(out["root_n"], C.c_int32), _ptr(out["n_nodes"], C.c_int32),
                             _ptr(out["stats"], C.c_int64))
    if rc:
        raise RuntimeError(f"azo_run_batch failed: {rc}")
    return out


def generate(game: str, sims: int, n_slots: int, salt: int = 0, seed: int | None = 0, predict_cb=None,
             uniform_cb=None, trace: bool = True):
    """Returns dict with episodes in the azsp_fetch_episodes SoA layout."""
    gid = GAME_ID[game]
    rows, cols, A = GEOM[gid]
    cells = rows * cols
    max_plies = MAX_PLIES[gid]
    cap = n_slots * max_plies
    trace_cap = n_slots * max_plies * 8 if trace else 0
    o = {"ep_offset": np.zeros(n_slots + 1, np.int64), "ep_slot": np.zeros(n_slots, np.int32),
         "ep_ply": np.zeros(n_slots, np.int32), "states": np.zeros((cap, cells), np.int8),
         "policies": np.zeros((cap, A), np.float64), "values": np.zeros(cap, np.int32),
         "trace_counts": np.zeros((max(trace_cap, 1), A), np.int32),
         "trace_actions": np.zeros(max(trace_cap, 1), np.int32), "stats": np.zeros(7, np.int64)}
    rows_out = C.c_int64(0)
    nxt = C.c_double(0)
    rc = lib().azo_generate(gid, sims, n_slots, predict_cb if predict_cb is not None else PREDICT_FN(), None,
                            C.c_uint64(salt), C.c_int64(-1 if seed is None else seed),
                            uniform_cb if uniform_cb is not None else UNIFORM_FN(), None,
                            _ptr(o["ep_offset"], C.c_int64), _ptr(o["ep_slot"], C.c_int32), _ptr(o["ep_ply"], C.c_int32),
                            _ptr(o["states"], C.c_int8), _ptr(o["policies"], C.c_double), _ptr(o["values"], C.c_int32),
                            C.c_int64(cap), _ptr(o["trace_counts"], C.c_int32) if trace else None,
                            _ptr(o["trace_actions"], C.c_int32) if trace else None, C.c_int64(trace_cap),
                            C.byref(rows_out), C.byref(nxt), _ptr(o["stats"], C.c_int64))
    if rc < 0:
        raise RuntimeError(f"azo_generate failed: {rc}")
    ns = int(o["ep_offset"][rc])
    o["n_episodes"] = rc
    o["states"], o["policies"], o["values"] = o["states"][:ns], o["policies"][:ns], o["values"][:ns]
    o["trace_rows"] = rows_out.value
    o["trace_counts"] = o["trace_counts"][:min(rows_out.value, trace_cap)]
    o["trace_actions"] = o["trace_actions"][:min(rows_out.value, trace_cap)]
    o["next_uniform"] = nxt.value
    o["shape"] = (rows, cols) if rows > 1 else (cols,)
    return o


def np_uniform_cb():
    """Uniform source that consumes the global legacy numpy RNG exactly like the reference."""
    return UNIFORM_FN(lambda _u: float(np.random.random_sample()))


def mt_uniforms(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.float64)
    lib().azo_mt_uniforms(C.c_uint32(seed), _ptr(out, C.c_double), n)
    return out


def hash_eval(game: str, board: np.ndarray, salt: int = 0):
    gid = GAME_ID[game]
    A = GEOM[gid][2]
    b = np.ascontiguousarray(np.asarray(board).reshape(-1), dtype=np.int8)
    p = np.zeros(A, np.float32)
    v = C.c_float(0)
    lib().azo_hash_eval(gid, _ptr(b, C.c_int8), C.c_uint64(salt), _ptr(p, C.c_float), C.byref(v))
    return p, np.float32(v.value)
