"""ctypes binding of csrc/libazsp.so (C ABI: include/azsp.h).

The library is the product path: if it is missing or cannot be loaded this module raises -- there
is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AZSP_LIB: another build of the same library (A/B timing of kernel variants on one box); default = the in-tree build
LIB_PATH = os.environ.get("AZSP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libazsp.so")

GAME_CONNECT4, GAME_CONNECT2, GAME_GOMOKU9 = 0, 1, 2
EVAL_RESNET, EVAL_HASH, EVAL_HOST = 0, 1, 2
FLAG_TIMING = 1
FLAG_F16_SPLIT = 2
FLAG_TREE_REUSE = 4
FLAG_EVAL_CACHE = 8
RULES_SIMPLE, RULES_GEN1, RULES_SIMPLE_NUMPY1 = 0, 1, 2
ABI_VERSION = 3


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("game", C.c_int32), ("n_slots", C.c_int32), ("n_sims", C.c_int32),
                ("device", C.c_int32), ("res_blocks", C.c_int32), ("channels", C.c_int32), ("flags", C.c_uint32),
                ("hash_salt", C.c_uint64), ("n_groups", C.c_int32), ("arena_episodes", C.c_int32),
                ("search_rules", C.c_int32), ("eval_cache_entries", C.c_int32), ("timing_sample", C.c_int32),
                ("reserved", C.c_int32)]


class PlyResult(C.Structure):
    _fields_ = [("ply", C.c_int32), ("n_finished", C.c_int32), ("episodes_total", C.c_int64),
                ("samples_total", C.c_int64), ("uniforms_used", C.c_int32), ("finished", C.c_int32),
                ("last_counted_slot", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("plies", C.c_uint64), ("rounds", C.c_uint64), ("evals", C.c_uint64),
                ("terminal_leaves", C.c_uint64), ("infer_launches", C.c_uint64), ("infer_ms", C.c_double),
                ("infer_boards", C.c_uint64), ("select_ms", C.c_double), ("expand_ms", C.c_double),
                ("play_ms", C.c_double), ("select_launches", C.c_uint64), ("expand_launches", C.c_uint64),
                ("path_nodes", C.c_uint64), ("infer_union_ms", C.c_double), ("evals_executed", C.c_uint64),
                ("infer_exec_flops_per_board", C.c_double), ("eval_cache_entries", C.c_uint64),
                ("infer_launches_timed", C.c_uint64), ("infer_exec_flops", C.c_double)]


class SearchOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("sampling_plies", C.c_int32), ("noise_eps", C.c_double),
                ("c_init", C.c_double), ("c_base", C.c_double)]


PREDICT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int8), C.c_int32, C.POINTER(C.c_float),
                         C.POINTER(C.c_float))

_P = C.POINTER
# name -> (restype, argtypes): every symbol include/azsp.h declares
SIGNATURES = {
    "azsp_abi_version": (C.c_int, []),
    "azsp_create": (C.c_int, [_P(Config), _P(C.c_void_p)]),
    "azsp_destroy": (None, [C.c_void_p]),
    "azsp_last_error": (C.c_char_p, []),
    "azsp_game_info": (C.c_int, [C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "azsp_game_rules": (C.c_int, [C.c_int32, _P(C.c_int8), C.c_int32, _P(C.c_int8), _P(C.c_int32), _P(C.c_int32)]),
    "azsp_load_resnet": (C.c_int, [C.c_void_p, _P(C.c_float), C.c_size_t]),
    "azsp_resnet_blob_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "azsp_set_evaluator": (C.c_int, [C.c_void_p, C.c_int32, PREDICT_FN, C.c_void_p]),
    "azsp_predict": (C.c_int, [C.c_void_p, _P(C.c_int8), C.c_int32, _P(C.c_float), _P(C.c_float)]),
    "azsp_run_batch": (C.c_int, [C.c_void_p, _P(C.c_int8), C.c_int32, _P(C.c_int32), _P(C.c_float), _P(C.c_double),
                                 _P(C.c_float), _P(C.c_int32)]),
    "azsp_reset": (C.c_int, [C.c_void_p]),
    "azsp_root_values": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_int32)]),
    "azsp_set_initial_board": (C.c_int, [C.c_void_p, _P(C.c_int8)]),
    "azsp_fetch_moves": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _P(C.c_int32), C.c_int64]),
    "azsp_search": (C.c_int, [C.c_void_p]),
    "azsp_root_counts": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "azsp_set_search_options": (C.c_int, [C.c_void_p, _P(SearchOptions)]),
    "azsp_search_with_noise": (C.c_int, [C.c_void_p, _P(C.c_double)]),
    "azsp_play": (C.c_int, [C.c_void_p, _P(C.c_double), _P(PlyResult)]),
    "azsp_play_begin": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_int32)]),
    "azsp_play_commit": (C.c_int, [C.c_void_p, C.c_int64, _P(PlyResult)]),
    "azsp_fetch_episodes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32),
                                      _P(C.c_int8), _P(C.c_double), _P(C.c_int32), C.c_int64]),
    "azsp_export_records_device": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]),
    "azsp_record_bytes": (C.c_int32, [C.c_int32]),
    "azsp_export_dataset_device": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int64]),
    "azsp_probe_clock": (C.c_int, [C.c_void_p, _P(C.c_int8), C.c_int32, _P(C.c_double), _P(C.c_double)]),
    "azsp_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "azsp_reset_stats": (C.c_int, [C.c_void_p]),
    "azsp_sync": (C.c_int, [C.c_void_p]),
}

_lib = None


class AzspError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libazsp.so and bind every exported symbol.  Raises if the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AzspError(f"{LIB_PATH} is missing: build it with `python alphazero-implementation_amd/build.py` "
                        "(the self-play engine has no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.  If libazsp pulled in the system copy
    # first, a later `import torch` would load a second runtime, which then finds no GPU ("No HIP GPUs are
    # available").  Importing torch first makes libazsp bind to the copy that is already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.azsp_abi_version() != ABI_VERSION:
        raise AzspError(f"libazsp ABI {lib.azsp_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise AzspError(f"libazsp error {rc}: {load().azsp_last_error().decode(errors='replace')}")
