"""ctypes binding of ``libmlvdb_hip.so`` (C ABI declared in include/mlvdb_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C mlvectordb_amd/csrc``.
Loading fails loudly: there is no CPU fallback behind this module.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_NAME = "libmlvdb_hip.so"
_LIB_PATH = Path(__file__).resolve().parent / "csrc" / _LIB_NAME

OK = 0
ERR_OVERFLOW = 7
SPACE_CODES = {"l2": 0, "cosine": 1, "ip": 2}
STRATEGY_CODES = {"auto": 0, "exact": 1, "filter": 2}
MAX_TOPK = 64
MAX_TOPK_PAGED = 16384
ABI_VERSION = 6


class Stats(C.Structure):
    _fields_ = [
        ("strategy_used", C.c_int32),
        ("scan_launches", C.c_int32),
        ("rows_scanned", C.c_int64),
        ("candidates_rescored", C.c_int64),
        ("fallback_queries", C.c_int64),
        ("scan_ms", C.c_double),
        ("total_ms", C.c_double),
        ("bound_dtype", C.c_int32),
        ("reserved", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/mlvdb_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "mlvdb_abi_version": (C.c_int, []),
    "mlvdb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mlvdb_last_global_error": (C.c_char_p, []),
    "mlvdb_index_create": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_P)]),
    "mlvdb_index_destroy": (C.c_int, [_P]),
    "mlvdb_last_error": (C.c_char_p, [_P]),
    "mlvdb_index_append": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mlvdb_index_append_device": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mlvdb_index_tombstone": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mlvdb_index_compact": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mlvdb_index_counts": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mlvdb_index_reset": (C.c_int, [_P, C.c_int32]),
    "mlvdb_index_get_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "mlvdb_index_get_rows_at": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mlvdb_search_batch": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    "mlvdb_search_batch_filtered": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "mlvdb_search_batch_ex": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "mlvdb_search_batch_device": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "mlvdb_range_batch": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int64, _P, _P, _P]),
    "mlvdb_range_batch_packed": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int64, C.c_int64, _P, _P, _P, _P]),
    "mlvdb_pair_distances": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P]),
    "mlvdb_index_set_strategy": (C.c_int, [_P, C.c_int32]),
    "mlvdb_index_set_tuning": (C.c_int, [_P, C.c_char_p]),
    "mlvdb_index_get_tuning": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "mlvdb_index_set_profiling": (C.c_int, [_P, C.c_int32]),
    "mlvdb_index_last_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "mlvdb_layout_offset": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "mlvdb_layout_ld": (C.c_int32, [C.c_int32]),
}

_lib = None


def library_path() -> Path:
    return Path(os.environ.get("MLVDB_HIP_LIBRARY", _LIB_PATH))


def load() -> C.CDLL:
    """Load the shared library and type every entry point; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64
    # (SONAME libamdhip64.so.7, the name this library links).  If torch is importable, load it first
    # so that the dynamic linker resolves our dependency to the runtime torch already initialised --
    # two runtimes in one process leave the second without a GPU, and device pointers / streams
    # handed over from torch must belong to the runtime that launches our kernels.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP scan library is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C mlvectordb_amd/csrc`. "
            f"There is no CPU fallback for the search path.")
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.mlvdb_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib
