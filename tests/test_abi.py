"""The C-ABI library loads without a GPU and exports every symbol the header declares."""
import ctypes as C
import re
from pathlib import Path

import numpy as np

from mlvectordb_amd import _native

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "mlvdb_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlvdb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mlvdb_hip.h but not exported"
    assert sorted(_native.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.mlvdb_abi_version() == _native.ABI_VERSION


def test_header_constants_match_binding():
    text = (ROOT / "include" / "mlvdb_hip.h").read_text()
    consts = dict(re.findall(r"#define\s+(MLVDB_[A-Z0-9_]+)\s+(-?\d+)", text))
    assert int(consts["MLVDB_SPACE_L2"]) == _native.SPACE_CODES["l2"]
    assert int(consts["MLVDB_SPACE_COSINE"]) == _native.SPACE_CODES["cosine"]
    assert int(consts["MLVDB_SPACE_IP"]) == _native.SPACE_CODES["ip"]
    assert int(consts["MLVDB_MAX_TOPK"]) == _native.MAX_TOPK
    assert int(consts["MLVDB_ERR_OVERFLOW"]) == _native.ERR_OVERFLOW
    assert int(consts["MLVDB_STRATEGY_FILTER"]) == _native.STRATEGY_CODES["filter"]


def test_stats_struct_layout_matches_header():
    assert C.sizeof(_native.Stats) == 4 + 4 + 8 + 8 + 8 + 8 + 8 + 4 + 4


def test_no_device_is_a_loud_error_not_a_fallback(gpu_available):
    if gpu_available:
        return
    lib = _native.load()
    n = C.c_int(-1)
    assert lib.mlvdb_device_count(C.byref(n)) == 3 and n.value == 0  # MLVDB_ERR_NO_DEVICE
    h = C.c_void_p()
    assert lib.mlvdb_index_create(0, 16, 0, 0, C.byref(h)) == 3 and not h.value
    assert b"no HIP device" in lib.mlvdb_last_global_error()
    import pytest

    from mlvectordb_amd import Index, Vector

    with pytest.raises(RuntimeError):
        Index(space="l2").add([Vector(values=[1.0, 2.0])], "ns")  # product path: no CPU engine behind it


def test_panel_layout_restatement():
    """layout.h: offset(row, col) = (row/16)*16*ld + (col/16)*256 + (row%16)*16 + col%16."""
    lib = _native.load()
    for dim in (1, 3, 16, 17, 100, 128, 768, 1000):
        ld = lib.mlvdb_layout_ld(dim)
        assert ld == (dim + 15) // 16 * 16
        rows = np.arange(0, 70)
        cols = np.arange(0, ld)
        seen = set()
        for r in rows:
            for c in cols[:: max(1, ld // 40)]:
                want = (r // 16) * 16 * ld + (c // 16) * 256 + (r % 16) * 16 + c % 16
                got = lib.mlvdb_layout_offset(int(r), int(c), ld)
                assert got == want
                seen.add(got)
        # a wave's load: lane l = 16*g + r reads the float4 of row r, columns 4g..4g+3 of a group; the 64
        # float4 tile one 1 KiB group exactly, and a row's 16 columns are one contiguous 64-byte piece
        base = lib.mlvdb_layout_offset(32, 16, ld)
        offs = set()
        for lane in range(64):
            g, r = divmod(lane, 16)
            o = lib.mlvdb_layout_offset(32 + r, 16 + 4 * g, ld)
            assert o == base + 16 * r + 4 * g
            offs.update(range(o - base, o - base + 4))
        assert offs == set(range(256))
