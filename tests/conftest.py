import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

OUT_DIR = ROOT / "gpurun_out"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ab: tuning variants that exist only in the `make AB=1` library (tools/gpu_suite.sh --ab); skipped elsewhere")


def _has_gpu() -> bool:
    try:
        import ctypes as C

        from mlvectordb_amd import _native

        n = C.c_int(0)
        return _native.load().mlvdb_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


@pytest.fixture(autouse=True)
def _gpu_tests_fail_loudly_without_gpu(request, gpu_available):
    # a -m gpu test must never silently pass on a fallback: no GPU / no library is an error
    if request.node.get_closest_marker("gpu") and not gpu_available:
        pytest.fail("gpu-marked test but libmlvdb_hip.so cannot see a HIP device (no CPU fallback exists)")


def dump_mismatch(name: str, **arrays) -> None:
    """Keep the evidence of a GPU/oracle mismatch where gpurun brings it back."""
    try:
        OUT_DIR.mkdir(exist_ok=True)
        np.savez(OUT_DIR / f"mismatch_{name}.npz", **arrays)
    except Exception:
        pass
