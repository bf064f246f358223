"""Tuning variants of the filter scan (`make AB=1`): every generated geometry against the oracle.

Not part of `-m gpu`: these bodies are not in the default library.  Run on the GPU box by tools/gpu_suite.sh --ab, which
builds mlvectordb_amd/csrc/libmlvdb_hip_ab.so and exports MLVDB_HIP_LIBRARY; anywhere else (no GPU, or the default
library loaded) every case is skipped."""
import os

import pytest

from tests.helpers import assert_knn_matches, deleted_mask, make_case, oracle_knn

pytestmark = pytest.mark.ab

BF16 = {"MLVDB_I8": "0", "MLVDB_SHADOW": "bf16"}
AB_VARIANTS = [
    {"MLVDB_SCAN_PRIO": "0", "MLVDB_SCAN_VA": "0"},  # int8 body, AccVGPR accumulators (round 1), without the wave priorities
    {"MLVDB_SCAN_VA": "0"},                          # ... with them
    {**BF16, "MLVDB_SCAN_DMA": "0"},                 # bf16 body, Q staged through registers (global -> VGPR -> ds_write)
    {**BF16, "MLVDB_SCAN_NW": "4"},                  # two 4-wave workgroups per CU
    {**BF16, "MLVDB_SCAN_MT": "4"},                  # one wave per SIMD, 64 rows per wave
    {**BF16, "MLVDB_SCAN_NT": "0"},                  # (cosine only) temporal X loads
    {**BF16, "MLVDB_SCAN_ASM": "0"},                 # the hipcc-scheduled kernel
    {**BF16, "MLVDB_SCAN_STAG": "1"},                # later half of the waves half a tile behind (rotated k origin)
    {**BF16, "MLVDB_SCAN_PRIO": "1"},                # (cosine) progress-based wave priorities on the bf16 body
] + [{"MLVDB_SCAN_VAR": str(v)} for v in (214, 215, 216, 217, 218, 219, 220, 221, 222, 228, 229, 230, 231, 232, 233, 235, 236)]


@pytest.fixture(autouse=True)
def _needs_the_ab_library(gpu_available):
    lib = os.environ.get("MLVDB_HIP_LIBRARY", "")
    if not gpu_available or "_ab" not in os.path.basename(lib) and "_diag" not in os.path.basename(lib):
        pytest.skip("AB variants need a GPU and MLVDB_HIP_LIBRARY=.../libmlvdb_hip_ab.so (make AB=1; tools/gpu_suite.sh --ab)")


@pytest.mark.parametrize("variant", AB_VARIANTS, ids=lambda v: ",".join(f"{k[6:]}={x}" for k, x in v.items()))
@pytest.mark.parametrize("space,d", [("cosine", 128), ("l2", 192), ("ip", 64), ("cosine", 768), ("l2", 1536), ("ip", 768), ("l2", 256)])
def test_ab_scan_variants_agree_with_oracle(variant, space, d, monkeypatch):
    from mlvectordb_amd.engine import HipScanEngine

    for key, val in variant.items():  # read when the handle is created
        monkeypatch.setenv(key, val)
    n = 150_001 if d <= 192 else (70_003 if d <= 768 else 33_001)
    rows, qs = make_case(300 + d, n, d, 40, dup=True)
    deleted = deleted_mask(7, n, 0.05)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        for part in (rows[: n // 3], rows[n // 3: 2 * n // 3], rows[2 * n // 3:]):
            eng.append(part)
        eng.tombstone(deleted.nonzero()[0])
        got = eng.search(qs, 10)
        stats = eng.last_stats()
    finally:
        eng.close()
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0
    assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"ab variant {variant}/{space}/d{d}")


SMALL_KNOBS = [
    {"MLVDB_SMALL_SEED": "0"},          # dense int8 seeding pass + refine, fused finish
    {"MLVDB_SMALL_FINISH": "0"},        # prefix seed, the three finishing kernels
    {"MLVDB_SMALL_NQ": "0"},            # round 2's structure
    {"MLVDB_SMALL_NQ": "8"},            # both steps for up to 8 queries
    {"MLVDB_NARROW_I8_MAX": "8"},       # the int8 narrow kernel (round 3's scan for 1-8 queries) instead of the 4-tile assembly body
    {"MLVDB_SCAN_NQT": "16"},           # every pass padded to 16 query tiles (round 3)
]
SMALL_CASES = [
    ("cosine", 768, 1, 70_003, 10, 0.05), ("cosine", 768, 2, 70_003, 10, 0.05), ("l2", 768, 1, 40_001, 10, 0.3),
    ("ip", 256, 2, 150_001, 1, 0.0), ("l2", 1536, 1, 33_001, 64, 0.05), ("cosine", 256, 1, 150_001, 33, 0.9),
    ("ip", 768, 1, 3_000, 10, 0.05), ("l2", 256, 2, 9_000, 64, 0.995), ("cosine", 768, 5, 70_003, 10, 0.05), ("l2", 768, 40, 40_001, 10, 0.05),
]


@pytest.mark.parametrize("knobs", SMALL_KNOBS, ids=lambda v: ",".join(f"{k[6:]}={x}" for k, x in v.items()))
@pytest.mark.parametrize("space,d,nq,n,k,frac", SMALL_CASES)
def test_alternative_small_batch_and_l2_paths_agree_with_oracle(space, d, nq, n, k, frac, knobs, monkeypatch):
    from mlvectordb_amd.engine import HipScanEngine

    for key, val in knobs.items():
        monkeypatch.setenv(key, val)
    rows, qs = make_case(900 + d + nq + k, n, d, nq, dup=True)
    deleted = deleted_mask(13, n, frac)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        for part in (rows[: n // 2], rows[n // 2:]):
            eng.append(part)
        eng.tombstone(deleted.nonzero()[0])
        got = eng.search(qs, k)
        stats = eng.last_stats()
    finally:
        eng.close()
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0 and stats["bound_dtype"] == 2
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"alt {knobs}/{space}/d{d}/nq{nq}/k{k}")
