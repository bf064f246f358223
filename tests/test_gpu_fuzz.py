"""Seeded shape fuzzing of the GPU path against the oracle (AUTO strategy, as a caller would use it).

Each case draws corpus size, dimension (the int8 filter on a shadow zero-padded to a multiple of 256 columns -- 64 and
128 keep the bf16 shadow --; below 64 the exact scan keeps the batches that stream fewer bytes that way), batch size (1 .. several 256-query passes), k (to 300:
big-k passes), space, tombstone fraction and append chunking from a seeded generator, so the cases are the same on every run.  Bar: ids bit-exact, scores within 1e-5 (tests/helpers.py).
"""
import numpy as np
import pytest

from mlvectordb_amd.engine import HipScanEngine
from tests.helpers import assert_knn_matches, deleted_mask, make_case, oracle_knn

pytestmark = pytest.mark.gpu

DIMS = [64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 20, 100, 130, 700]


def draw(seed):
    rng = np.random.default_rng(seed)
    d = int(DIMS[rng.integers(len(DIMS))])
    budget = 40_000_000  # floats in the corpus: keeps the fp64 oracle to about a second per case
    n = int(rng.integers(1, max(2, min(400_000, budget // d))))
    nq = int(rng.choice([1, 3, 11, 12, 40, 255, 256, 257, 600]))
    k = int(rng.choice([1, 5, 10, 33, 64, 100, 300]))  # (round 4: top_k > 64 stays on the filter path where AUTO picks it)
    space = str(rng.choice(["l2", "cosine", "ip"]))
    # the fp64 NumPy oracle: a GEMM for cosine / ip, element-wise (q - x)^2 for l2 (~2 s per 1e9 terms on the GPU box's host share)
    cap = 3e9 if space == "l2" else 2e10
    if n * nq * d > cap:
        nq = max(1, int(cap // (n * d)))
    frac = float(rng.choice([0.0, 0.0, 0.1, 0.5, 0.95]))
    chunks = int(rng.integers(1, 5))
    return n, d, nq, k, space, frac, chunks


@pytest.mark.parametrize("seed", range(100, 117))
def test_fuzzed_shapes_match_oracle(seed):
    n, d, nq, k, space, frac, chunks = draw(seed)
    rows, qs = make_case(seed, n, d, nq, dup=n > 50)
    deleted = deleted_mask(seed, n, frac)
    eng = HipScanEngine(d, space, device=0)
    try:
        for part in np.array_split(rows, chunks):
            if len(part):
                eng.append(part)
        if deleted.any():
            eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, k)
    finally:
        eng.close()
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"fuzz seed {seed}: n{n} d{d} nq{nq} k{k} {space} del{frac}")
