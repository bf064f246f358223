"""Generates the golden vectors under tests/golden/ from THIS repo's oracle.

The reference (SudYar/MLVectorDB) holds no golden vectors and its arithmetic (hnswlib 0.8.0)
is not available offline, so these files pin the oracle's own canonical arithmetic against
regressions and give the GPU tests a fixture that does not depend on the oracle code path at
test time.  Numeric scores: parity unpinned w.r.t. the reference (see oracle/__init__.py).

Inputs are never stored: they are regenerated from the seed in ``meta``.
Run:  python -m tests.golden.make_golden
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

HERE = Path(__file__).parent

CASES = [
    # name, seed, n, d, nq, k, space, deleted_frac, dup
    ("knn_l2_1000x16", 11, 1000, 16, 8, 5, "l2", 0.0, True),
    ("knn_cos_1000x16", 12, 1000, 16, 8, 5, "cosine", 0.1, True),
    ("knn_ip_1000x16", 13, 1000, 16, 8, 5, "ip", 0.1, False),
    ("knn_l2_4096x128", 14, 4096, 128, 16, 5, "l2", 0.1, True),
    ("knn_cos_4096x128", 15, 4096, 128, 16, 5, "cosine", 0.0, True),
    ("knn_cos_10000x128", 16, 10000, 128, 32, 5, "cosine", 0.1, False),
    ("knn_l2_2048x768", 17, 2048, 768, 8, 10, "l2", 0.0, True),
    ("knn_cos_2048x768", 18, 2048, 768, 8, 10, "cosine", 0.1, True),
    ("knn_cos_3000x192", 19, 3000, 192, 24, 10, "cosine", 0.05, True),
]


def regenerate_inputs(meta):
    from tests.helpers import deleted_mask, make_case

    rows, qs = make_case(meta["seed"], meta["n"], meta["d"], meta["nq"], dup=meta["dup"])
    return rows, qs, deleted_mask(meta["seed"], meta["n"], meta["deleted_frac"])


def main():
    from oracle import exact_scan

    for name, seed, n, d, nq, k, space, frac, dup in CASES:
        meta = dict(seed=seed, n=n, d=d, nq=nq, k=k, space=space, deleted_frac=frac, dup=dup)
        rows, qs, deleted = regenerate_inputs(meta)
        labels, dist, counts = exact_scan.knn(qs, rows, k, space, deleted=deleted)
        np.savez_compressed(HERE / f"{name}.npz", labels=labels, dist=dist, counts=counts, meta=json.dumps(meta))
        print(name, labels.shape)


if __name__ == "__main__":
    main()
