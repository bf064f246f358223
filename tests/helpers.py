"""Shared test helpers (inputs are always regenerated from seeds)."""
from __future__ import annotations

import numpy as np

from oracle import exact_scan

SCORE_ATOL = 1e-5  # north star: scores within 1e-5 (fp32), ids bit-exact


def make_case(seed: int, n: int, d: int, nq: int, *, dup: bool = False, clustered: bool = False):
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, d), dtype=np.float32)
    if clustered:  # many near-ties: rows are small perturbations of a few centres
        centres = rng.standard_normal((8, d), dtype=np.float32)
        rows = centres[rng.integers(0, 8, n)] + 1e-3 * rows
    if dup and n >= 8:  # exact duplicates: tie-break by label must hold
        rows[n // 2] = rows[1]
        rows[n - 1] = rows[1]
        rows[3] = rows[2]
    qs = rng.standard_normal((nq, d), dtype=np.float32)
    if nq >= 2 and n >= 2:
        qs[0] = rows[1]  # a stored vector queried back: distance exactly 0 (l2) / similarity 1
    return rows, qs


def deleted_mask(seed: int, n: int, frac: float) -> np.ndarray:
    if frac <= 0:
        return np.zeros(n, dtype=bool)
    return np.random.default_rng(seed + 99).random(n) < frac


def assert_knn_matches(got, want, tag: str):
    gl, gd, gc = got
    wl, wd, wc = want
    if not (np.array_equal(gl, wl) and np.array_equal(gc, wc)):
        from tests.conftest import dump_mismatch

        dump_mismatch(tag.replace("/", "_"), got_labels=gl, want_labels=wl, got_dist=gd, want_dist=wd)
        bad = np.nonzero((gl != wl).any(axis=1))[0]
        raise AssertionError(f"{tag}: ids differ for {bad.size} queries, first {bad[:5]}: "
                             f"got {gl[bad[0]]} want {wl[bad[0]]} (d got {gd[bad[0]]} want {wd[bad[0]]})")
    fin = np.isfinite(wd)
    assert np.array_equal(np.isfinite(gd), fin), f"{tag}: padding differs"
    err = np.abs(gd[fin] - wd[fin]).max() if fin.any() else 0.0
    assert err <= SCORE_ATOL, f"{tag}: max |distance error| {err}"


def oracle_knn(qs, rows, k, space, deleted=None):
    return exact_scan.knn(qs, rows, k, space, deleted=deleted)
