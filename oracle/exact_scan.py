"""Exact exhaustive kNN / range query in NumPy -- the parity oracle (test infrastructure).

What it restates
----------------
* hnswlib 0.8.0 distance spaces, as used by the reference through
  ``hnswlib.Index(space=...)`` (reference src/mlvectordb/implementations/index.py:36,111):
    l2      d = sum_i (q_i - x_i)^2                      (squared; no sqrt)
    cosine  rows and queries are L2-normalised with ``x / (|x| + 1e-30)`` when they are
            inserted / queried, d = 1 - <q^, x^>
    ip      d = 1 - <q, x>
  hnswlib walks an HNSW graph (approximate); this oracle is the recall-1.0 limit of that:
  every live row is scored.
* ``knn_query`` contract (index.py:111): nearest first, tombstoned labels skipped.
* the post-processing of ``Index.search`` (index.py:121-129): score = float(dist), and
  ``1 - dist`` when the *metric argument* is "cosine" (whatever the space searched).

Canonical arithmetic (what "bit-exact ids" means for this repo): distances are evaluated
in float64 from the float32 inputs, rows are ranked by (distance ascending, label
ascending), and the float64 distance is rounded once to float32 for return.  hnswlib
itself sums in float32 SIMD order; two correct float32 implementations differ by ~1e-7,
so ranking on the float64 value is the only order both a CPU and a GPU implementation can
reproduce.  Numeric score values: parity unpinned (no reference test asserts one).
"""
from __future__ import annotations

import numpy as np

SPACES = ("l2", "cosine", "ip")

_CHUNK_ROWS = 65536


def _as_f32_2d(a) -> np.ndarray:
    a = np.asarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    return np.ascontiguousarray(a)


def exact_distances(queries, rows, space: str) -> np.ndarray:
    """float64 distance matrix [nq, n] of float32 ``queries`` against float32 ``rows``.

    Follows the hnswlib space definitions cited in the module docstring; the products of
    two float32 values are exact in float64, so the only rounding is in the summation.
    """
    if space not in SPACES:
        raise ValueError(f"unknown space {space!r}")
    q = _as_f32_2d(queries).astype(np.float64)
    x = _as_f32_2d(rows)
    nq, n = q.shape[0], x.shape[0]
    out = np.empty((nq, n), dtype=np.float64)
    if space == "cosine":
        qinv = 1.0 / (np.sqrt((q * q).sum(axis=1)) + 1e-30)
    for s in range(0, n, _CHUNK_ROWS):
        xc = x[s:s + _CHUNK_ROWS].astype(np.float64)
        if space == "l2":
            # direct (q - x)^2 form: exact zero for a stored vector queried back,
            # never negative (reference tests/test_index.py:39 pins score >= 0)
            # (row blocks of ~2 MB and one reused buffer: the same subtraction and the same per-row summation as on the whole
            # chunk -- a row's sum does not depend on how many rows stand beside it -- without a chunk-sized temporary per query)
            step = max(1, (1 << 18) // max(1, xc.shape[1]))
            buf = np.empty((step, xc.shape[1]), dtype=np.float64)
            for b in range(0, xc.shape[0], step):
                xb = xc[b:b + step]
                diff = buf[:xb.shape[0]]
                for i in range(nq):
                    np.subtract(xb, q[i], out=diff)
                    out[i, s + b:s + b + xb.shape[0]] = np.einsum("ij,ij->i", diff, diff)
        else:
            dots = q @ xc.T
            if space == "cosine":
                xinv = 1.0 / (np.sqrt(np.einsum("ij,ij->i", xc, xc)) + 1e-30)
                out[:, s:s + xc.shape[0]] = 1.0 - dots * qinv[:, None] * xinv[None, :]
            else:
                out[:, s:s + xc.shape[0]] = 1.0 - dots
    return out


def _rank(dist_row: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """Indices ordering one query's candidates by (distance asc, label asc)."""
    return np.lexsort((labels, dist_row))


def knn(queries, rows, k: int, space: str, deleted=None):
    """Exact k nearest live rows for each query.

    Returns (labels int64 [nq, k], dist float32 [nq, k], counts int32 [nq]); rows that do
    not exist (fewer than k live rows) are padded with label -1 / distance +inf, which is
    how the C ABI pads (include/mlvdb_hip.h: mlvdb_search_batch).
    ``deleted`` is a boolean mask or an iterable of labels (hnswlib mark_deleted,
    reference index.py:80).
    """
    x = _as_f32_2d(rows) if len(rows) else np.zeros((0, _as_f32_2d(queries).shape[1]), np.float32)
    q = _as_f32_2d(queries)
    n = x.shape[0]
    live = np.ones(n, dtype=bool)
    if deleted is not None:
        deleted = np.asarray(deleted)
        if deleted.dtype == bool:
            live &= ~deleted
        elif deleted.size:
            live[deleted.astype(np.int64)] = False
    labels_live = np.nonzero(live)[0].astype(np.int64)
    nq = q.shape[0]
    out_l = np.full((nq, k), -1, dtype=np.int64)
    out_d = np.full((nq, k), np.inf, dtype=np.float32)
    out_c = np.zeros(nq, dtype=np.int32)
    if labels_live.size == 0 or k == 0:
        return out_l, out_d, out_c
    d = exact_distances(q, x[labels_live], space)
    kk = min(k, labels_live.size)
    for i in range(nq):
        order = _rank(d[i], labels_live)[:kk]
        out_l[i, :kk] = labels_live[order]
        out_d[i, :kk] = d[i, order].astype(np.float32)
        out_c[i] = kk
    return out_l, out_d, out_c


def range_query(queries, rows, radius: float, space: str, deleted=None):
    """All live rows with distance <= radius per query, nearest first, ties by label.

    No reference behaviour exists for range queries (README prose only:
    reference README.md:30-41); semantics are this repo's.  The comparison
    is made on the float64 distance against ``float64(float32(radius))``.
    Returns a list (one per query) of (labels int64, dist float32).
    """
    x = _as_f32_2d(rows)
    q = _as_f32_2d(queries)
    n = x.shape[0]
    live = np.ones(n, dtype=bool)
    if deleted is not None:
        deleted = np.asarray(deleted)
        if deleted.dtype == bool:
            live &= ~deleted
        elif deleted.size:
            live[deleted.astype(np.int64)] = False
    labels_live = np.nonzero(live)[0].astype(np.int64)
    r = float(np.float32(radius))
    res = []
    if labels_live.size == 0:
        return [(np.zeros(0, np.int64), np.zeros(0, np.float32)) for _ in range(q.shape[0])]
    d = exact_distances(q, x[labels_live], space)
    for i in range(q.shape[0]):
        sel = np.nonzero(d[i] <= r)[0]
        order = sel[_rank(d[i, sel], labels_live[sel])]
        res.append((labels_live[order], d[i, order].astype(np.float32)))
    return res


def postprocess_score(dist: float, metric: str) -> float:
    """``Index.search``'s score rule (reference index.py:125-127)."""
    score = float(dist)
    if metric == "cosine":
        score = 1 - score
    return score
