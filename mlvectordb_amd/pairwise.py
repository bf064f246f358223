"""Pairwise ``distance`` / ``similarity`` / ``normalize`` of two stored vectors (SURVEY.md section 8a, row a8').

The reference names these methods only in prose (``SimpleVector.distance(other, metric="euclidean")`` /
``.similarity(other, metric="cosine")`` / ``.normalize()``: reference README.md:30-41,178-181); no code
exists for them, so the definitions are the ones SURVEY.md section 8(a) fixed, chosen to agree with what
``Index.search`` returns for the same pair of vectors:

    distance   "l2"         sum_i (a_i - b_i)^2                      (squared: hnswlib's l2 space, index.py:36)
               "euclidean"  sqrt of that                              (README.md:37)
               "cosine"     1 - <a,b> / ((|a| + 1e-30)(|b| + 1e-30))  (hnswlib's normalisation)
               "ip"         1 - <a,b>
    similarity "cosine"     <a,b> / ((|a| + 1e-30)(|b| + 1e-30))      (= Index.search's cosine score, index.py:126-127)
               "ip"/"dot"   <a,b>
               "l2"/"euclidean"  the negated distance                 (larger = more similar)
    normalize  x / (|x| + 1e-30), float32

Arithmetic: float64 from the float32 values (the products of two float32 are exact in float64), which is the
canonical arithmetic of the scan kernels' rescoring step and of oracle/exact_scan.py -- a pair scored here and
the same pair scored by ``Index.search`` differ only by float64 summation order (~1e-15 relative).  Two vectors are host work (2d flops);
no kernel is involved.  Parity unpinned: no reference implementation or test exists.
"""
from __future__ import annotations

import numpy as np

DISTANCE_METRICS = ("l2", "euclidean", "cosine", "ip")
SIMILARITY_METRICS = ("cosine", "ip", "dot", "l2", "euclidean")


def _pair(a, b):
    a = np.asarray(a, dtype=np.float32).astype(np.float64).ravel()
    b = np.asarray(b, dtype=np.float32).astype(np.float64).ravel()
    if a.shape != b.shape:
        raise ValueError(f"vectors have different shapes: {a.shape} and {b.shape}")
    return a, b


def _cosine(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.dot(a, b) * (1.0 / (np.sqrt(np.dot(a, a)) + 1e-30)) * (1.0 / (np.sqrt(np.dot(b, b)) + 1e-30)))


def distance(a, b, metric: str = "euclidean") -> float:
    a, b = _pair(a, b)
    if metric == "l2" or metric == "euclidean":
        diff = a - b
        d = float(np.dot(diff, diff))
        return float(np.sqrt(d)) if metric == "euclidean" else d
    if metric == "cosine":
        return 1.0 - _cosine(a, b)
    if metric == "ip":
        return 1.0 - float(np.dot(a, b))
    raise ValueError(f"unknown distance metric {metric!r}: one of {DISTANCE_METRICS}")


def similarity(a, b, metric: str = "cosine") -> float:
    if metric == "cosine":
        return _cosine(*_pair(a, b))
    if metric in ("ip", "dot"):
        a, b = _pair(a, b)
        return float(np.dot(a, b))
    if metric in ("l2", "euclidean"):
        return -distance(a, b, metric)
    raise ValueError(f"unknown similarity metric {metric!r}: one of {SIMILARITY_METRICS}")


def normalize(values) -> np.ndarray:
    v = np.asarray(values, dtype=np.float32)
    v64 = v.astype(np.float64)
    return (v64 * (1.0 / (np.sqrt(np.dot(v64.ravel(), v64.ravel())) + 1e-30))).astype(np.float32)
