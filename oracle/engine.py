"""Oracle-backed ``ScanEngine`` (TEST INFRASTRUCTURE; see oracle/__init__.py).

Implements the engine interface of ``mlvectordb_amd.engine.ScanEngine`` on top of the NumPy
exact scan, so that the host logic of ``mlvectordb_amd.Index`` (UUID maps, clamping, score
flip, tombstone accounting) can be exercised on a machine without a GPU, and so the HIP
engine's answers can be compared call for call.  Injected by tests via
``Index(engine_factory=OracleScanEngine)``; never imported by the product package.
"""
from __future__ import annotations

import numpy as np

from . import exact_scan


class OracleScanEngine:
    def __init__(self, dim: int, space: str) -> None:
        if space not in exact_scan.SPACES:
            raise RuntimeError(f"space must be one of l2, ip, cosine (got {space!r})")
        self.dim = int(dim)
        self.space = space
        self._rows = np.zeros((0, self.dim), dtype=np.float32)
        self._deleted = np.zeros(0, dtype=bool)

    def append(self, rows: np.ndarray) -> int:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {rows.shape}, index dim {self.dim}")
        first = self._rows.shape[0]
        self._rows = np.concatenate([self._rows, rows], axis=0)
        self._deleted = np.concatenate([self._deleted, np.zeros(rows.shape[0], dtype=bool)])
        return first

    def tombstone(self, labels: np.ndarray) -> int:
        labels = np.asarray(labels, dtype=np.int64)
        labels = labels[(labels >= 0) & (labels < self._deleted.size)]
        fresh = np.unique(labels[~self._deleted[labels]])
        self._deleted[fresh] = True
        return int(fresh.size)

    def counts(self):
        return int(self._rows.shape[0]), int(self._deleted.sum())

    def get_rows(self, first: int, n: int) -> np.ndarray:
        if first < 0 or n < 0 or first + n > self._rows.shape[0]:
            raise RuntimeError("row range out of bounds")
        return self._rows[first:first + n].copy()

    def compact(self) -> np.ndarray:
        old = np.nonzero(~self._deleted)[0].astype(np.int64)
        self._rows = np.ascontiguousarray(self._rows[old])
        self._deleted = np.zeros(old.size, dtype=bool)
        return old

    def search(self, queries: np.ndarray, k: int, mask=None):
        deleted = self._deleted if mask is None else (self._deleted | (np.asarray(mask) == 0))
        return exact_scan.knn(queries, self._rows, k, self.space, deleted=deleted)

    def get_rows_at(self, labels: np.ndarray) -> np.ndarray:
        return self._rows[np.asarray(labels, dtype=np.int64).ravel()].copy()

    def pair_distances(self, queries: np.ndarray, labels: np.ndarray):
        labels = np.asarray(labels, dtype=np.int64)
        if labels.size and labels.max() >= self._rows.shape[0]:
            raise RuntimeError("label out of range")
        full = exact_scan.exact_distances(queries, self._rows, self.space)
        d64 = np.where(labels >= 0, np.take_along_axis(full, np.maximum(labels, 0), axis=1), np.inf)
        return d64, d64.astype(np.float32)

    def search64(self, queries: np.ndarray, k: int, mask=None):
        labels, dist, counts = self.search(queries, k, mask)
        d64 = np.full(labels.shape, np.inf)
        if self._rows.shape[0]:
            full = exact_scan.exact_distances(queries, self._rows, self.space)
            d64 = np.where(labels >= 0, np.take_along_axis(full, np.maximum(labels, 0), axis=1), np.inf)
        return labels, dist, counts, d64

    def range(self, queries: np.ndarray, radius: float, capacity: int, truncate: bool = False):
        hits = exact_scan.range_query(queries, self._rows, radius, self.space, deleted=self._deleted)
        return [(l[:capacity], d[:capacity]) for l, d in hits] if truncate else hits

    def close(self) -> None:
        self._rows = np.zeros((0, self.dim), dtype=np.float32)
        self._deleted = np.zeros(0, dtype=bool)
