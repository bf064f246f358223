"""CPU exact-scan baseline: fp32 GEMM (NumPy/OpenBLAS, all host cores) + argpartition top-k.

TEST/BENCH INFRASTRUCTURE (see oracle/__init__.py).  This is the "CPU exact-scan baseline (this
repo's NumPy backend)" of BASELINE.md section 2 -- NOT the reference: the reference's index is
hnswlib (approximate, 10k-row cap) and cannot run here.  It computes the same distances as
oracle/exact_scan.py but in float32 BLAS arithmetic, which is what a fast CPU implementation
would do; ids are ranked on those float32 scores, so it is a speed baseline, not the parity oracle.
"""
from __future__ import annotations

import numpy as np


class BlasScanIndex:
    """Pre-normalised (cosine) / pre-squared-norm (l2) corpus for repeated query waves."""

    def __init__(self, rows: np.ndarray, space: str) -> None:
        self.space = space
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if space == "cosine":
            nrm = np.sqrt(np.einsum("ij,ij->i", rows, rows, dtype=np.float64)).astype(np.float32)
            self.rows = rows / (nrm[:, None] + np.float32(1e-30))
            self.sq = None
        else:
            self.rows = rows
            self.sq = np.einsum("ij,ij->i", rows, rows) if space == "l2" else None

    def search(self, queries: np.ndarray, k: int):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if self.space == "cosine":
            q = q / (np.sqrt(np.einsum("ij,ij->i", q, q))[:, None] + np.float32(1e-30))
        dots = q @ self.rows.T  # [nq, n] fp32, OpenBLAS sgemm
        if self.space == "l2":
            dist = np.einsum("ij,ij->i", q, q)[:, None] - 2.0 * dots + self.sq[None, :]
        else:
            dist = 1.0 - dots
        n = dist.shape[1]
        kk = min(k, n)
        part = np.argpartition(dist, kk - 1, axis=1)[:, :kk]
        pd = np.take_along_axis(dist, part, axis=1)
        order = np.lexsort((part, pd), axis=1) if False else np.argsort(pd, axis=1, kind="stable")
        labels = np.take_along_axis(part, order, axis=1).astype(np.int64)
        return labels, np.take_along_axis(pd, order, axis=1).astype(np.float32)


class BlasScanEngine:
    """``BlasScanIndex`` behind the engine interface of ``mlvectordb_amd.Index`` (append / search / get_rows_at / counts),
    so that the CPU baseline is timed through the same ``QueryProcessor.find_similar_many -> Index.search_many`` surface as
    the GPU path -- UUID tables, score post-processing and enrichment included.  No tombstones, masks or ranges:
    a baseline, not an engine."""

    def __init__(self, dim: int, space: str) -> None:
        self.dim, self.space = int(dim), space
        self._chunks, self._index, self._n = [], None, 0

    def append(self, rows: np.ndarray) -> int:
        first = self._n
        self._chunks.append(np.ascontiguousarray(rows, dtype=np.float32))
        self._n += rows.shape[0]
        self._index = None
        return first

    def _built(self) -> BlasScanIndex:
        if self._index is None:
            self._rows = self._chunks[0] if len(self._chunks) == 1 else np.concatenate(self._chunks)
            self._chunks = [self._rows]
            self._index = BlasScanIndex(self._rows, self.space)
        return self._index

    def counts(self):
        return self._n, 0

    def search(self, queries: np.ndarray, k: int, mask=None):
        assert mask is None
        labels, dist = self._built().search(queries, k)
        return labels, dist, np.full(labels.shape[0], labels.shape[1], dtype=np.int32)

    def get_rows_at(self, labels: np.ndarray) -> np.ndarray:
        self._built()
        return self._rows[np.asarray(labels, dtype=np.int64).ravel()]

    def close(self) -> None:
        self._chunks, self._index, self._n = [], None, 0
