"""``hnswlib``-shaped facade over ``libmlvdb_hip.so``: the reference-side binding of INTEGRATION.md section 2, shipped.

The reference reaches its arithmetic through exactly eight hnswlib calls
(src/mlvectordb/implementations/index.py): the constructor ``hnswlib.Index(space=, dim=)`` (:36),
``init_index(max_elements=, ef_construction=, M=)`` (:37), ``set_ef`` (:38), ``get_current_count`` (:56),
``add_items(data, labels)`` (:65,158), ``mark_deleted(label)`` (:80) and ``knn_query(data, k=)`` (:111,115).  A
maintainer swaps the scan in with one line -- ``import mlvectordb_amd.hnswlib_compat as hnswlib`` at index.py:1 -- and
nothing else in the reference changes.  This module is that binding: same names, argument meaning and error
behaviour (``RuntimeError`` with hnswlib's messages for an unknown space, a wrong dimensionality, and a ``knn_query``
that cannot fill ``k``), on the C ABI through ``HipScanEngine``.  No CPU fallback: ``init_index`` fails loudly
without the library or a GPU.

Differences a caller can observe: results are exact (the recall-1.0 limit of the graph walk), ``max_elements`` is
not a cap, ``ef`` / ``M`` / ``ef_construction`` are accepted and ignored, labels must be the dense insertion-order
integers the reference always passes (index.py:56-63) -- anything else raises.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .engine import HipScanEngine

_SPACES = ("l2", "cosine", "ip")


class Index:
    def __init__(self, space: str, dim: int, *, device: int = 0) -> None:
        if space not in _SPACES:
            raise RuntimeError("Space name must be one of l2, ip, or cosine.")
        self.space = space
        self.dim = int(dim)
        self._device = device
        self._engine: Optional[HipScanEngine] = None
        self.max_elements = 0
        self.ef = 10

    # ---- index.py:37
    def init_index(self, max_elements: int, ef_construction: int = 200, M: int = 16, random_seed: int = 100,
                   allow_replace_deleted: bool = False) -> None:
        if self._engine is not None:
            raise RuntimeError("The index is already initiated.")
        self._engine = HipScanEngine(self.dim, self.space, device=self._device)
        self.max_elements = int(max_elements)  # kept for introspection; capacity is bounded by HBM, not by this

    def _need(self) -> HipScanEngine:
        if self._engine is None:
            raise RuntimeError("Index not initialized: call init_index first")
        return self._engine

    # ---- index.py:38: the scan is exhaustive, there is no search-time accuracy knob
    def set_ef(self, ef: int) -> None:
        self.ef = int(ef)

    # ---- index.py:56
    def get_current_count(self) -> int:
        return self._need().counts()[0]

    def get_max_elements(self) -> int:
        return self.max_elements

    # ---- index.py:65,158
    def add_items(self, data, ids=None, num_threads: int = -1, replace_deleted: bool = False) -> None:
        eng = self._need()
        data = np.asarray(data, dtype=np.float32)
        if data.ndim == 1:
            data = data[None, :]
        if data.ndim != 2 or data.shape[1] != self.dim:
            raise RuntimeError("Wrong dimensionality of the vectors")
        first = eng.counts()[0]
        if ids is not None:
            ids = np.asarray(ids, dtype=np.int64).ravel()
            if ids.size != data.shape[0] or not np.array_equal(ids, np.arange(first, first + data.shape[0])):
                raise RuntimeError("labels must be the dense insertion-order integers "
                                   f"{first}..{first + data.shape[0] - 1} (what index.py:56-63 passes)")
        if eng.append(np.ascontiguousarray(data)) != first:
            raise RuntimeError("label base mismatch")

    # ---- index.py:80
    def mark_deleted(self, label: int) -> None:
        eng = self._need()
        label = int(label)
        if not 0 <= label < eng.counts()[0]:
            raise RuntimeError("Label not found")
        if eng.tombstone(np.array([label], dtype=np.int64)) == 0:
            raise RuntimeError("The requested to delete element is already deleted")

    # ---- index.py:111,115
    def knn_query(self, data, k: int = 1, num_threads: int = -1, filter=None):  # noqa: A002 (hnswlib's name)
        eng = self._need()
        q = np.asarray(data, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise RuntimeError("Wrong dimensionality of the vectors")
        total, deleted = eng.counts()
        if k < 1 or k > total - deleted:
            # hnswlib's message when the graph walk cannot fill k results; index.py:112-119 catches it and retries k=1
            raise RuntimeError("Cannot return the results in a contiguous 2D array. Probably ef or M is too small")
        labels, dist, counts = eng.search(np.ascontiguousarray(q), int(k))
        if (counts < k).any():
            raise RuntimeError("Cannot return the results in a contiguous 2D array. Probably ef or M is too small")
        return labels.astype(np.uint64), dist

    def close(self) -> None:
        if self._engine is not None:
            self._engine.close()
            self._engine = None
