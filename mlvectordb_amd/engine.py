"""Scan engines: what an ``Index`` namespace delegates its arithmetic to.

``HipScanEngine`` is the product backend -- a ctypes handle onto one ``mlvdb_index`` in
HBM (include/mlvdb_hip.h).  It takes the role hnswlib.Index plays in the reference
(src/mlvectordb/implementations/index.py:36-38,65,80,111).  There is deliberately no
CPU engine in this package: tests inject the oracle's engine through ``Index(engine_factory=...)``.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from typing import List, Protocol, Tuple

import numpy as np

from . import _native


class ScanEngine(Protocol):
    """One namespace's corpus: dense labels 0..total-1 in insertion order."""

    dim: int
    space: str

    def append(self, rows: np.ndarray) -> int: ...

    def tombstone(self, labels: np.ndarray) -> int: ...

    def counts(self) -> Tuple[int, int]: ...

    def compact(self) -> np.ndarray: ...

    def get_rows(self, first: int, n: int) -> np.ndarray: ...

    def get_rows_at(self, labels: np.ndarray) -> np.ndarray: ...

    def pair_distances(self, queries: np.ndarray, labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray]: ...

    def search(self, queries: np.ndarray, k: int, mask: np.ndarray | None = None
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]: ...

    def range(self, queries: np.ndarray, radius: float, capacity: int, truncate: bool = False
              ) -> List[Tuple[np.ndarray, np.ndarray]]: ...

    def close(self) -> None: ...


class RangeHits(Sequence):
    """The answer of a range call: ``hits[i]`` is ``(labels int64[n_i], distances float32[n_i])`` of query i, nearest first --
    views of the packed arrays ``labels`` / ``dist`` between ``offsets[i]`` and ``offsets[i + 1]`` (what the C ABI's packed
    entry returns; nothing is copied per query)."""

    __slots__ = ("labels", "dist", "offsets", "_off")

    def __init__(self, labels: np.ndarray, dist: np.ndarray, offsets: np.ndarray) -> None:
        self.labels, self.dist, self.offsets = labels, dist, offsets
        self._off = offsets.tolist()

    def __len__(self) -> int:
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("query index out of range")
        a, b = self._off[i], self._off[i + 1]
        return self.labels[a:b], self.dist[a:b]


class HipScanEngine:
    """Exhaustive fp32 corpus scan on one MI355X, through the C ABI."""

    def __init__(self, dim: int, space: str, device: int = 0, capacity_hint: int = 0,
                 strategy: str = "auto") -> None:
        if space not in _native.SPACE_CODES:
            # hnswlib raises RuntimeError("Space name must be one of l2, ip, or cosine.")
            raise RuntimeError(f"space must be one of l2, ip, cosine (got {space!r})")
        self._lib = _native.load()
        self.dim = int(dim)
        self.space = space
        self.device = int(device)
        handle = C.c_void_p()
        rc = self._lib.mlvdb_index_create(self.device, self.dim, _native.SPACE_CODES[space],
                                          int(capacity_hint), C.byref(handle))
        if rc != _native.OK:
            msg = self._lib.mlvdb_last_global_error().decode(errors="replace")
            raise RuntimeError(f"mlvdb_index_create failed ({rc}): {msg}")
        self._h = handle
        if strategy != "auto":
            self.set_strategy(strategy)

    # -- helpers -------------------------------------------------------------------
    def _check(self, rc: int, what: str, allow=()) -> int:
        if rc != _native.OK and rc not in allow:
            msg = self._lib.mlvdb_last_error(self._h).decode(errors="replace")
            raise RuntimeError(f"{what} failed ({rc}): {msg}")
        return rc

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def set_strategy(self, strategy: str) -> None:
        self._check(self._lib.mlvdb_index_set_strategy(self._h, _native.STRATEGY_CODES[strategy]), "set_strategy")

    def set_tuning(self, **knobs: int) -> None:
        """Tuning state of the handle (``mlvdb_index_set_tuning``): ``set_tuning(SCAN_VAR=237, I8=0)``.  The MLVDB_* environment
        variables are only read when the handle is created; this is how a live handle is switched (tools/scan_ab.py)."""
        for key, value in knobs.items():
            self._check(self._lib.mlvdb_index_set_tuning(self._h, f"{key}={int(value)}".encode()), f"set_tuning({key})")

    def get_tuning(self, key: str) -> int:
        out = C.c_int32(0)
        self._check(self._lib.mlvdb_index_get_tuning(self._h, key.encode(), C.byref(out)), f"get_tuning({key})")
        return int(out.value)

    def set_profiling(self, enabled: bool) -> None:
        self._check(self._lib.mlvdb_index_set_profiling(self._h, int(bool(enabled))), "set_profiling")

    def last_stats(self) -> dict:
        st = _native.Stats()
        self._check(self._lib.mlvdb_index_last_stats(self._h, C.byref(st)), "last_stats")
        return {name: getattr(st, name) for name, _ in _native.Stats._fields_}

    # -- ScanEngine ----------------------------------------------------------------
    def append(self, rows: np.ndarray) -> int:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {rows.shape}, index dim {self.dim}")
        first = C.c_int64(-1)
        self._check(self._lib.mlvdb_index_append(self._h, rows.ctypes.data, rows.shape[0], C.byref(first)), "append")
        return int(first.value)

    def append_device(self, device_ptr: int, n: int) -> int:
        first = C.c_int64(-1)
        self._check(self._lib.mlvdb_index_append_device(self._h, C.c_void_p(device_ptr), int(n), C.byref(first)),
                    "append_device")
        return int(first.value)

    def tombstone(self, labels: np.ndarray) -> int:
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        changed = C.c_int64(0)
        self._check(self._lib.mlvdb_index_tombstone(self._h, labels.ctypes.data, labels.size, C.byref(changed)),
                    "tombstone")
        return int(changed.value)

    def counts(self) -> Tuple[int, int]:
        total, deleted = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.mlvdb_index_counts(self._h, C.byref(total), C.byref(deleted)), "counts")
        return int(total.value), int(deleted.value)

    def compact(self) -> np.ndarray:
        """Drop the tombstoned rows on the device; returns old_labels with old_labels[new] = old."""
        total, deleted = self.counts()
        old = np.empty(max(total - deleted, 1), dtype=np.int64)
        live = C.c_int64(0)
        self._check(self._lib.mlvdb_index_compact(self._h, old.ctypes.data_as(C.c_void_p), old.size, C.byref(live)),
                    "mlvdb_index_compact")
        return old[: live.value]

    def reset(self, space: str | None = None) -> None:
        code = -1 if space is None else _native.SPACE_CODES[space]
        self._check(self._lib.mlvdb_index_reset(self._h, code), "reset")
        if space is not None:
            self.space = space

    def get_rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), dtype=np.float32)
        self._check(self._lib.mlvdb_index_get_rows(self._h, int(first), int(n), out.ctypes.data), "get_rows")
        return out

    def get_rows_at(self, labels: np.ndarray) -> np.ndarray:
        labels = np.ascontiguousarray(labels, dtype=np.int64).ravel()
        out = np.empty((labels.size, self.dim), dtype=np.float32)
        self._check(self._lib.mlvdb_index_get_rows_at(self._h, labels.ctypes.data, labels.size, out.ctypes.data),
                    "get_rows_at")
        return out

    def pair_distances(self, queries: np.ndarray, labels: np.ndarray):
        """Exact distances of the pairs (queries[q], row labels[q, j]) in this index's space, by the kernels' own fp64
        summation (``mlvdb_pair_distances``): (float64 [nq, m], float32 [nq, m]); label -1 gives +inf."""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {queries.shape}, index dim {self.dim}")
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        if labels.ndim != 2 or labels.shape[0] != queries.shape[0]:
            raise RuntimeError(f"labels must be [nq, m]; got {labels.shape} for {queries.shape[0]} queries")
        d64 = np.empty(labels.shape, dtype=np.float64)
        d32 = np.empty(labels.shape, dtype=np.float32)
        self._check(self._lib.mlvdb_pair_distances(self._h, queries.ctypes.data, queries.shape[0], labels.ctypes.data,
                                                   labels.shape[1], d64.ctypes.data, d32.ctypes.data), "pair_distances")
        return d64, d32

    def search64(self, queries: np.ndarray, k: int, mask: np.ndarray | None = None):
        """kNN; ``mask`` (optional, one byte per row, non-zero = allowed) restricts the search to those rows.
        Returns (labels int64, dist float32, counts int32, dist64 float64): the last is what a merge over several
        engines (row shards) has to rank on."""
        return self._search(queries, k, mask, True)

    def search(self, queries: np.ndarray, k: int, mask: np.ndarray | None = None):
        """kNN -> (labels int64 [nq, k], dist float32 [nq, k], counts int32 [nq]); ``mask`` as in ``search64``."""
        return self._search(queries, k, mask, False)

    def _search(self, queries: np.ndarray, k: int, mask, want64: bool):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {queries.shape}, index dim {self.dim}")
        nq = queries.shape[0]
        labels = np.empty((nq, k), dtype=np.int64)
        dist = np.empty((nq, k), dtype=np.float32)
        counts = np.empty(nq, dtype=np.int32)
        d64 = np.empty((nq, k), dtype=np.float64) if want64 else None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            if mask.shape != (self.counts()[0],):
                raise RuntimeError(f"row mask has shape {mask.shape}, the index holds {self.counts()[0]} rows")
        self._check(self._lib.mlvdb_search_batch_ex(self._h, queries.ctypes.data, nq, int(k),
                                                    None if mask is None else mask.ctypes.data, labels.ctypes.data,
                                                    dist.ctypes.data, counts.ctypes.data,
                                                    None if d64 is None else d64.ctypes.data), "search_batch")
        return (labels, dist, counts, d64) if want64 else (labels, dist, counts)

    def search_device(self, q_ptr: int, nq: int, k: int, labels_ptr: int, dist_ptr: int, counts_ptr: int,
                      dist64_ptr: int = 0, stream: int = 0) -> None:
        """Device-pointer search; results are complete when ``stream`` is.

        ``dist64_ptr`` (optional) receives the unrounded fp64 distances, which is what a
        multi-shard merge must rank on.
        """
        self._check(self._lib.mlvdb_search_batch_device(
            self._h, C.c_void_p(q_ptr), int(nq), int(k), C.c_void_p(labels_ptr), C.c_void_p(dist_ptr),
            C.c_void_p(counts_ptr), C.c_void_p(dist64_ptr or None), C.c_void_p(stream or None)), "search_batch_device")

    def range(self, queries: np.ndarray, radius: float, capacity: int, truncate: bool = False):
        """Per query (labels, fp32 distances) of the live rows within ``radius``, nearest first: a ``RangeHits`` sequence
        (``hits[i]`` -> the two arrays of query i, views of the call's packed outputs).

        ``truncate=False``: ``capacity`` is an initial size; the call is repeated once with the exact largest count,
        so every hit comes back (up to MLVDB_MAX_TOPK_PAGED = 16384 per query, the most one call can rank: beyond
        that the nearest 16384 are returned).  ``truncate=True``: at most ``capacity`` hits per query, the nearest.
        Through ``mlvdb_range_batch_packed``: hit counts differ by orders of magnitude between queries, the packed arrays
        hold the hits and nothing else (a first call with room for 256 hits per query on average; when the hits need
        more, the call is repeated with the size the first one reported)."""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {queries.shape}, index dim {self.dim}")
        nq = queries.shape[0]
        capacity = max(1, min(int(capacity), _native.MAX_TOPK_PAGED))
        total = min(nq * capacity, max(65_536, 256 * nq))
        while True:
            labels = np.empty(total, dtype=np.int64)
            dist = np.empty(total, dtype=np.float32)
            offsets = np.zeros(nq + 1, dtype=np.int64)
            counts = np.zeros(nq, dtype=np.int64)
            rc = self._check(self._lib.mlvdb_range_batch_packed(self._h, queries.ctypes.data, nq, float(radius), capacity, total,
                                                                labels.ctypes.data, dist.ctypes.data, offsets.ctypes.data,
                                                                counts.ctypes.data),
                             "range_batch_packed", allow=(_native.ERR_OVERFLOW,))
            if rc == _native.OK:
                break
            need_cap = capacity if truncate else min(max(int(counts.max(initial=0)), capacity), _native.MAX_TOPK_PAGED)
            need_total = int(np.minimum(counts, need_cap).sum())
            if need_cap == capacity and need_total <= total:
                break  # per-query truncation was asked for (or the engine's limit reached): the outputs hold the nearest
            capacity, total = need_cap, max(need_total, 1)
        return RangeHits(labels, dist, offsets)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.mlvdb_index_destroy(self._h)
            self._h = None

    def __del__(self) -> None:  # best effort; HBM is released with the handle
        try:
            self.close()
        except Exception:
            pass
