"""``Index``: MLVectorDB's per-namespace kNN index, backed by an exhaustive MI355X scan.

Drop-in for the reference's ``Index`` (src/mlvectordb/implementations/index.py:17-165): same
constructor, same four Protocol methods (interfaces/index.py:9-13), same ``is_rebuild_required``
and ``_space`` that ``QueryProcessor.delete`` reads (query_processor.py:58-61).  Where the
reference hands rows to an ``hnswlib.Index`` per namespace, this class hands them to a
``ScanEngine`` per namespace -- in production ``HipScanEngine`` (one ``mlvdb_index`` in HBM).
This side owns only what the reference's Python owns: UUID<->label maps, float32 coercion
(index.py:108), top-k clamping (:107), tombstone accounting (:84-89,103-105), and the
``1 - d`` flip (:126-127).

Behaviour kept from the reference: Q1/Q2 the ``metric`` argument of
``search`` never changes the space searched, it only flips the score; Q5 unknown / empty
namespace -> ``[]`` and top_k clamps to the live count; Q7 queries may be lists or float64.
Not kept: Q3 the 10,000-row cap (capacity is HBM-bound).  A query of the wrong
dimensionality returns ``[]`` exactly as the reference does (hnswlib's RuntimeError is
swallowed by index.py:110-119); rows of the wrong dimensionality raise ``RuntimeError``.

Additive (no reference counterpart): ``search_many`` (one scan for a whole query batch),
``range_search`` / ``range_search_many``, ``metric="euclidean"`` as sqrt(l2), ``compact`` and
``save_index`` / ``load_index`` (named in the reference's README.md:240-241 only).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from collections.abc import Sequence as SequenceABC
from typing import Callable, Dict, Iterable, List, Mapping, Optional, Sequence
from uuid import UUID

import numpy as np

from .engine import HipScanEngine, ScanEngine
from .idtable import IdTable, mint_uuid4_bytes
from .interfaces import VectorDTO, VectorProtocol

_SPACE_ALIASES = {"l2": "l2", "cosine": "cosine", "ip": "ip", "euclidean": "l2"}


@dataclass
class SearchResult:
    vector_id: UUID
    score: float


class _Namespace:
    """Host-side bookkeeping for one namespace (reference index.py:19-29, per key)."""

    __slots__ = ("engine", "dim", "ids", "total", "deleted", "rebuild_required")

    def __init__(self, engine: ScanEngine, dim: int) -> None:
        self.engine = engine
        self.dim = dim
        self.ids = IdTable()  # label <-> UUID (arrays, not dicts: idtable.py)
        self.total = 0
        self.deleted = 0
        self.rebuild_required = False


class BatchHits(SequenceABC):
    """What ``search_many`` returns: a sequence of ``nq`` hit lists (entry ``i`` is exactly what ``search`` returns for
    query ``i``) whose ``SearchResult`` objects are only built when an entry is read.  The arrays behind it are public,
    so a batched caller (``QueryProcessor.find_similar_many``) never pays a Python object per hit it does not look at:

    ``labels``  int64 [nq, k]   row labels, -1 = padding
    ``scores``  float64 [nq, k] the post-processed score (``1 - d`` for metric "cosine", index.py:125-127)
    ``counts``  int32 [nq]      valid prefix of each row
    ``ids()``   object [nq, k]  ``uuid.UUID`` per hit (``None`` at padding)
    """

    __slots__ = ("labels", "scores", "counts", "_table", "_ids", "_rows")

    def __init__(self, labels: np.ndarray, scores: np.ndarray, counts: np.ndarray, table: Optional[IdTable]) -> None:
        self.labels, self.scores, self.counts = labels, scores, counts
        self._table = table
        self._ids: Optional[np.ndarray] = None
        self._rows: Dict[int, List[SearchResult]] = {}

    @classmethod
    def empty(cls, nq: int) -> "BatchHits":
        return cls(np.full((nq, 0), -1, dtype=np.int64), np.zeros((nq, 0)), np.zeros(nq, dtype=np.int32), None)

    def valid(self) -> np.ndarray:
        """bool [nq, k]: which slots hold a hit."""
        return np.arange(self.labels.shape[1])[None, :] < self.counts[:, None]

    def id_bytes(self) -> np.ndarray:
        """uint8 [nq, k, 16]: the hits' UUID bytes (zeros at padding); no Python object per hit."""
        if self._table is None:
            return np.zeros(self.labels.shape + (16,), dtype=np.uint8)
        v = self.valid()
        out = self._table.raw[np.where(v, self.labels, 0)]
        out[~v] = 0
        return out

    def handles(self) -> np.ndarray:
        """int64 [nq, k]: the caller's per-row payload given to ``add_arrays`` (-1 = none / padding)."""
        if self._table is None:
            return np.full(self.labels.shape, -1, dtype=np.int64)
        v = self.valid()
        return np.where(v, self._table.handles[np.where(v, self.labels, 0)], -1)

    def ids(self) -> np.ndarray:
        if self._ids is None:
            valid = np.arange(self.labels.shape[1])[None, :] < self.counts[:, None]
            lab = np.where(valid, self.labels, -1)
            self._ids = (self._table.uuids_at(lab) if self._table is not None
                         else np.full(self.labels.shape, None, dtype=object))
        return self._ids

    def __len__(self) -> int:
        return int(self.labels.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        row = self._rows.get(i)
        if row is None:
            n = int(self.counts[i])
            row = [SearchResult(vector_id=u, score=s)
                   for u, s in zip(self.ids()[i, :n].tolist(), self.scores[i, :n].tolist()) if u is not None]
            self._rows[i] = row
        return row

    def __eq__(self, other) -> bool:
        if isinstance(other, (list, BatchHits)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented


EngineFactory = Callable[[int, str], ScanEngine]


class Index:
    def __init__(self, space: str = "l2", ef_construction: int = 200, M: int = 16,
                 rebuild_threshold: float = 0.2, *, device: int = 0, devices: Optional[Sequence[int]] = None,
                 strategy: str = "auto", capacity_hint: int = 0,
                 engine_factory: Optional[EngineFactory] = None) -> None:
        # ef_construction / M are HNSW build knobs (index.py:18,37); an exhaustive scan has none.
        self._space = space
        self._ef_construction = ef_construction
        self._M = M
        self._rebuild_threshold = float(rebuild_threshold)
        self._device = device
        # devices=[0, 1, ...]: every namespace is row-sharded over these GPUs inside this one process (SURVEY 8e,
        # multi_device.py); a device may be listed more than once (logical shards on one GPU)
        self._devices = None if devices is None else [int(x) for x in devices]
        self._strategy = strategy
        self._capacity_hint = int(capacity_hint)  # rows to reserve per namespace (and shard) up front: no regrowth copies
        self._engine_factory = engine_factory
        self._ns: Dict[str, _Namespace] = {}

    # ------------------------------------------------------------------ internals
    def _new_engine(self, dim: int, space: str) -> ScanEngine:
        native_space = _SPACE_ALIASES.get(space)
        if native_space is None:
            raise RuntimeError(f"Space name must be one of l2, ip, cosine or euclidean (got {space!r})")
        if self._devices is not None and len(self._devices) > 1:
            from .multi_device import MultiDeviceEngine

            factory = self._engine_factory
            per_shard = -(-self._capacity_hint // len(self._devices))
            return MultiDeviceEngine(dim, native_space, self._devices, strategy=self._strategy, capacity_hint=per_shard,
                                     shard_factory=None if factory is None else (lambda dev: factory(dim, native_space)))
        if self._engine_factory is not None:
            return self._engine_factory(dim, native_space)
        device = self._devices[0] if self._devices else self._device
        return HipScanEngine(dim, native_space, device=device, strategy=self._strategy,
                             capacity_hint=self._capacity_hint)

    def _get_or_create(self, namespace: str, dim: int, space: str) -> _Namespace:
        ns = self._ns.get(namespace)
        if ns is None:
            ns = _Namespace(self._new_engine(dim, space), dim)
            self._ns[namespace] = ns
        return ns

    @staticmethod
    def _stack_rows(vectors: Sequence[VectorProtocol], dim: int) -> np.ndarray:
        try:  # one C-level pass when every row has the right shape (the only case that succeeds)
            rows = np.asarray([v.values for v in vectors], dtype=np.float32)
            if rows.shape == (len(vectors), dim):
                return rows
        except ValueError:
            pass
        for i, v in enumerate(vectors):
            vals = np.asarray(v.values, dtype=np.float32)
            if vals.shape != (dim,):
                raise RuntimeError(f"Wrong dimensionality of the vectors: row {i} has shape {vals.shape}, index dim {dim}")
        raise RuntimeError(f"Wrong dimensionality of the vectors: expected [{len(vectors)}, {dim}]")

    @staticmethod
    def _check_finite(rows: np.ndarray) -> None:
        # A non-finite row would get a NaN norm, which the scan kernels read as "tombstoned": the row would vanish from
        # every search without being counted as deleted.  hnswlib accepts such rows and returns garbage; this index
        # refuses them.
        if not np.isfinite(rows).all():
            bad = int(np.flatnonzero(~np.isfinite(rows).all(axis=1))[0])
            raise RuntimeError(f"row {bad} of the batch holds a non-finite value (NaN / inf): not indexable")

    @staticmethod
    def _check_ids(vectors: Sequence[VectorProtocol]) -> List[UUID]:
        # the id table stores 16 bytes per row: anything else (SimpleVector("abc", ...)) is refused BEFORE the rows
        # reach the engine, so a bad batch leaves the namespace exactly as it was
        ids = [v.id for v in vectors]
        for i, u in enumerate(ids):
            if not isinstance(u, UUID):
                raise RuntimeError(f"row {i} of the batch has id {u!r}: this index keys rows by uuid.UUID")
        return ids

    def _stage(self, vectors: Sequence[VectorProtocol], dim: int):
        """Everything that can refuse a batch, before anything is mutated: (rows float32 [n, dim], ids)."""
        for i, v in enumerate(vectors):
            if getattr(v, "values", None) is None:
                raise RuntimeError(f"row {i} of the batch carries no values (a storage row whose values live in HBM only)")
        rows = self._stack_rows(vectors, dim)
        self._check_finite(rows)
        return rows, self._check_ids(vectors)

    def _append(self, ns: _Namespace, vectors: Sequence[VectorProtocol], staged=None) -> None:
        rows, ids = staged if staged is not None else self._stage(vectors, ns.dim)
        first = ns.engine.append(rows)
        if first != ns.total or ns.ids.append_uuids(ids) != first:
            raise RuntimeError(f"engine label base {first} != host row count {ns.total}")
        ns.total += len(ids)

    # ------------------------------------------------------------------ IndexProtocol
    def add(self, vectors: Iterable[VectorProtocol], namespace: str) -> None:
        """Append rows; labels continue from the namespace's row count (index.py:50-67)."""
        vectors = list(vectors)
        if not vectors:
            return
        known = self._ns.get(namespace)
        dim = known.dim if known is not None else int(np.asarray(vectors[0].values).shape[0])
        staged = self._stage(vectors, dim)  # a refused batch creates no namespace and appends nothing
        ns = self._get_or_create(namespace, dim, self._space)
        self._append(ns, vectors, staged)

    def validate_arrays(self, rows: np.ndarray, namespace: str, handles: Optional[np.ndarray] = None) -> None:
        """Raises what ``add_arrays`` would raise for this batch, without touching the index: a caller that writes to a
        second store first (``QueryProcessor.upsert_arrays``) checks here before it writes anything."""
        if rows.ndim != 2:
            raise RuntimeError(f"Wrong dimensionality of the vectors: expected a matrix, got shape {rows.shape}")
        ns = self._ns.get(namespace)
        if ns is not None and rows.shape[0] and rows.shape[1] != ns.dim:
            raise RuntimeError(f"Wrong dimensionality of the vectors: got {rows.shape}, index dim {ns.dim}")
        if handles is not None and np.asarray(handles).shape != (rows.shape[0],):
            raise RuntimeError(f"{rows.shape[0]} rows but handles of shape {np.asarray(handles).shape}")
        self._check_finite(rows)

    def add_arrays(self, rows: np.ndarray, namespace: str, ids: Optional[np.ndarray] = None,
                   handles: Optional[np.ndarray] = None) -> np.ndarray:
        """Additive bulk form of ``add``: ``rows`` is a float ``[n, dim]`` matrix, ``ids`` an optional ``[n, 16] uint8``
        table of UUID bytes (minted as uuid4 when omitted), ``handles`` an optional int64 payload per row that comes
        back with every hit (``BatchHits.handles``: a storage row number).  No Python object per row is created;
        returns the id table.  Same label rule as ``add`` (index.py:56-63)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2:
            raise RuntimeError(f"Wrong dimensionality of the vectors: expected a matrix, got shape {rows.shape}")
        n = rows.shape[0]
        ids = mint_uuid4_bytes(n) if ids is None else np.ascontiguousarray(ids, dtype=np.uint8).reshape(-1, 16)
        if ids.shape[0] != n:
            raise RuntimeError(f"{n} rows but {ids.shape[0]} ids")
        if n == 0:
            return ids
        self.validate_arrays(rows, namespace, handles)  # every refusal happens before the engine is touched
        ns = self._get_or_create(namespace, rows.shape[1], self._space)
        first = ns.engine.append(rows)
        if first != ns.total or ns.ids.append_raw(ids, handles) != first:
            raise RuntimeError(f"engine label base {first} != host row count {ns.total}")
        ns.total += n
        return ids

    def remove(self, ids: Sequence[UUID], namespace: str) -> None:
        """Tombstone rows; raise the rebuild flag at deleted/total >= threshold (index.py:69-89)."""
        ns = self._ns.get(namespace)
        if ns is None:
            return
        labels = ns.ids.lookup(ids)
        labels = np.unique(labels[labels >= 0])
        if labels.size:
            ns.engine.tombstone(labels)
            ns.ids.kill(labels)
        ns.deleted += int(labels.size)
        if ns.deleted / max(1, ns.total) >= self._rebuild_threshold:
            ns.rebuild_required = True

    def search(self, query: VectorDTO, top_k: int, namespace: str, metric: str) -> List[SearchResult]:
        """Single-query kNN (index.py:91-129); one-row case of ``search_many``."""
        values = np.asarray(query.values, dtype=np.float32)
        if values.ndim != 1:
            return []
        return self.search_many(values[None, :], top_k, namespace, metric)[0]

    def rebuild(self, source: Mapping[str, Iterable[VectorProtocol]], metric: str) -> None:
        """Replace *every* namespace by ``source``, searching ``metric`` as the space (index.py:131-162).

        The whole source is staged and validated on the host first (values present, one dimensionality per namespace,
        finite, UUID ids, a known space): a source that cannot be indexed raises and leaves the index as it was --
        closing the engines first would have destroyed the only copy of rows that live in HBM only."""
        if _SPACE_ALIASES.get(metric) is None:
            raise RuntimeError(f"Space name must be one of l2, ip, cosine or euclidean (got {metric!r})")
        staged = []
        for namespace, vectors in source.items():
            vectors = list(vectors)
            if not vectors:
                continue
            if getattr(vectors[0], "values", None) is None:
                raise RuntimeError(f"namespace {namespace!r}: the source rows carry no values")
            dim = int(np.asarray(vectors[0].values).shape[0])
            staged.append((namespace, dim, vectors, self._stage(vectors, dim)))
        for ns in self._ns.values():
            ns.engine.close()
        self._ns.clear()
        for namespace, dim, vectors, st in staged:
            ns = self._get_or_create(namespace, dim, metric)
            self._append(ns, vectors, st)

    def is_rebuild_required(self, namespace: str) -> bool:
        ns = self._ns.get(namespace)
        return bool(ns and ns.rebuild_required)

    def compact(self, namespace: str) -> bool:
        """Additive: what ``rebuild`` from this namespace's surviving vectors (in insertion order) leaves
        behind -- labels renumbered from 0, tombstones gone, flag cleared (index.py:145-162) -- computed on
        the device from the rows the index already owns, touching no other namespace (quirk Q4) and moving
        nothing over PCIe.  Returns False when the namespace is unknown."""
        ns = self._ns.get(namespace)
        if ns is None:
            return False
        old = ns.engine.compact()  # old[new label] = old label
        ns.ids = ns.ids.take(old)
        ns.total = ns.ids.n
        ns.deleted = 0
        ns.rebuild_required = False
        return True

    # ------------------------------------------------------------------ additive: batches and ranges
    @staticmethod
    def _scores(dist: np.ndarray, metric: str) -> np.ndarray:
        """``Index.search``'s score rule (index.py:125-127) on a whole array: the float32 distance as a Python float
        (= float64), ``1 - d`` for metric "cosine"; additive: the square root for "euclidean"."""
        score = dist.astype(np.float64)
        if metric == "cosine":
            score = 1 - score
        elif metric == "euclidean":
            score = np.sqrt(np.maximum(score, 0.0))
        return score

    def search_many(self, queries, top_k: int, namespace: str, metric: str,
                    allowed_ids: Optional[Iterable[UUID]] = None) -> BatchHits:
        """kNN for a batch of queries in one corpus scan.

        ``queries`` is an ``[nq, dim]`` array or a sequence of ``VectorDTO``.  Each entry of
        the result is what ``search`` would return for that query (``BatchHits``: a lazy sequence over the result
        arrays).  ``allowed_ids`` (additive: the row mask of a metadata-filtered search, README.md:121,130 intent)
        restricts the search to those vectors; the answer is the exact top-k among them.
        """
        q = self._coerce_queries(queries)
        nq = q.shape[0]
        ns = self._ns.get(namespace)
        if ns is None:
            return BatchHits.empty(nq)
        active = ns.total - ns.deleted
        if active <= 0 or top_k <= 0 or nq == 0:
            return BatchHits.empty(nq)
        if q.shape[1] != ns.dim:
            return BatchHits.empty(nq)  # reference: RuntimeError swallowed at index.py:110-119
        mask = None
        if allowed_ids is not None:
            picked = ns.ids.lookup(allowed_ids)
            picked = picked[picked >= 0]
            if not picked.size:
                return BatchHits.empty(nq)
            mask = np.zeros(ns.total, dtype=np.uint8)
            mask[picked] = 1
            active = int(mask.sum())
        k = min(int(top_k), active, self._MAX_TOP_K)  # the reference clamps to the live count (index.py:107)
        labels, dist, counts = self._search_engine(ns, q, k, mask)
        return BatchHits(labels, self._scores(dist, metric), counts, ns.ids)

    _MAX_TOP_K = 16384  # MLVDB_MAX_TOPK_PAGED: the most neighbours one call returns per query

    def search_stream(self, batches, top_k: int, namespace: str, metric: str):
        """Additive: ``search_many`` over an iterable of query batches, pipelined -- yields one ``BatchHits`` per batch, in
        order, while the next batch is already being scanned.  On a row-sharded namespace (``Index(devices=[...])``) the
        shard scans of wave i+1 are queued before wave i's per-shard candidates are merged
        (``MultiDeviceEngine.search_stream``); on a single engine a worker thread runs the next ``search_many`` (the ctypes
        call holds no GIL) while the caller consumes the current one.  The namespace must not be mutated meanwhile."""
        ns = self._ns.get(namespace)
        active = 0 if ns is None else ns.total - ns.deleted
        if ns is None or active <= 0 or top_k <= 0:
            for q in batches:
                yield BatchHits.empty(self._coerce_queries(q).shape[0])
            return
        k = min(int(top_k), active, self._MAX_TOP_K)
        stream = getattr(ns.engine, "search_stream", None)
        if stream is not None:
            ok = []  # per batch: the query count if it could not be scanned (wrong dimensionality), else None

            def feed():
                for q in batches:
                    q = self._coerce_queries(q)
                    if q.shape[1] != ns.dim or q.shape[0] == 0:
                        ok.append(q.shape[0])
                        continue
                    ok.append(None)
                    yield q

            results = stream(feed(), k)
            done = 0
            for labels, dist, counts, _ in results:
                while done < len(ok) and ok[done] is not None:  # batches skipped before this one
                    yield BatchHits.empty(ok[done])
                    done += 1
                done += 1
                yield BatchHits(labels, self._scores(dist, metric), counts, ns.ids)
            while done < len(ok):
                if ok[done] is not None:
                    yield BatchHits.empty(ok[done])
                done += 1
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            for q in batches:
                nxt = pool.submit(self.search_many, q, top_k, namespace, metric)
                if pending is not None:
                    yield pending.result()
                pending = nxt
            if pending is not None:
                yield pending.result()

    def range_search(self, query: VectorDTO, radius: float, namespace: str, metric: str,
                     max_results: int = 1024) -> List[SearchResult]:
        values = np.asarray(query.values, dtype=np.float32)
        if values.ndim != 1:
            return []
        return self.range_search_many(values[None, :], radius, namespace, metric, max_results)[0]

    def range_search_many(self, queries, radius: float, namespace: str, metric: str,
                          max_results: int = 1024) -> List[List[SearchResult]]:
        """The live rows within ``radius`` of each query, nearest first (ties by insertion order), at most
        ``max_results`` per query (the nearest ones; ``None`` = all, up to the engine's 16384 per query).

        ``radius`` is a distance in the namespace's space (squared for l2; plain for
        ``metric="euclidean"``); scores are post-processed exactly like ``search``.
        No reference implementation exists for range queries (README prose only).
        """
        q = self._coerce_queries(queries)
        nq = q.shape[0]
        ns = self._ns.get(namespace)
        if ns is None or ns.total - ns.deleted <= 0 or nq == 0 or q.shape[1] != ns.dim:
            return [[] for _ in range(nq)]
        native_radius = float(radius) ** 2 if metric == "euclidean" else float(radius)
        cap = self._MAX_TOP_K if max_results is None else max(1, int(max_results))
        per_query = ns.engine.range(q, native_radius, cap, truncate=True)
        out: List[List[SearchResult]] = []
        for labels, dist in per_query:
            uids = ns.ids.uuids_at(labels).tolist()
            scores = self._scores(dist, metric).tolist()
            out.append([SearchResult(vector_id=u, score=s) for u, s in zip(uids, scores) if u is not None])
        return out

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _coerce_queries(queries) -> np.ndarray:
        if isinstance(queries, np.ndarray):
            q = queries.astype(np.float32, copy=False)
        else:
            q = np.array([np.asarray(getattr(v, "values", v), dtype=np.float32) for v in queries], dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :] if q.size else q.reshape(0, 0)
        return np.ascontiguousarray(q)

    @staticmethod
    def _search_engine(ns: _Namespace, q: np.ndarray, k: int, mask=None):
        """Engines select at most ``max_topk`` per scan; larger k is served in rank-ordered pages."""
        return ns.engine.search(q, k) if mask is None else ns.engine.search(q, k, mask)

    def namespace_counts(self, namespace: str):
        ns = self._ns.get(namespace)
        return (0, 0) if ns is None else (ns.total, ns.deleted)

    def fetch_values(self, namespace: str, labels: np.ndarray) -> np.ndarray:
        """Additive: the stored float32 rows of ``labels`` read back from the index's own copy in HBM (bit-exact), so a
        caller that keeps no second copy of the corpus on the host can still enrich hits with ``values``."""
        ns = self._ns[namespace]
        return ns.engine.get_rows_at(np.asarray(labels, dtype=np.int64))

    def distances(self, queries, ids, namespace: str, metric: str) -> np.ndarray:
        """Additive (SURVEY 8a row a8': the vector-level ``distance()`` / ``similarity()`` of reference README.md:30-41,
        178-181, on the device): the score ``search`` reports for each pair (queries[q], stored vector ids[q][j]) --
        float64 array ``[nq, m]`` of the float32-rounded distance in the namespace's space, post-processed like
        ``search`` (``1 - d`` for metric "cosine", sqrt for "euclidean").  ``ids`` is ``[nq][m]`` UUIDs or an int64 label
        array; unknown / removed ids give NaN.  Computed by the kernels' own fp64 summation over the rows in HBM, so
        ``distances(q, [hit.vector_id])`` equals that hit's score exactly."""
        q = self._coerce_queries(queries)
        ns = self._ns.get(namespace)
        if ns is None or q.shape[1] != ns.dim:
            raise RuntimeError(f"namespace {namespace!r} is unknown or its dimensionality differs from the queries'")
        if isinstance(ids, np.ndarray) and ids.dtype.kind in "iu":
            labels = np.asarray(ids, dtype=np.int64).reshape(q.shape[0], -1).copy()
            inside = (labels >= 0) & (labels < ns.ids.n)
            dead = np.zeros(labels.shape, dtype=bool)
            dead[inside] = ~ns.ids.live[labels[inside]]
            labels[dead | ~inside] = -1  # removed or unknown labels read as NaN, like removed ids (the engine itself would score them)
        else:
            rows = [list(r) for r in ids]
            m = len(rows[0]) if rows else 0
            if any(len(r) != m for r in rows) or len(rows) != q.shape[0]:
                raise RuntimeError("ids must hold the same number of ids for every query")
            labels = ns.ids.lookup([u for r in rows for u in r]).reshape(q.shape[0], m)
        known = labels >= 0
        _, d32 = ns.engine.pair_distances(q, np.where(known, labels, -1))
        out = self._scores(d32, metric)
        out[~known] = np.nan
        return out

    def fetch_values_by_id(self, namespace: str, ids: Sequence[UUID]) -> np.ndarray:
        """``fetch_values`` addressed by id (every id must be live in the namespace)."""
        ns = self._ns[namespace]
        labels = ns.ids.lookup(ids)
        if (labels < 0).any():
            raise RuntimeError("fetch_values_by_id: unknown or removed id")
        return ns.engine.get_rows_at(labels)

    # ------------------------------------------------------------------ additive: persistence
    # Directory layout ("mlvdb-index-v1"): index.json + per namespace i
    #   ns<i>.rows.f32     raw row-major float32 [total, dim], every label incl. tombstoned ones (labels stay stable)
    #   ns<i>.ids.u8       [total, 16] UUID bytes, all-zero for tombstoned labels
    # The fp32 rows are the ones the device holds (bit-exact round trip); norms, bf16 shadow and panel layout are
    # rebuilt by the ingest kernels at load.  No reference behaviour (README.md:240-241 names the two methods only).
    _FORMAT = "mlvdb-index-v1"
    _CHUNK_BYTES = 256 << 20

    def save_index(self, path: str) -> bool:
        """Write every namespace to directory ``path`` (created if missing); streams the rows off the device in
        chunks, so host memory stays bounded."""
        os.makedirs(path, exist_ok=True)
        meta = {"format": self._FORMAT, "space": self._space, "rebuild_threshold": self._rebuild_threshold, "namespaces": []}
        for i, (name, ns) in enumerate(self._ns.items()):
            chunk = max(1, self._CHUNK_BYTES // (4 * ns.dim))
            with open(os.path.join(path, f"ns{i}.rows.f32"), "wb") as f:
                for first in range(0, ns.total, chunk):
                    f.write(ns.engine.get_rows(first, min(chunk, ns.total - first)).tobytes())
            dead = ns.ids.dead_labels()
            ids = ns.ids.raw[:ns.total].copy()
            ids[dead] = 0
            ids.tofile(os.path.join(path, f"ns{i}.ids.u8"))
            dead.tofile(os.path.join(path, f"ns{i}.deleted.i64"))
            meta["namespaces"].append({"name": name, "dim": ns.dim, "space": ns.engine.space, "total": ns.total,
                                       "deleted": ns.deleted, "rebuild_required": ns.rebuild_required})
        tmp = os.path.join(path, "index.json.tmp")
        with open(tmp, "w") as f:
            json.dump(meta, f, indent=1)
        os.replace(tmp, os.path.join(path, "index.json"))  # the manifest appears last and atomically
        return True

    def load_index(self, path: str) -> bool:
        """Replace the contents of this index by the directory written by ``save_index``.  Returns False (index
        untouched) when ``path`` holds no manifest; raises ``RuntimeError`` on a manifest it cannot honour."""
        manifest = os.path.join(path, "index.json")
        if not os.path.isfile(manifest):
            return False
        with open(manifest) as f:
            meta = json.load(f)
        if meta.get("format") != self._FORMAT:
            raise RuntimeError(f"unknown index format {meta.get('format')!r}")
        for i, m in enumerate(meta["namespaces"]):  # validate sizes before touching anything
            total, dim = int(m["total"]), int(m["dim"])
            if os.path.getsize(os.path.join(path, f"ns{i}.rows.f32")) != total * dim * 4 or \
                    os.path.getsize(os.path.join(path, f"ns{i}.ids.u8")) != total * 16:
                raise RuntimeError(f"namespace {m['name']!r}: file sizes do not match the manifest")
        self.close()
        self._space = meta["space"]
        self._rebuild_threshold = float(meta["rebuild_threshold"])
        for i, m in enumerate(meta["namespaces"]):
            total, dim = int(m["total"]), int(m["dim"])
            ns = self._get_or_create(m["name"], dim, m["space"])
            if total:
                rows = np.memmap(os.path.join(path, f"ns{i}.rows.f32"), dtype=np.float32, mode="r", shape=(total, dim))
                chunk = max(1, self._CHUNK_BYTES // (4 * dim))
                for first in range(0, total, chunk):
                    if ns.engine.append(np.ascontiguousarray(rows[first:first + chunk])) != first:
                        raise RuntimeError("engine label base does not match the file offset")
                del rows
            deleted = np.fromfile(os.path.join(path, f"ns{i}.deleted.i64"), dtype=np.int64)
            if deleted.size:
                ns.engine.tombstone(deleted)
            ns.ids.append_raw(np.fromfile(os.path.join(path, f"ns{i}.ids.u8"), dtype=np.uint8).reshape(total, 16))
            ns.ids.kill(deleted)
            ns.total = total
            ns.deleted = int(m["deleted"])
            ns.rebuild_required = bool(m["rebuild_required"])
        return True

    def close(self) -> None:
        for ns in self._ns.values():
            ns.engine.close()
        self._ns.clear()
