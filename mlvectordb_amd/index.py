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
from typing import Callable, Dict, Iterable, List, Mapping, Optional, Sequence
from uuid import UUID

import numpy as np

from .engine import HipScanEngine, ScanEngine
from .interfaces import VectorDTO, VectorProtocol

_SPACE_ALIASES = {"l2": "l2", "cosine": "cosine", "ip": "ip", "euclidean": "l2"}


@dataclass
class SearchResult:
    vector_id: UUID
    score: float


class _Namespace:
    """Host-side bookkeeping for one namespace (reference index.py:19-29, per key)."""

    __slots__ = ("engine", "dim", "uuid_to_label", "label_to_uuid", "total", "deleted", "rebuild_required")

    def __init__(self, engine: ScanEngine, dim: int) -> None:
        self.engine = engine
        self.dim = dim
        self.uuid_to_label: Dict[UUID, int] = {}
        self.label_to_uuid: Dict[int, UUID] = {}
        self.total = 0
        self.deleted = 0
        self.rebuild_required = False


EngineFactory = Callable[[int, str], ScanEngine]


class Index:
    def __init__(self, space: str = "l2", ef_construction: int = 200, M: int = 16,
                 rebuild_threshold: float = 0.2, *, device: int = 0, strategy: str = "auto",
                 engine_factory: Optional[EngineFactory] = None) -> None:
        # ef_construction / M are HNSW build knobs (index.py:18,37); an exhaustive scan has none.
        self._space = space
        self._ef_construction = ef_construction
        self._M = M
        self._rebuild_threshold = float(rebuild_threshold)
        self._device = device
        self._strategy = strategy
        self._engine_factory = engine_factory
        self._ns: Dict[str, _Namespace] = {}

    # ------------------------------------------------------------------ internals
    def _new_engine(self, dim: int, space: str) -> ScanEngine:
        native_space = _SPACE_ALIASES.get(space)
        if native_space is None:
            raise RuntimeError(f"Space name must be one of l2, ip, cosine or euclidean (got {space!r})")
        if self._engine_factory is not None:
            return self._engine_factory(dim, native_space)
        return HipScanEngine(dim, native_space, device=self._device, strategy=self._strategy)

    def _get_or_create(self, namespace: str, dim: int, space: str) -> _Namespace:
        ns = self._ns.get(namespace)
        if ns is None:
            ns = _Namespace(self._new_engine(dim, space), dim)
            self._ns[namespace] = ns
        return ns

    @staticmethod
    def _stack_rows(vectors: Sequence[VectorProtocol], dim: int) -> np.ndarray:
        rows = np.empty((len(vectors), dim), dtype=np.float32)
        for i, v in enumerate(vectors):
            vals = np.asarray(v.values, dtype=np.float32)
            if vals.shape != (dim,):
                raise RuntimeError(f"Wrong dimensionality of the vectors: row {i} has shape {vals.shape}, index dim {dim}")
            rows[i] = vals
        return rows

    def _append(self, ns: _Namespace, vectors: Sequence[VectorProtocol]) -> None:
        rows = self._stack_rows(vectors, ns.dim)
        first = ns.engine.append(rows)
        if first != ns.total:
            raise RuntimeError(f"engine label base {first} != host row count {ns.total}")
        for i, v in enumerate(vectors):
            ns.uuid_to_label[v.id] = first + i
            ns.label_to_uuid[first + i] = v.id
        ns.total += len(vectors)

    # ------------------------------------------------------------------ IndexProtocol
    def add(self, vectors: Iterable[VectorProtocol], namespace: str) -> None:
        """Append rows; labels continue from the namespace's row count (index.py:50-67)."""
        vectors = list(vectors)
        if not vectors:
            return
        dim = int(np.asarray(vectors[0].values).shape[0])
        ns = self._get_or_create(namespace, dim, self._space)
        self._append(ns, vectors)

    def remove(self, ids: Sequence[UUID], namespace: str) -> None:
        """Tombstone rows; raise the rebuild flag at deleted/total >= threshold (index.py:69-89)."""
        ns = self._ns.get(namespace)
        if ns is None:
            return
        labels = []
        for uid in ids:
            label = ns.uuid_to_label.pop(uid, None)
            if label is not None:
                ns.label_to_uuid.pop(label, None)
                labels.append(label)
        if labels:
            ns.engine.tombstone(np.asarray(labels, dtype=np.int64))
        ns.deleted += len(labels)
        if ns.deleted / max(1, ns.total) >= self._rebuild_threshold:
            ns.rebuild_required = True

    def search(self, query: VectorDTO, top_k: int, namespace: str, metric: str) -> List[SearchResult]:
        """Single-query kNN (index.py:91-129); one-row case of ``search_many``."""
        values = np.asarray(query.values, dtype=np.float32)
        if values.ndim != 1:
            return []
        return self.search_many(values[None, :], top_k, namespace, metric)[0]

    def rebuild(self, source: Mapping[str, Iterable[VectorProtocol]], metric: str) -> None:
        """Replace *every* namespace by ``source``, searching ``metric`` as the space (index.py:131-162)."""
        for ns in self._ns.values():
            ns.engine.close()
        self._ns.clear()
        for namespace, vectors in source.items():
            vectors = list(vectors)
            if not vectors:
                continue
            dim = int(np.asarray(vectors[0].values).shape[0])
            ns = self._get_or_create(namespace, dim, metric)
            self._append(ns, vectors)

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
        new_l2u: Dict[int, UUID] = {}
        for new, prev in enumerate(old.tolist()):
            uid = ns.label_to_uuid.get(prev)
            if uid is None:
                raise RuntimeError(f"compaction kept label {prev}, which the host maps do not know")
            new_l2u[new] = uid
        ns.label_to_uuid = new_l2u
        ns.uuid_to_label = {uid: label for label, uid in new_l2u.items()}
        ns.total = len(new_l2u)
        ns.deleted = 0
        ns.rebuild_required = False
        return True

    # ------------------------------------------------------------------ additive: batches and ranges
    def search_many(self, queries, top_k: int, namespace: str, metric: str,
                    allowed_ids: Optional[Iterable[UUID]] = None) -> List[List[SearchResult]]:
        """kNN for a batch of queries in one corpus scan.

        ``queries`` is an ``[nq, dim]`` array or a sequence of ``VectorDTO``.  Each entry of
        the result is what ``search`` would return for that query.  ``allowed_ids`` (additive: the
        row mask of a metadata-filtered search, README.md:121,130 intent) restricts the search to
        those vectors; the answer is the exact top-k among them.
        """
        q = self._coerce_queries(queries)
        nq = q.shape[0]
        ns = self._ns.get(namespace)
        if ns is None:
            return [[] for _ in range(nq)]
        active = ns.total - ns.deleted
        if active <= 0 or top_k <= 0 or nq == 0:
            return [[] for _ in range(nq)]
        if q.shape[1] != ns.dim:
            return [[] for _ in range(nq)]  # reference: RuntimeError swallowed at index.py:110-119
        mask = None
        if allowed_ids is not None:
            mask = np.zeros(ns.total, dtype=np.uint8)
            picked = [ns.uuid_to_label[u] for u in allowed_ids if u in ns.uuid_to_label]
            if not picked:
                return [[] for _ in range(nq)]
            mask[np.asarray(picked, dtype=np.int64)] = 1
            active = int(mask.sum())
        k = min(int(top_k), active)
        labels, dist, counts = self._search_engine(ns, q, k, mask)
        sqrt_score = metric == "euclidean"
        out: List[List[SearchResult]] = []
        for i in range(nq):
            hits: List[SearchResult] = []
            for label, d in zip(labels[i, :counts[i]].tolist(), dist[i, :counts[i]].tolist()):
                uid = ns.label_to_uuid.get(label)
                if uid is None:
                    continue
                score = float(d)
                if metric == "cosine":
                    score = 1 - score
                elif sqrt_score:
                    score = float(np.sqrt(max(score, 0.0)))
                hits.append(SearchResult(vector_id=uid, score=score))
            out.append(hits)
        return out

    def range_search(self, query: VectorDTO, radius: float, namespace: str, metric: str,
                     max_results: int = 1024) -> List[SearchResult]:
        values = np.asarray(query.values, dtype=np.float32)
        if values.ndim != 1:
            return []
        return self.range_search_many(values[None, :], radius, namespace, metric, max_results)[0]

    def range_search_many(self, queries, radius: float, namespace: str, metric: str,
                          max_results: int = 1024) -> List[List[SearchResult]]:
        """Every live row within ``radius`` of each query, nearest first (ties by insertion order).

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
        per_query = ns.engine.range(q, native_radius, max_results)
        out: List[List[SearchResult]] = []
        for labels, dist in per_query:
            hits = []
            for label, d in zip(labels.tolist(), dist.tolist()):
                uid = ns.label_to_uuid.get(label)
                if uid is None:
                    continue
                score = float(d)
                if metric == "cosine":
                    score = 1 - score
                elif metric == "euclidean":
                    score = float(np.sqrt(max(score, 0.0)))
                hits.append(SearchResult(vector_id=uid, score=score))
            out.append(hits)
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
            ids = np.zeros((ns.total, 16), dtype=np.uint8)
            for label, uid in ns.label_to_uuid.items():
                ids[label] = np.frombuffer(uid.bytes, dtype=np.uint8)
            ids.tofile(os.path.join(path, f"ns{i}.ids.u8"))
            live = np.zeros(ns.total, dtype=bool)
            live[list(ns.label_to_uuid.keys())] = True
            np.nonzero(~live)[0].astype(np.int64).tofile(os.path.join(path, f"ns{i}.deleted.i64"))
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
            ids = np.fromfile(os.path.join(path, f"ns{i}.ids.u8"), dtype=np.uint8).reshape(total, 16)
            dead = set(deleted.tolist())
            for label in range(total):
                if label not in dead:
                    uid = UUID(bytes=ids[label].tobytes())
                    ns.label_to_uuid[label] = uid
                    ns.uuid_to_label[uid] = label
            ns.total = total
            ns.deleted = int(m["deleted"])
            ns.rebuild_required = bool(m["rebuild_required"])
        return True

    def close(self) -> None:
        for ns in self._ns.values():
            ns.engine.close()
        self._ns.clear()
