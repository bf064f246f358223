"""``QueryProcessor``: storage + index orchestration around the search hot path.

Counterpart of the reference's ``QueryProcessor``
(src/mlvectordb/implementations/query_processor.py:11-82).  ``find_similar`` is the
dispatch the north star names: ``Index.search`` then enrichment of each hit with the stored
values/metadata, in index order, silently dropping ids the storage no longer has
(query_processor.py:33-49).  ``find_similar_many`` is the additive batched sibling: one
corpus scan for the whole query batch.

Deliberate divergence: the reference's ``delete`` rebuilds the index
from *only* the affected namespace (query_processor.py:60), and ``Index.rebuild`` clears
every namespace first (index.py:136-143), so other namespaces silently vanish from the
index.  Here the rebuild source is the whole storage map; pass ``rebuild_scope="namespace"``
to reproduce the reference byte for byte.
"""
from __future__ import annotations

import numpy as np
from typing import Any, Dict, Iterable, List, Sequence
from uuid import UUID

from .interfaces import IndexProtocol, VectorDTO
from .vector import Vector


class QueryProcessor:
    def __init__(self, storage_engine, index: IndexProtocol, *, rebuild_scope: str = "all") -> None:
        if rebuild_scope not in ("all", "namespace"):
            raise ValueError("rebuild_scope must be 'all' or 'namespace'")
        self._storage = storage_engine
        self._index = index
        self._rebuild_scope = rebuild_scope

    # ---- writes (query_processor.py:16-24): every call mints new ids, nothing is updated in place
    def insert(self, vector: VectorDTO, namespace: str = "default") -> None:
        row = Vector(values=vector.values, metadata=vector.metadata)
        self._storage.write(row, namespace)
        self._index.add([row], namespace)

    def upsert_many(self, vectors: Iterable[VectorDTO], namespace: str = "default") -> None:
        rows = [Vector(values=v.values, metadata=v.metadata) for v in vectors]
        self._storage.write_vectors(rows, namespace)
        self._index.add(rows, namespace)

    # ---- the hot path
    def _enrich(self, hits, namespace: str) -> List[dict]:
        if not hits:
            return []
        stored = {v.id: v for v in self._storage.read_vectors([h.vector_id for h in hits], namespace) if v}
        out = []
        for h in hits:
            v = stored.get(h.vector_id)
            if v:
                out.append({"id": v.id, "values": v.values, "metadata": v.metadata, "score": h.score})
        return out

    def find_similar(self, query: VectorDTO, top_k: int, namespace: str = "default",
                     metric: str = "cosine") -> List[dict]:
        if hasattr(self._storage, "read_rows_raw") and hasattr(self._index, "search_many"):
            values = np.asarray(query.values, dtype=np.float32)  # array-backed storage: the one-row case of the batch path
            if values.ndim != 1:
                return []
            return self._enrich_many(self._index.search_many(values[None, :], top_k, namespace, metric), namespace)[0]
        hits = self._index.search(query, top_k=top_k, namespace=namespace, metric=metric)
        return self._enrich(hits, namespace)

    def find_similar_many(self, queries, top_k: int, namespace: str = "default",
                          metric: str = "cosine", where=None) -> List[List[dict]]:
        """Batched ``find_similar``: ``queries`` is an [nq, dim] array or a sequence of VectorDTO.

        ``where`` (additive; README.md:121,130,252,274 intent, no reference code): a predicate over a stored
        vector's metadata dict.  It is evaluated once over the namespace's stored vectors and handed to the
        index as a row mask, so the answer is the exact top-k among the matching vectors (not a post-filter of
        an unrestricted top-k)."""
        return self._enrich_many(self._search_many(queries, top_k, namespace, metric, where), namespace)

    def _search_many(self, queries, top_k: int, namespace: str, metric: str, where):
        if where is None:
            return self._index.search_many(queries, top_k=top_k, namespace=namespace, metric=metric)
        allowed = [v.id for v in self._storage.namespace_map.get(namespace, []) if where(v.metadata)]
        return self._index.search_many(queries, top_k=top_k, namespace=namespace, metric=metric, allowed_ids=allowed)

    def _enrich_many(self, per_query, namespace: str) -> List[List[dict]]:
        """``_enrich`` for a whole batch.  With an array-backed storage and this package's ``Index`` the 2,560 hits of a
        256-query wave are resolved by array operations -- id bytes -> storage rows by one sorted lookup, values by one
        gather (from the storage's matrix, or from the index's rows in HBM) -- and Python only builds the result
        dicts; any other storage / index goes hit list by hit list through ``_enrich``."""
        fast = getattr(self._storage, "read_rows_raw", None)
        if fast is None or not hasattr(per_query, "id_bytes"):
            return [self._enrich(hits, namespace) for hits in per_query]
        valid = per_query.valid()
        counts = valid.sum(axis=1).tolist()
        if not valid.any():
            return [[] for _ in counts]
        handles = per_query.handles()[valid] if hasattr(self._storage, "read_rows_at") else None
        if handles is not None and (handles >= 0).all():
            found, values, metas = self._storage.read_rows_at(handles, namespace)  # storage row numbers ride with the hits
        else:
            found, values, metas = fast(per_query.id_bytes()[valid], namespace)
        if values is None:
            values = self._index.fetch_values(namespace, per_query.labels[valid])
        ids = per_query.ids()[valid].tolist()
        scores = per_query.scores[valid].tolist()
        rows = list(values)  # one ndarray view per hit
        out, pos = [], 0
        if found.all():
            for n in counts:
                out.append([{"id": ids[j], "values": rows[j], "metadata": metas[j], "score": scores[j]}
                            for j in range(pos, pos + n)])
                pos += n
        else:
            ok = found.tolist()
            for n in counts:
                out.append([{"id": ids[j], "values": rows[j], "metadata": metas[j], "score": scores[j]}
                            for j in range(pos, pos + n) if ok[j]])
                pos += n
        return out

    def find_similar_stream(self, batches, top_k: int, namespace: str = "default", metric: str = "cosine"):
        """Additive: ``find_similar_many`` over an iterable of query batches, pipelined -- while the GPU scans batch
        i+1 (a worker thread inside the ctypes call, which holds no GIL) this thread enriches batch i.  Yields one
        ``List[List[dict]]`` per batch, in order."""
        stream = getattr(self._index, "search_stream", None)
        if stream is not None:  # this package's Index: one engine -> a prefetching worker; row shards -> scan / merge pipeline
            for hits in stream(batches, top_k, namespace, metric):
                yield self._enrich_many(hits, namespace)
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            for q in batches:
                nxt = pool.submit(self._search_many, q, top_k, namespace, metric, None)
                if pending is not None:
                    yield self._enrich_many(pending.result(), namespace)
                pending = nxt
            if pending is not None:
                yield self._enrich_many(pending.result(), namespace)

    def find_similar_where(self, query: VectorDTO, top_k: int, where, namespace: str = "default",
                           metric: str = "cosine") -> List[dict]:
        """``find_similar`` restricted to the vectors whose metadata satisfies ``where``."""
        values = np.asarray(query.values, dtype=np.float32)
        if values.ndim != 1:
            return []
        return self.find_similar_many(values[None, :], top_k, namespace, metric, where=where)[0]

    def find_in_radius(self, query: VectorDTO, radius: float, namespace: str = "default",
                       metric: str = "cosine", max_results: int = 1024) -> List[dict]:
        """Range query (no reference counterpart; README.md:30-41 intent only)."""
        hits = self._index.range_search(query, radius, namespace=namespace, metric=metric, max_results=max_results)
        out = self._enrich(hits, namespace)
        missing = [i for i, h in enumerate(out) if h["values"] is None]  # array storage that keeps the rows in HBM only
        if missing and hasattr(self._index, "fetch_values_by_id"):
            rows = self._index.fetch_values_by_id(namespace, [out[i]["id"] for i in missing])
            for i, r in zip(missing, rows):
                out[i]["values"] = r
        return out

    # ---- delete -> lazy rebuild (query_processor.py:51-62)
    def delete(self, ids: Sequence[UUID], namespace: str = "default") -> Sequence[UUID]:
        removed = [vid for vid in ids if self._storage.delete(vid, namespace)]
        self._index.remove(ids, namespace)
        probe = getattr(self._index, "is_rebuild_required", None)
        if probe and probe(namespace):
            compact = getattr(self._index, "compact", None)
            if compact and self._rebuild_scope != "namespace" and compact(namespace):
                return removed  # same end state as the rebuild below, computed on the device
            full = self._storage.namespace_map
            if self._rebuild_scope == "namespace":
                source = {namespace: full.get(namespace, [])}
            else:
                source = dict(full)
                source.setdefault(namespace, [])
            self._index.rebuild(self._with_values(source), metric=self._index._space)
        return removed

    def _with_values(self, source):
        """Rebuild sources whose rows keep their values in HBM only (``upsert_arrays(keep_host_copy=False)``): the
        values are read back from the index BEFORE ``rebuild`` closes the engines that hold the only copy."""
        fetch = getattr(self._index, "fetch_values_by_id", None)
        out = {}
        for name, rows in source.items():
            rows = list(rows)
            missing = [i for i, v in enumerate(rows) if getattr(v, "values", None) is None]
            if missing:
                if fetch is None:
                    raise RuntimeError(f"namespace {name!r}: stored rows carry no values and the index cannot supply them")
                from .storage import StoredRow

                vals = fetch(name, [rows[i].id for i in missing])
                for i, val in zip(missing, vals):
                    rows[i] = StoredRow(rows[i].id, val, rows[i].metadata)
            out[name] = rows
        return out

    # ---- introspection (query_processor.py:64-82)
    def list_namespaces(self) -> List[str]:
        return self._storage.list_namespaces

    def get_namespace_vectors(self, namespace: str) -> List[Dict[str, Any]]:
        return [{"id": v.id, "values": v.values, "metadata": v.metadata}
                for v in self._storage.namespace_map.get(namespace, [])]

    def get_namespace_count(self, namespace: str) -> int:
        return len(self._storage.namespace_map.get(namespace, []))

    def get_storage_info(self) -> Dict[str, Any]:
        """Passthrough the REST layer relies on (query_processor.py:81-82; rest_api.py:283)."""
        return self._storage.get_storage_info()

    # ---- additive: bulk ingest without one Python object per row (ArrayStorage + Index.add_arrays)
    def upsert_arrays(self, values: np.ndarray, namespace: str = "default", metadata=None, *,
                      keep_host_copy: bool = True) -> np.ndarray:
        """``upsert_many`` for an ``[n, dim]`` matrix: mints ids, hands the rows to the index and (ids, metadata and --
        unless ``keep_host_copy=False`` -- the values) to the storage.  Returns the ``[n, 16] uint8`` id table."""
        if not hasattr(self._storage, "write_arrays") or not hasattr(self._index, "add_arrays"):
            raise RuntimeError("upsert_arrays needs an array-backed storage (ArrayStorage) and this package's Index")
        from .idtable import mint_uuid4_bytes

        values = np.ascontiguousarray(values, dtype=np.float32)
        if values.ndim != 2:
            raise RuntimeError(f"Wrong dimensionality of the vectors: expected a matrix, got shape {values.shape}")
        if not keep_host_copy and self._rebuild_scope == "namespace":
            # Q4's rebuild drops every other namespace from the index; with the values in HBM only that would be data loss
            raise ValueError('keep_host_copy=False needs rebuild_scope="all": a namespace-scoped rebuild closes the other '
                             "namespaces' engines, which would hold the only copy of their rows")
        # the index's refusals (dimension, non-finite rows) come before the storage is written: no ghost rows
        if hasattr(self._index, "validate_arrays"):
            self._index.validate_arrays(values, namespace)
        ids = mint_uuid4_bytes(values.shape[0])
        first = self._storage.write_arrays(ids, namespace, values if keep_host_copy else None, metadata)
        self._index.add_arrays(values, namespace, ids=ids,
                               handles=np.arange(first, first + values.shape[0], dtype=np.int64))
        return ids
