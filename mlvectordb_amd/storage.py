"""Row stores, only what ``QueryProcessor`` needs around the hot path.

The reference's storage engine (src/mlvectordb/implementations/storage_engine_in_memory.py)
is out of scope (host-side metadata store, no arithmetic).  Two minimal equivalents:

* ``InMemoryStorage`` -- a dict of row objects per namespace, the shape of the reference's engine: lets
  ``find_similar`` enrich hits (query_processor.py:36-48) and ``delete`` trigger a rebuild (query_processor.py:51-62).
* ``ArrayStorage`` -- the same surface over arrays, for corpora of millions of rows where one Python object per
  row is not an option (10M ``Vector`` objects are ~40 GB and minutes of allocation): per namespace an id table
  (``IdTable``), an optional ``[n, dim]`` float32 matrix and an optional metadata list.  A namespace written without
  ``values`` keeps no second copy of the corpus on the host: ``QueryProcessor`` then reads the hits' values back from
  the index's own rows in HBM (``Index.fetch_values``, bit-exact).
"""
from __future__ import annotations

from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple
from uuid import UUID

import numpy as np

from .idtable import IdTable, uuids_to_bytes
from .interfaces import VectorProtocol


class InMemoryStorage:
    storage_type = "memory"

    def __init__(self) -> None:
        self._rows: Dict[str, Dict[UUID, VectorProtocol]] = {}

    @property
    def total_vectors(self) -> int:
        return sum(len(ns) for ns in self._rows.values())

    def write(self, vector: VectorProtocol, namespace: str) -> bool:
        self._rows.setdefault(namespace, {})[vector.id] = vector
        return True

    def write_vectors(self, vectors: List[VectorProtocol], namespace: str) -> List[bool]:
        return [self.write(v, namespace) for v in vectors]

    def read_vectors(self, vector_ids: List[UUID], namespace: str) -> List[Optional[VectorProtocol]]:
        ns = self._rows.get(namespace, {})
        return [ns.get(i) for i in vector_ids]

    def delete(self, vector_id: UUID, namespace: str) -> bool:
        ns = self._rows.get(namespace)
        if ns is None or vector_id not in ns:
            return False
        del ns[vector_id]
        if not ns:
            del self._rows[namespace]
        return True

    @property
    def namespace_map(self) -> Mapping[str, List[VectorProtocol]]:
        return {name: list(ns.values()) for name, ns in self._rows.items()}

    @property
    def list_namespaces(self) -> List[str]:
        return list(self._rows)

    def get_storage_info(self) -> Dict[str, Any]:
        """Counterpart of the reference engine's summary (storage_engine_in_memory.py:61-69), which the REST layer
        reads through ``QueryProcessor.get_storage_info`` (query_processor.py:81-82)."""
        return {"storage_type": self.storage_type, "total_vectors": self.total_vectors,
                "storage_size_bytes": sum(np.asarray(v.values).nbytes for ns in self._rows.values() for v in ns.values()),
                "namespaces": list(self._rows), "vectors_per_namespace": {n: len(r) for n, r in self._rows.items()},
                "namespace_count": len(self._rows)}


class StoredRow:
    """What ``ArrayStorage.read_vectors`` hands out: a ``VectorProtocol`` view of one row of the arrays."""

    __slots__ = ("id", "values", "metadata")

    def __init__(self, id: UUID, values, metadata) -> None:  # noqa: A002
        self.id, self.values, self.metadata = id, values, metadata

    def shape(self) -> tuple:
        return np.asarray(self.values).shape


class _MetaColumn:
    """Per-row metadata of a namespace, kept per written chunk: ``None`` for a chunk written without metadata (every row
    then reads as the one shared empty dict).  A flat Python list of one reference per row -- 10M references for the
    benchmark corpus -- is walked by every full pass of the cyclic garbage collector: 50-100 ms pauses in the middle of
    a query stream (round 3: the protocol-level stream dropped from 2.2 to 8 ms per wave whenever one landed in it)."""

    __slots__ = ("chunks", "starts", "any")

    def __init__(self) -> None:
        self.chunks: List[Optional[list]] = []
        self.starts: List[int] = [0]
        self.any = False

    _COALESCE = 4096  # rows: consecutive small chunks are merged up to this size (QueryProcessor.insert writes one row at a time)

    def append(self, metadata, n: int) -> None:
        last = self.chunks[-1] if self.chunks else None
        last_n = self.starts[-1] - self.starts[-2] if self.chunks else 0
        if self.chunks and last_n + n <= self._COALESCE and (metadata is None) == (last is None):
            if metadata is not None:
                last.extend(metadata)
            self.starts[-1] += n
        else:
            self.chunks.append(None if metadata is None else list(metadata))
            self.starts.append(self.starts[-1] + n)
        self.any = self.any or metadata is not None

    def __getitem__(self, r: int):
        if not self.any:
            return _EMPTY
        c = int(np.searchsorted(self.starts, r, side="right")) - 1
        chunk = self.chunks[c]
        return _EMPTY if chunk is None else chunk[r - self.starts[c]]

    def take(self, rows) -> List[Any]:
        """Metadata of ``rows`` (a list of row numbers, -1 = not found -> None)."""
        if not self.any:
            return [_EMPTY if r >= 0 else None for r in rows]
        if len(self.chunks) == 1:
            chunk = self.chunks[0]
            return [chunk[r] if r >= 0 else None for r in rows]
        which = (np.searchsorted(self.starts, np.asarray(rows, dtype=np.int64), side="right") - 1).tolist()
        out = []
        for r, c in zip(rows, which):
            if r < 0:
                out.append(None)
            else:
                chunk = self.chunks[c]
                out.append(_EMPTY if chunk is None else chunk[r - self.starts[c]])
        return out


class _ArrayNamespace:
    __slots__ = ("ids", "chunks", "starts", "metadata", "dim")

    def __init__(self) -> None:
        self.ids = IdTable()
        self.chunks: List[Optional[np.ndarray]] = []  # values of rows [starts[i], starts[i+1]) or None (kept in HBM only)
        self.starts: List[int] = [0]
        self.metadata = _MetaColumn()
        self.dim: Optional[int] = None


class ArrayStorage:
    storage_type = "array"

    def __init__(self) -> None:
        self._ns: Dict[str, _ArrayNamespace] = {}

    # ------------------------------------------------------------------ bulk surface
    def write_arrays(self, ids: np.ndarray, namespace: str, values: Optional[np.ndarray] = None,
                     metadata: Optional[Sequence[Mapping[str, Any]]] = None) -> int:
        """Append ``n`` rows: ``ids`` ``[n, 16] uint8``, optional ``values`` ``[n, dim]`` (kept by reference, not
        copied) and optional per-row metadata (default: one shared empty dict).  Returns the first row number: rows
        are numbered densely per namespace in write order and never renumbered, so the numbers can be handed to
        ``Index.add_arrays(handles=...)`` and used with ``read_rows_at``."""
        ids = np.ascontiguousarray(ids, dtype=np.uint8).reshape(-1, 16)
        n = ids.shape[0]
        if values is not None and (values.ndim != 2 or values.shape[0] != n):
            raise ValueError(f"values has shape {values.shape}, expected [{n}, dim]")
        if metadata is not None and len(metadata) != n:
            raise ValueError(f"{len(metadata)} metadata entries for {n} rows")
        ns = self._ns.setdefault(namespace, _ArrayNamespace())
        first = ns.ids.append_raw(ids)
        ns.chunks.append(None if values is None else np.asarray(values, dtype=np.float32))
        ns.starts.append(ns.starts[-1] + n)
        ns.metadata.append(metadata, n)
        return first

    def read_rows_at(self, rows: np.ndarray, namespace: str) -> Tuple[np.ndarray, Optional[np.ndarray], List[Any]]:
        """``read_rows_raw`` by row number (the handles ``write_arrays`` returned): no id lookup at all."""
        rows = np.asarray(rows, dtype=np.int64).ravel()
        ns = self._ns.get(namespace)
        if ns is None:
            return np.zeros(rows.size, dtype=bool), None, [None] * rows.size
        ok = (rows >= 0) & (rows < ns.ids.n)
        found = ok & ns.ids.live[np.where(ok, rows, 0)]
        rows = np.where(found, rows, -1)
        return found, self._values_of(ns, rows), ns.metadata.take(rows.tolist())

    def read_rows_raw(self, raw_ids: np.ndarray, namespace: str
                      ) -> Tuple[np.ndarray, Optional[np.ndarray], List[Any]]:
        """Fast path of ``QueryProcessor``: ``(found bool [m], values float32 [m, dim] or None, metadata list [m])`` for
        a ``[m, 16]`` table of id bytes.  ``values`` is None when any requested row's values live in HBM only."""
        raw_ids = np.ascontiguousarray(raw_ids, dtype=np.uint8).reshape(-1, 16)
        ns = self._ns.get(namespace)
        if ns is None:
            return np.zeros(raw_ids.shape[0], dtype=bool), None, [None] * raw_ids.shape[0]
        rows = ns.ids.lookup_raw(raw_ids)
        found = rows >= 0
        return found, self._values_of(ns, rows), ns.metadata.take(rows.tolist())

    @staticmethod
    def _values_of(ns: _ArrayNamespace, rows: np.ndarray) -> Optional[np.ndarray]:
        if not rows.size or any(c is None for c in ns.chunks):
            return None
        if len(ns.chunks) == 1:
            return ns.chunks[0][np.maximum(rows, 0)]
        which = np.searchsorted(np.asarray(ns.starts[1:]), np.maximum(rows, 0), side="right")
        out = np.empty((rows.size, ns.chunks[0].shape[1]), dtype=np.float32)
        for c in np.unique(which).tolist():
            sel = which == c
            out[sel] = ns.chunks[c][np.maximum(rows[sel], 0) - ns.starts[c]]
        return out

    # ------------------------------------------------------------------ the reference's storage surface
    def write(self, vector: VectorProtocol, namespace: str) -> bool:
        return self.write_vectors([vector], namespace)[0]

    def write_vectors(self, vectors: List[VectorProtocol], namespace: str) -> List[bool]:
        if vectors:
            self.write_arrays(uuids_to_bytes([v.id for v in vectors]), namespace,
                              np.asarray([v.values for v in vectors], dtype=np.float32), [v.metadata for v in vectors])
        return [True] * len(vectors)

    def read_vectors(self, vector_ids: List[UUID], namespace: str) -> List[Optional[VectorProtocol]]:
        ns = self._ns.get(namespace)
        if ns is None:
            return [None] * len(vector_ids)
        rows = ns.ids.lookup(vector_ids)
        vals = self._values_of(ns, rows)
        rl = rows.tolist()
        meta = ns.metadata.take(rl)  # one vectorised chunk lookup (a per-row searchsorted made this quadratic after single inserts)
        return [StoredRow(u, None if vals is None else vals[i], meta[i]) if r >= 0 else None
                for i, (u, r) in enumerate(zip(vector_ids, rl))]

    def delete(self, vector_id: UUID, namespace: str) -> bool:
        ns = self._ns.get(namespace)
        if ns is None:
            return False
        row = int(ns.ids.lookup([vector_id])[0])
        if row < 0:
            return False
        ns.ids.kill([row])
        return True

    @property
    def total_vectors(self) -> int:
        return sum(int(ns.ids.live[:ns.ids.n].sum()) for ns in self._ns.values())

    @property
    def namespace_map(self) -> Mapping[str, List[VectorProtocol]]:
        """Materialises row objects: meant for small namespaces (tests, rebuild-from-storage)."""
        out = {}
        for name, ns in self._ns.items():
            rows = np.flatnonzero(ns.ids.live[:ns.ids.n])
            vals = self._values_of(ns, rows)
            ids = ns.ids.uuids_at(rows).tolist()
            meta = ns.metadata.take(rows.tolist())
            out[name] = [StoredRow(u, None if vals is None else vals[i], meta[i]) for i, u in enumerate(ids)]
        return out

    @property
    def list_namespaces(self) -> List[str]:
        return list(self._ns)

    def get_storage_info(self) -> Dict[str, Any]:
        per = {name: int(ns.ids.live[:ns.ids.n].sum()) for name, ns in self._ns.items()}
        return {"storage_type": self.storage_type, "total_vectors": sum(per.values()),
                "storage_size_bytes": sum(c.nbytes for ns in self._ns.values() for c in ns.chunks if c is not None),
                "namespaces": list(self._ns), "vectors_per_namespace": per, "namespace_count": len(self._ns)}


_EMPTY: Mapping[str, Any] = {}
