"""Minimal in-memory row store, only what ``QueryProcessor`` needs around the hot path.

The reference's storage engine (src/mlvectordb/implementations/storage_engine_in_memory.py)
is out of scope (host-side metadata store, no arithmetic); this is the
smallest equivalent that lets ``find_similar`` enrich hits (query_processor.py:36-48) and
``delete`` trigger a rebuild (query_processor.py:51-62).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional
from uuid import UUID

from .interfaces import VectorProtocol


class InMemoryStorage:
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
