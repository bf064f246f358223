"""Protocol surface of the hot path -- the drop-in boundary on the Python side.

Restates (does not copy) the reference's structural types so that code written against
MLVectorDB's interfaces type-checks against this package unchanged:

* ``VectorProtocol`` / ``VectorDTO``      reference src/mlvectordb/interfaces/vector.py:7-22
* ``SearchResultProtocol`` / ``IndexProtocol``  reference src/mlvectordb/interfaces/index.py:5-13
* ``QueryProcessorProtocol``              reference src/mlvectordb/interfaces/query_processor.py:7-11
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Iterable, List, Mapping, Protocol, Sequence, runtime_checkable
from uuid import UUID

import numpy as np


@runtime_checkable
class VectorProtocol(Protocol):
    """A stored row: immutable id, float32 values, free-form metadata."""

    id: UUID
    values: np.ndarray
    metadata: Mapping[str, Any]

    def shape(self) -> tuple: ...


@dataclass
class VectorDTO:
    """Carrier for a query or a row to insert (no id yet)."""

    values: Sequence[float]
    metadata: Mapping[str, Any] = field(default_factory=dict)


class SearchResultProtocol(Protocol):
    vector_id: UUID
    score: float


class IndexProtocol(Protocol):
    def add(self, vectors: Iterable[VectorProtocol], namespace: str) -> None: ...

    def remove(self, ids: Sequence[UUID], namespace: str) -> None: ...

    def search(self, query: VectorDTO, top_k: int, namespace: str, metric: str) -> List[SearchResultProtocol]: ...

    def rebuild(self, source: Mapping[str, Iterable[VectorProtocol]], metric: str) -> None: ...


class QueryProcessorProtocol(Protocol):
    def insert(self, vector: VectorDTO, namespace: str = "default") -> None: ...

    def upsert_many(self, vectors: Iterable[VectorDTO], namespace: str = "default") -> None: ...

    def find_similar(self, query: VectorDTO, top_k: int, namespace: str = "default",
                     metric: str = "cosine") -> List[dict]: ...

    def delete(self, ids: Sequence[UUID], namespace: str = "default") -> Sequence[UUID]: ...
