"""Row record used by ``Index.add`` / ``Index.rebuild``.

Counterpart of the reference's ``Vector`` (src/mlvectordb/implementations/vector.py:10-42):
a fresh ``uuid4`` id per instance and values coerced to ``np.float32`` -- float32 is the
layout contract of the corpus in HBM.
"""
from __future__ import annotations

import uuid
from typing import Any, Mapping, Sequence
from uuid import UUID

import numpy as np

from . import pairwise


class Vector:
    __slots__ = ("_id", "_values", "_metadata")

    def __init__(self, values: Sequence[float], metadata: Mapping[str, Any] | None = None) -> None:
        self._id = uuid.uuid4()
        self._values = np.array(values, dtype=np.float32)
        self._metadata = metadata or {}

    @property
    def id(self) -> UUID:
        return self._id

    @property
    def values(self) -> np.ndarray:
        return self._values

    @property
    def metadata(self) -> Mapping[str, Any]:
        return self._metadata

    def shape(self) -> tuple:
        return self._values.shape

    # ---- pairwise arithmetic (reference README.md:30-41,178-181 names the methods; definitions: pairwise.py)
    def distance(self, other, metric: str = "euclidean") -> float:
        return pairwise.distance(self._values, getattr(other, "values", other), metric)

    def similarity(self, other, metric: str = "cosine") -> float:
        return pairwise.similarity(self._values, getattr(other, "values", other), metric)

    def normalize(self) -> "Vector":
        """A new vector (new id, same metadata) with values x / (|x| + 1e-30)."""
        return Vector(pairwise.normalize(self._values), self._metadata)

    def __repr__(self) -> str:
        return f"Vector(id={self._id}, dim={self.shape()}, metadata={self._metadata})"

    def __eq__(self, other: object) -> bool:
        return (isinstance(other, Vector) and self._id == other._id
                and np.array_equal(self._values, other._values) and self._metadata == other._metadata)

    def __hash__(self) -> int:
        return hash(self._id)
