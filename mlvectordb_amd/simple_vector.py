"""``SimpleVector``: the vector class the reference's README shows (README.md:27-41,170-185) and the north star
names (``SimpleVector.distance()`` / ``.similarity()``), which the reference never implemented
(no ``implementations/simple_vector.py`` exists there).  Constructor and attribute names follow the README:
``SimpleVector(id: str, data: np.ndarray, metadata: dict)``, ``distance`` / ``similarity`` / ``normalize`` /
``to_dict`` / ``from_dict``.  It also satisfies ``VectorProtocol`` (``values`` aliases ``data``; ``shape()``), so an
``Index`` accepts it when its id is a ``uuid.UUID``.  Arithmetic: ``pairwise.py``.  Parity unpinned.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import numpy as np

from . import pairwise


class SimpleVector:
    __slots__ = ("id", "data", "metadata")

    def __init__(self, id, data, metadata: Optional[Mapping[str, Any]] = None) -> None:  # noqa: A002 (README's name)
        self.id = id
        self.data = np.array(data, dtype=np.float32)
        if self.data.ndim != 1:
            raise ValueError(f"a vector is one-dimensional, got shape {self.data.shape}")
        self.metadata = dict(metadata or {})

    @property
    def values(self) -> np.ndarray:
        return self.data

    @property
    def dimension(self) -> int:
        return int(self.data.shape[0])

    def shape(self) -> tuple:
        return self.data.shape

    def distance(self, other, metric: str = "euclidean") -> float:
        return pairwise.distance(self.data, _values_of(other), metric)

    def similarity(self, other, metric: str = "cosine") -> float:
        return pairwise.similarity(self.data, _values_of(other), metric)

    def normalize(self) -> "SimpleVector":
        return SimpleVector(self.id, pairwise.normalize(self.data), self.metadata)

    def to_dict(self) -> Dict[str, Any]:
        return {"id": self.id, "data": self.data.tolist(), "metadata": dict(self.metadata)}

    @classmethod
    def from_dict(cls, d: Mapping[str, Any]) -> "SimpleVector":
        return cls(d["id"], d["data"], d.get("metadata"))

    def __repr__(self) -> str:
        return f"SimpleVector(id={self.id!r}, dim={self.dimension}, metadata={self.metadata})"

    def __eq__(self, other: object) -> bool:
        return (isinstance(other, SimpleVector) and self.id == other.id and np.array_equal(self.data, other.data)
                and self.metadata == other.metadata)

    def __hash__(self) -> int:
        return hash(self.id)


def _values_of(other):
    for name in ("data", "values"):
        v = getattr(other, name, None)
        if v is not None:
            return v
    return other
