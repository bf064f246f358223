"""``IdTable``: the UUID <-> label maps of one namespace as arrays.

The reference keeps two Python dicts per namespace (``_uuid_to_label`` / ``_label_to_uuid``,
src/mlvectordb/implementations/index.py:21-22,56-63).  At the corpus sizes this index is built for
(10M-80M rows per node) two dicts of ``uuid.UUID`` objects are gigabytes of host memory and minutes of
``add``; the per-hit dict lookups of a 256-query wave (2,560 hits) cost as much as the GPU scan itself.  Here:

* label -> id is a ``[rows, 16] uint8`` table (the on-disk format of ``Index.save_index`` too); a whole result
  batch is mapped by one fancy-indexing step;
* ``uuid.UUID`` objects are kept only as a cache (an object array: the ids of rows that were added as
  ``Vector`` objects are already alive in the caller's storage, so the cache costs one pointer per row; rows
  added in bulk get their object on the first hit);
* id -> label is a sorted array of 8-byte keys mixed from all 16 id bytes + ``searchsorted``, built on the first
  ``remove`` / filtered search and extended by merging, never a dict.

Semantics kept from the reference: labels are dense in insertion order; an id removed twice is a no-op the second
time (index.py:76-81, ``dict.pop``); when the same id was added twice the newest row is the only one the id resolves
to (the dict entry was overwritten, index.py:62): after it is removed the id is unknown, the older row stays.
"""
from __future__ import annotations

import os
from typing import Iterable, Sequence
from uuid import UUID

import numpy as np


def mint_uuid4_bytes(n: int) -> np.ndarray:
    """``n`` random version-4 UUIDs as a ``[n, 16] uint8`` table (what ``uuid.uuid4().bytes`` would give)."""
    raw = np.frombuffer(os.urandom(16 * n), dtype=np.uint8).reshape(n, 16).copy()
    raw[:, 6] = (raw[:, 6] & 0x0F) | 0x40  # version 4
    raw[:, 8] = (raw[:, 8] & 0x3F) | 0x80  # RFC 4122 variant
    return raw


def uuids_to_bytes(ids: Sequence[UUID]) -> np.ndarray:
    if not len(ids):
        return np.zeros((0, 16), dtype=np.uint8)
    return np.frombuffer(b"".join([u.bytes for u in ids]), dtype=np.uint8).reshape(len(ids), 16)


class IdTable:
    __slots__ = ("raw", "objs", "live", "handles", "n", "_keys", "_labels", "_indexed")

    def __init__(self) -> None:
        self.raw = np.zeros((0, 16), dtype=np.uint8)
        self.objs = np.empty(0, dtype=object)
        self.live = np.zeros(0, dtype=bool)
        # optional per-row payload of the caller (an ArrayStorage row number): lets a batch of hits be resolved in the
        # caller's store by direct indexing instead of 2,560 id lookups per query wave; -1 = none
        self.handles = np.zeros(0, dtype=np.int64)
        self.n = 0
        self._keys = np.zeros(0, dtype=np.uint64)    # sorted first-8-byte keys of labels [0, _indexed)
        self._labels = np.zeros(0, dtype=np.int64)   # their labels
        self._indexed = 0

    # ------------------------------------------------------------------ growth
    def _reserve(self, extra: int) -> None:
        need = self.n + extra
        if need <= self.raw.shape[0]:
            return
        cap = max(need, self.raw.shape[0] + self.raw.shape[0] // 2, 64)
        raw = np.zeros((cap, 16), dtype=np.uint8)
        raw[: self.n] = self.raw[: self.n]
        objs = np.empty(cap, dtype=object)
        objs[: self.n] = self.objs[: self.n]
        live = np.zeros(cap, dtype=bool)
        live[: self.n] = self.live[: self.n]
        handles = np.full(cap, -1, dtype=np.int64)
        handles[: self.n] = self.handles[: self.n]
        self.raw, self.objs, self.live, self.handles = raw, objs, live, handles

    def append_uuids(self, ids: Sequence[UUID]) -> int:
        """Rows added as objects that carry their id (``Index.add``); returns the first label."""
        first, m = self.n, len(ids)
        self._reserve(m)
        self.raw[first:first + m] = uuids_to_bytes(ids)
        self.objs[first:first + m] = ids
        self.live[first:first + m] = True
        self.n += m
        return first

    def append_raw(self, raw: np.ndarray, handles: np.ndarray | None = None) -> int:
        """Rows added in bulk with a ``[m, 16] uint8`` id table; UUID objects are made on demand."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 16)
        first, m = self.n, raw.shape[0]
        self._reserve(m)
        self.raw[first:first + m] = raw
        self.live[first:first + m] = True
        if handles is not None:
            self.handles[first:first + m] = handles
        self.n += m
        return first

    # ------------------------------------------------------------------ label -> id
    def uuids_at(self, labels: np.ndarray) -> np.ndarray:
        """Object array of ``uuid.UUID`` with the shape of ``labels``; ``None`` where the label is < 0 or not live."""
        labels = np.asarray(labels, dtype=np.int64)
        flat = labels.ravel()
        ok = (flat >= 0) & (flat < self.n)
        safe = np.where(ok, flat, 0)
        if self.n == 0:
            return np.full(labels.shape, None, dtype=object)
        ok &= self.live[safe]
        out = self.objs[safe]
        missing = np.flatnonzero(ok & np.equal(out, None))
        if missing.size:
            for lab in np.unique(flat[missing]).tolist():
                self.objs[lab] = UUID(bytes=self.raw[lab].tobytes())
            out = self.objs[safe]
        out[~ok] = None
        return out.reshape(labels.shape)

    # ------------------------------------------------------------------ id -> label
    def _key_of(self, raw: np.ndarray) -> np.ndarray:
        # all 16 bytes take part (hi ^ lo * odd constant, wrapping): ids that share their first 8 bytes -- UUID(int=i)
        # -- still get distinct keys, so lookups stay on the vectorised one-candidate path
        halves = np.ascontiguousarray(raw).view(np.uint64).reshape(-1, 2)
        return halves[:, 0] ^ (halves[:, 1] * np.uint64(0x9E3779B97F4A7C15))

    def _extend_index(self) -> None:
        if self._indexed == self.n:
            return
        tail_keys = self._key_of(self.raw[self._indexed:self.n])
        order = np.argsort(tail_keys, kind="stable")
        keys = np.concatenate([self._keys, tail_keys[order]])
        labels = np.concatenate([self._labels, order.astype(np.int64) + self._indexed])
        if self._indexed:  # two sorted runs: a stable sort of their concatenation is a merge
            order = np.argsort(keys, kind="stable")
            keys, labels = keys[order], labels[order]
        self._keys, self._labels, self._indexed = keys, labels, self.n

    def lookup(self, ids: Iterable[UUID]) -> np.ndarray:
        """Labels of ``ids`` (int64, -1 for ids that are unknown or already removed)."""
        ids = ids if isinstance(ids, (list, tuple)) else list(ids)
        out = np.full(len(ids), -1, dtype=np.int64)
        if not ids or self.n == 0:
            return out
        good = [i for i, u in enumerate(ids) if isinstance(u, UUID)]
        if not good:
            return out
        out[np.asarray(good, dtype=np.int64)] = self.lookup_raw(uuids_to_bytes([ids[i] for i in good]))
        return out

    def lookup_raw(self, raw: np.ndarray) -> np.ndarray:
        """``lookup`` for a ``[m, 16] uint8`` table of id bytes: no Python object per id."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 16)
        out = np.full(raw.shape[0], -1, dtype=np.int64)
        if not raw.shape[0] or self.n == 0:
            return out
        good = np.arange(raw.shape[0], dtype=np.int64)
        self._extend_index()
        qk = self._key_of(raw)
        lo = np.searchsorted(self._keys, qk, side="left")
        hi = np.searchsorted(self._keys, qk, side="right")
        one = np.flatnonzero(hi - lo == 1)
        if one.size:
            cand = self._labels[lo[one]]
            match = (self.raw[cand] == raw[one]).all(axis=1) & self.live[cand]
            out[good[one[match]]] = cand[match]
        for j in np.flatnonzero(hi - lo > 1).tolist():  # key collision or the same id added twice
            cand = self._labels[lo[j]:hi[j]]
            match = cand[(self.raw[cand] == raw[j]).all(axis=1)]
            # the reference's dict holds ONE entry per id, the newest row (index.py:62); once that row is removed the id
            # is unknown (dict.pop, index.py:76-81) -- an older row of the same id is never found again
            if match.size and self.live[match.max()]:
                out[good[j]] = match.max()
        return out

    def kill(self, labels: np.ndarray) -> None:
        self.live[np.asarray(labels, dtype=np.int64)] = False

    # ------------------------------------------------------------------ whole-table operations
    def dead_labels(self) -> np.ndarray:
        return np.flatnonzero(~self.live[: self.n]).astype(np.int64)

    def take(self, old_labels: np.ndarray) -> "IdTable":
        """The table after a compaction that keeps ``old_labels`` (ascending) as labels 0..len-1."""
        old_labels = np.asarray(old_labels, dtype=np.int64)
        if old_labels.size and not self.live[old_labels].all():
            bad = int(old_labels[~self.live[old_labels]][0])
            raise RuntimeError(f"compaction kept label {bad}, which the host maps do not know")
        t = IdTable()
        t.raw = self.raw[old_labels].copy()
        t.objs = self.objs[old_labels].copy()
        t.live = np.ones(old_labels.size, dtype=bool)
        t.handles = self.handles[old_labels].copy()
        t.n = int(old_labels.size)
        return t
