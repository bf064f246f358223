"""Row-sharded search across the GPUs of one node: one process per GPU, per-shard top-k,
host-side k-way merge.

The reference has no multi-device code.  The sharding follows the north star:
rank g holds the contiguous label range [offset_g, offset_g + n_g); every rank scans its shard
for the whole query wave; the per-shard ``[nq, k]`` candidates (12-20 bytes each) are gathered
on rank 0 and merged there by (fp64 distance, global label).  No collective touches the
corpus path; the only exchange is that gather (a few hundred KB at most), done over the
process group's CPU (gloo) side so it also runs on machines without a GPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def merge_topk(labels: Sequence[np.ndarray], dist64: Sequence[np.ndarray], k: int, return_dist64: bool = False
               ) -> Tuple[np.ndarray, ...]:
    """Merge per-shard results ``[nq, k_s]`` (global labels, -1 = padding) into the global top-k.

    Ranking is (distance ascending, label ascending) on the float64 distances -- the same
    order each shard used internally -- so the merged ids equal a single-index search.
    Returns (labels int64 [nq, k], dist float32 [nq, k], counts int32 [nq]) and, with ``return_dist64``, the merged
    float64 distances [nq, k] as a fourth entry (for a caller that merges further).
    """
    lab = np.concatenate([np.asarray(l, dtype=np.int64) for l in labels], axis=1)
    d = np.concatenate([np.asarray(x, dtype=np.float64) for x in dist64], axis=1)
    nq = lab.shape[0]
    d = np.where(lab < 0, np.inf, d)
    big = np.iinfo(np.int64).max
    key_lab = np.where(lab < 0, big, lab)
    # Vectorised (a Python loop over 1024 queries costs more than the GPU wave): cut every row down to its
    # k + 8 smallest distances first, sort those lexicographically, and fall back to sorting whole rows only
    # if a row has more than 8 entries tied with its k-th distance (the cut could then drop a lower label).
    width = min(k, lab.shape[1])
    m = k + 8
    if lab.shape[1] > m:
        part = np.argpartition(d, m - 1, axis=1)[:, :m]
        pd, pk = np.take_along_axis(d, part, axis=1), np.take_along_axis(key_lab, part, axis=1)
        sub = np.lexsort((pk, pd), axis=1)
        order = np.take_along_axis(part, sub, axis=1)
        sd = np.take_along_axis(pd, sub, axis=1)
        tied = np.isfinite(sd[:, k - 1]) & (sd[:, m - 1] == sd[:, k - 1])
        if tied.any():
            order[tied] = np.lexsort((key_lab[tied], d[tied]), axis=1)[:, :m]
        order = order[:, :width]
    else:
        order = np.lexsort((key_lab, d), axis=1)[:, :width]
    sel_l = np.take_along_axis(lab, order, axis=1)
    sel_d = np.take_along_axis(d, order, axis=1)
    valid = sel_l >= 0  # padding sorts last: (inf, max label)
    out_l = np.full((nq, k), -1, dtype=np.int64)
    out_d = np.full((nq, k), np.inf, dtype=np.float32)
    out_l[:, :width] = np.where(valid, sel_l, -1)
    out_d[:, :width] = np.where(valid, sel_d, np.inf).astype(np.float32)
    counts = valid.sum(axis=1).astype(np.int32)
    if return_dist64:
        out_d64 = np.full((nq, k), np.inf, dtype=np.float64)
        out_d64[:, :width] = np.where(valid, sel_d, np.inf)
        return out_l, out_d, counts, out_d64
    return out_l, out_d, counts


class ShardedSearcher:
    """One rank's view of a row-sharded corpus.

    ``local_search(queries, k) -> (labels_local int64 [nq,k], dist64 float64 [nq,k])`` is the
    rank's shard scan (HIP engine in production, oracle engine in CPU tests); ``row_offset`` is
    the global label of the shard's row 0.
    """

    def __init__(self, local_search, row_offset: int, group=None) -> None:
        self._local_search = local_search
        self._offset = int(row_offset)
        self._group = group

    def search(self, queries: np.ndarray, k: int) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        """Collective: every rank calls it with the same queries; rank 0 gets the merged result."""
        import torch
        import torch.distributed as dist

        labels, d64 = self._local_search(queries, k)
        labels = np.where(labels >= 0, labels + self._offset, -1).astype(np.int64)
        world = dist.get_world_size(self._group) if dist.is_initialized() else 1
        if world == 1:
            return merge_topk([labels], [d64], k)
        rank = dist.get_rank(self._group)
        t_lab = torch.from_numpy(np.ascontiguousarray(labels))
        t_d = torch.from_numpy(np.ascontiguousarray(d64, dtype=np.float64))
        if rank == 0:
            g_lab = [torch.empty_like(t_lab) for _ in range(world)]
            g_d = [torch.empty_like(t_d) for _ in range(world)]
        else:
            g_lab = g_d = None
        dst = dist.get_global_rank(self._group, 0) if self._group is not None else 0
        dist.gather(t_lab, g_lab, dst=dst, group=self._group)
        dist.gather(t_d, g_d, dst=dst, group=self._group)
        if rank != 0:
            return None
        return merge_topk([t.numpy() for t in g_lab], [t.numpy() for t in g_d], k)


def shard_bounds(n_total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal row ranges [begin, end) per rank."""
    base, rem = divmod(int(n_total), int(world))
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out
