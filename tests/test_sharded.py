"""Row-sharded search: host merge, and the N>1 path over gloo with world_size 2 (CPU)."""
import os
import socket

import numpy as np
import pytest

from mlvectordb_amd.sharded import ShardedSearcher, merge_topk, shard_bounds
from oracle import exact_scan


def test_shard_bounds_cover_everything():
    assert shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard_bounds(80_000_000, 8)[-1] == (70_000_000, 80_000_000)


def _local(rows, space):
    def fn(q, k):
        labels, _, _ = exact_scan.knn(q, rows, k, space)
        d = exact_scan.exact_distances(q, rows, space) if len(rows) else np.zeros((len(q), 0))
        d64 = np.full(labels.shape, np.inf)
        for i in range(labels.shape[0]):
            ok = labels[i] >= 0
            d64[i, ok] = d[i, labels[i, ok]]
        return labels, d64
    return fn


@pytest.mark.parametrize("space", ["l2", "cosine"])
def test_merge_of_shards_equals_single_index(space):
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((1000, 24)).astype(np.float32)
    rows[700] = rows[5]  # a cross-shard exact tie: the lower label must win
    qs = rng.standard_normal((9, 24)).astype(np.float32)
    qs[0] = rows[5]
    want = exact_scan.knn(qs, rows, 10, space)
    parts_l, parts_d = [], []
    for b, e in shard_bounds(1000, 4):
        l, d = _local(rows[b:e], space)(qs, 10)
        parts_l.append(np.where(l >= 0, l + b, -1))
        parts_d.append(d)
    got = merge_topk(parts_l, parts_d, 10)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert np.allclose(got[1], want[1], atol=1e-6)


def test_merge_with_short_shards_pads():
    l = [np.array([[0, -1]]), np.array([[5, 6]])]
    d = [np.array([[0.5, np.inf]]), np.array([[0.25, 0.75]])]
    labels, dist, counts = merge_topk(l, d, 4)
    assert labels.tolist() == [[5, 0, 6, -1]] and counts.tolist() == [3]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seed, out_q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        rows = rng.standard_normal((600, 16)).astype(np.float32)
        qs = rng.standard_normal((5, 16)).astype(np.float32)
        b, e = shard_bounds(600, world)[rank]
        searcher = ShardedSearcher(_local(rows[b:e], "cosine"), row_offset=b)
        res = searcher.search(qs, 7)
        if rank == 0:
            want = exact_scan.knn(qs, rows, 7, "cosine")
            out_q.put(bool(np.array_equal(res[0], want[0]) and np.allclose(res[1], want[1], atol=1e-6)))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_search():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 21, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
