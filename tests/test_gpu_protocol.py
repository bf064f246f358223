"""The Protocol-level paths on the GPU (through the C ABI), each against the same flow on the NumPy oracle engine:

* ``QueryProcessor.find_similar_many`` / ``find_similar_stream`` (the enrichment they batch is
  reference query_processor.py:36-48), object storage and array storage, with and without a host copy of the rows;
* ``Index(devices=[0, 0, 0, 0])``: the single-process row-sharded index as four logical shards on device 0
  ("G logical shards on one device"), merged ids == single index == oracle;
* ``hnswlib_compat.Index``: the eight calls the reference makes (index.py:36-38,56,65,80,111,115) and hnswlib's
  RuntimeError contracts.
"""
import uuid

import numpy as np
import pytest

from mlvectordb_amd import ArrayStorage, Index, InMemoryStorage, QueryProcessor, VectorDTO, hnswlib_compat
from mlvectordb_amd.idtable import mint_uuid4_bytes
from mlvectordb_amd.multi_device import MultiDeviceEngine
from oracle import exact_scan
from oracle.engine import OracleScanEngine
from tests.helpers import SCORE_ATOL, make_case

pytestmark = pytest.mark.gpu


def _same_hits(a, b, values=True):
    assert len(a) == len(b)
    for ha, hb in zip(a, b):
        assert [h["id"] for h in ha] == [h["id"] for h in hb]
        assert [h["metadata"] for h in ha] == [h["metadata"] for h in hb]
        assert np.allclose([h["score"] for h in ha], [h["score"] for h in hb], atol=SCORE_ATOL, rtol=0)
        if values:
            assert all(np.array_equal(x["values"], y["values"]) for x, y in zip(ha, hb))


@pytest.mark.parametrize("n,d,nq,k", [(6000, 128, 40, 10), (40000, 256, 256, 10)])
@pytest.mark.parametrize("keep_host_copy", [True, False])
def test_find_similar_many_on_hip_equals_the_oracle_engine(n, d, nq, k, keep_host_copy):
    rows, qs = make_case(101, n, d, nq, dup=True)
    metas = [{"i": j} for j in range(n)]
    hip = QueryProcessor(ArrayStorage(), Index(space="cosine"))
    ids = hip.upsert_arrays(rows, "ns", metas, keep_host_copy=keep_host_copy)
    cpu = QueryProcessor(ArrayStorage(), Index(space="cosine", engine_factory=OracleScanEngine))
    cpu._index.add_arrays(rows, "ns", ids=ids)
    cpu._storage.write_arrays(ids, "ns", rows, metas)
    want = cpu.find_similar_many(qs, top_k=k, namespace="ns")
    got = hip.find_similar_many(qs, top_k=k, namespace="ns")
    _same_hits(got, want)
    assert got[0][0]["metadata"] == {"i": 1} and abs(got[0][0]["score"] - 1.0) < 1e-6   # qs[0] is row 1
    assert [h["metadata"]["i"] for h in got[0][:3]] == [1, n // 2, n - 1]               # duplicates: ascending label
    streamed = list(hip.find_similar_stream([qs[:nq // 2], qs[nq // 2:]], top_k=k, namespace="ns"))
    _same_hits(streamed[0] + streamed[1], want)
    one = hip.find_similar(VectorDTO(values=qs[3], metadata={}), top_k=k, namespace="ns")
    _same_hits([one], [want[3]])
    hip._index.close()


def test_find_similar_many_object_storage_and_deletes_on_hip():
    rows, qs = make_case(102, 3000, 64, 16)
    hip = QueryProcessor(InMemoryStorage(), Index(space="cosine"))
    cpu = QueryProcessor(InMemoryStorage(), Index(space="cosine", engine_factory=OracleScanEngine))
    dtos = [VectorDTO(values=r, metadata={"i": j}) for j, r in enumerate(rows)]
    hip.upsert_many(dtos, "ns")
    cpu.upsert_many(dtos, "ns")
    key = lambda hits: [[h["metadata"]["i"] for h in hs] for hs in hits]  # ids are minted per processor: compare rows
    assert key(hip.find_similar_many(qs, 8, "ns")) == key(cpu.find_similar_many(qs, 8, "ns"))
    for qp in (hip, cpu):
        top = qp.find_similar_many(qs[:4], 3, "ns")
        qp.delete([h["id"] for hs in top for h in hs], "ns")
    assert key(hip.find_similar_many(qs, 8, "ns")) == key(cpu.find_similar_many(qs, 8, "ns"))
    assert key(hip.find_similar_many(qs, 5, "ns", where=lambda m: m["i"] % 3 == 0)) == \
        key(cpu.find_similar_many(qs, 5, "ns", where=lambda m: m["i"] % 3 == 0))
    hip._index.close()


def test_search_stream_over_four_logical_shards_equals_search_many_and_oracle():
    """MultiDeviceEngine.search_stream on devices=[0, 0, 0, 0] -- shard scans of wave i+1 queued before
    wave i is merged -- returns, wave for wave, what search_many returns and what the oracle says; QueryProcessor.
    find_similar_stream rides on it."""
    n, d, k = 60_000, 256, 10
    rows, _ = make_case(211, n, d, 1, dup=True)
    many = Index(space="cosine", devices=[0, 0, 0, 0], strategy="filter")
    many.add_arrays(rows, "ns")
    rng = np.random.default_rng(5)
    waves = [rng.standard_normal((nq, d)).astype(np.float32) for nq in (40, 3, 256, 17)]
    try:
        streamed = list(many.search_stream(iter(waves), k, "ns", "cosine"))
        for q, got in zip(waves, streamed):
            want = exact_scan.knn(q, rows, k, "cosine")
            assert np.array_equal(got.labels, want[0]) and np.abs((1 - got.scores) - want[1]).max() <= 1e-5
            assert got == many.search_many(q, k, "ns", "cosine")
        assert all(st["strategy_used"] == 2 for st in many._ns["ns"].engine.last_stats())
    finally:
        many.close()


@pytest.mark.parametrize("space,n,d,nq", [("cosine", 50000, 256, 64), ("l2", 9000, 64, 20), ("ip", 40000, 128, 5)])
def test_multi_device_index_as_four_logical_shards_on_device_0(space, n, d, nq):
    rows, qs = make_case(103, n, d, nq, dup=True)       # rows 1, n//2, n-1 identical: they land in different shards
    ids = mint_uuid4_bytes(n)
    many = Index(space=space, devices=[0, 0, 0, 0])
    one = Index(space=space)
    for ix in (many, one):
        ix.add_arrays(rows[: n // 3], "ns", ids=ids[: n // 3])
        ix.add_arrays(rows[n // 3:], "ns", ids=ids[n // 3:])
    eng = many._ns["ns"].engine
    assert isinstance(eng, MultiDeviceEngine) and len(eng.shards) == 4
    assert sorted(m.size for m in eng._l2g)[0] >= n // 4 - 1
    metric = "cosine" if space == "cosine" else "l2"
    k = 10
    got, single = many.search_many(qs, k, "ns", metric), one.search_many(qs, k, "ns", metric)
    want = exact_scan.knn(qs, rows, k, space)
    assert np.array_equal(got.labels, want[0]) and np.array_equal(single.labels, want[0])
    assert got.labels[0, :3].tolist() == [1, n // 2, n - 1]            # cross-shard tie resolved by label
    assert np.array_equal(got.scores, single.scores)
    assert got == single
    if n >= 32768 and d % 64 == 0:   # each 12.5k-row shard would pick the exact scan on its own: force the filter too
        eng.set_strategy("filter")
        assert many.search_many(qs, k, "ns", metric) == single
        assert all(st["strategy_used"] == 2 for st in eng.last_stats())
        eng.set_strategy("auto")
    dead = [uuid.UUID(bytes=bytes(ids[j])) for j in list(want[0][:, 0]) + [n // 2]]
    many.remove(dead, "ns")
    one.remove(dead, "ns")
    assert many.search_many(qs, k, "ns", metric) == one.search_many(qs, k, "ns", metric)
    radius = float(want[1][:, k - 1].mean())
    assert many.range_search_many(qs, radius, "ns", metric, None) == one.range_search_many(qs, radius, "ns", metric, None)
    assert many.compact("ns") and one.compact("ns")
    assert many.search_many(qs, k, "ns", metric) == one.search_many(qs, k, "ns", metric)
    assert np.array_equal(many.fetch_values("ns", np.arange(0, 2000, 7)), one.fetch_values("ns", np.arange(0, 2000, 7)))
    many.close()
    one.close()


def test_hnswlib_compat_drives_the_eight_calls_like_the_reference():
    rows, qs = make_case(104, 500, 32, 6)
    with pytest.raises(RuntimeError, match="Space name"):
        hnswlib_compat.Index(space="manhattan", dim=32)
    for space in ("l2", "cosine", "ip"):
        index = hnswlib_compat.Index(space=space, dim=32)                       # index.py:36
        with pytest.raises(RuntimeError):
            index.get_current_count()                                           # before init_index
        index.init_index(max_elements=10_000, ef_construction=200, M=16)        # index.py:37
        index.set_ef(50)                                                        # index.py:38
        assert index.get_current_count() == 0                                   # index.py:56
        index.add_items(np.array(rows[:300], dtype=np.float32), list(range(300)))   # index.py:65
        start = index.get_current_count()
        index.add_items(np.array(rows[300:], dtype=np.float32), list(range(start, start + 200)))
        assert index.get_current_count() == 500
        with pytest.raises(RuntimeError, match="dimensionality"):
            index.add_items(np.ones((2, 31), np.float32), [500, 501])
        with pytest.raises(RuntimeError, match="labels"):
            index.add_items(np.ones((1, 32), np.float32), [777])
        for label in (0, 17, 499):
            index.mark_deleted(label)                                           # index.py:80
        with pytest.raises(RuntimeError, match="already deleted"):
            index.mark_deleted(17)
        deleted = np.zeros(500, bool)
        deleted[[0, 17, 499]] = True
        for q in qs:                                                            # index.py:111: (1, d) float32, k
            labels, distances = index.knn_query(np.asarray(q, dtype=np.float32).reshape(1, -1), k=5)
            wl, wd, _ = exact_scan.knn(q[None, :], rows, 5, space, deleted=deleted)
            assert labels.dtype == np.uint64 and distances.dtype == np.float32 and labels.shape == (1, 5)
            assert np.array_equal(labels.astype(np.int64), wl) and np.abs(distances - wd).max() <= SCORE_ATOL
        with pytest.raises(RuntimeError, match="dimensionality"):
            index.knn_query(np.ones((1, 31), np.float32), k=1)
        with pytest.raises(RuntimeError, match="contiguous 2D array"):            # cannot fill k: index.py:112-119 relies on it
            index.knn_query(qs[:1], k=498)
        labels, _ = index.knn_query(qs[:1], k=497)
        assert labels.shape == (1, 497)
        index.close()
