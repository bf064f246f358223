"""Array-backed host structures (CPU, oracle engine injected): IdTable, BatchHits, ArrayStorage, the batched
enrichment of QueryProcessor, and the single-process multi-device Index (as logical shards over oracle engines).

The generic object path (InMemoryStorage + hit-by-hit ``_enrich``, the shape of reference query_processor.py:33-49)
is the yardstick: every fast path has to return exactly what it returns.
"""
import uuid

import numpy as np
import pytest

from mlvectordb_amd import ArrayStorage, Index, InMemoryStorage, QueryProcessor, Vector, VectorDTO
from mlvectordb_amd.idtable import IdTable, mint_uuid4_bytes
from mlvectordb_amd.index import BatchHits, SearchResult
from mlvectordb_amd.multi_device import MultiDeviceEngine
from oracle import exact_scan
from oracle.engine import OracleScanEngine


def idx(space="cosine", **kw):
    return Index(space=space, engine_factory=OracleScanEngine, **kw)


# ------------------------------------------------------------------------------------------------ IdTable
def test_idtable_matches_the_references_two_dicts():
    rng = np.random.default_rng(3)
    t = IdTable()
    u2l, l2u = {}, {}          # what reference index.py:21-22,56-63,76-81 maintains
    ids = []
    for round_ in range(6):
        batch = [uuid.uuid4() for _ in range(int(rng.integers(1, 40)))]
        first = t.append_uuids(batch) if round_ % 2 == 0 else t.append_raw(np.array([list(u.bytes) for u in batch], np.uint8))
        for i, u in enumerate(batch):
            u2l[u] = first + i
            l2u[first + i] = u
        ids += batch
        victims = [ids[j] for j in rng.choice(len(ids), size=3, replace=False)] + [uuid.uuid4(), "not-a-uuid"]
        labels = t.lookup(victims)
        want = [u2l.get(v, -1) if isinstance(v, uuid.UUID) else -1 for v in victims]
        assert labels.tolist() == want
        t.kill(labels[labels >= 0])
        for v in victims:
            lab = u2l.pop(v, None) if isinstance(v, uuid.UUID) else None
            if lab is not None:
                l2u.pop(lab)
    assert t.n == len(ids)
    got = t.uuids_at(np.arange(-1, t.n + 1))
    assert got[0] is None and got[-1] is None
    assert {i: u for i, u in enumerate(got[1:-1].tolist()) if u is not None} == l2u
    assert sorted(t.dead_labels().tolist()) == sorted(set(range(t.n)) - set(l2u))


def test_idtable_same_id_twice_newest_row_wins_and_minted_ids_are_uuid4():
    u = uuid.uuid4()
    t = IdTable()
    t.append_uuids([u, uuid.uuid4(), u])
    assert t.lookup([u]).tolist() == [2]       # the dict entry was overwritten (index.py:62)
    t.kill([2])
    assert t.lookup([u]).tolist() == [-1]      # dict.pop removed the id's one entry (index.py:76-81): the old row is unreachable
    seq = IdTable()                            # ids that share their first 8 bytes still get distinct sorted keys
    seq.append_uuids([uuid.UUID(int=i) for i in range(1, 400)])
    assert seq.lookup([uuid.UUID(int=7), uuid.UUID(int=399), uuid.UUID(int=400)]).tolist() == [6, 398, -1]
    seq._extend_index()
    assert np.unique(seq._keys).size == 399
    raw = mint_uuid4_bytes(100)
    assert len({bytes(r) for r in raw}) == 100
    assert all(uuid.UUID(bytes=bytes(r)).version == 4 and uuid.UUID(bytes=bytes(r)).variant == uuid.RFC_4122 for r in raw)
    k = IdTable()
    k.append_raw(raw[:60])
    k.lookup_raw(raw[:3])                         # builds the sorted index ...
    k.append_raw(raw[60:])                        # ... which must then be extended by merging
    assert k.lookup_raw(raw[[99, 0, 60, 59]]).tolist() == [99, 0, 60, 59]
    kept = k.take(np.array([1, 5, 99]))
    assert kept.n == 3 and kept.lookup_raw(raw[[5, 2]]).tolist() == [1, -1]


# ------------------------------------------------------------------------------------------------ BatchHits
def test_batch_hits_is_a_lazy_list_of_lists():
    rows = np.random.default_rng(1).standard_normal((30, 6)).astype(np.float32)
    i = idx("l2")
    ids = i.add_arrays(rows, "ns")
    hits = i.search_many(rows[:4], 3, "ns", "l2")
    assert isinstance(hits, BatchHits) and len(hits) == 4 and hits.labels.shape == (4, 3)
    assert not hits._rows                                   # nothing materialised yet
    first = hits[0]
    assert isinstance(first[0], SearchResult) and first[0].vector_id == uuid.UUID(bytes=bytes(ids[0])) and first[0].score == 0.0
    assert isinstance(first[0].score, float) and list(hits._rows) == [0]
    assert hits[-1] == hits[3] and [len(h) for h in hits] == [3] * 4 and hits[1:3] == [hits[1], hits[2]]
    assert np.array_equal(hits.id_bytes()[:, 0], ids[:4])
    assert hits == [hits[j] for j in range(4)]
    with pytest.raises(IndexError):
        hits[4]
    empty = i.search_many(rows[:2], 3, "unknown", "l2")
    assert len(empty) == 2 and empty[0] == [] and empty == [[], []]


def test_add_arrays_equals_add_of_vector_objects():
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((40, 8)).astype(np.float32)
    vs = [Vector(r) for r in rows]
    a, b = idx(), idx()
    a.add(vs, "ns")
    b.add_arrays(rows, "ns", ids=np.array([list(v.id.bytes) for v in vs], np.uint8))
    qs = rng.standard_normal((5, 8)).astype(np.float32)
    assert a.search_many(qs, 6, "ns", "cosine") == b.search_many(qs, 6, "ns", "cosine")
    a.remove([vs[3].id, vs[7].id], "ns")
    b.remove([vs[3].id, vs[7].id], "ns")
    assert a.search_many(qs, 40, "ns", "cosine") == b.search_many(qs, 40, "ns", "cosine")
    assert a.namespace_counts("ns") == b.namespace_counts("ns") == (40, 2)
    with pytest.raises(RuntimeError):
        b.add_arrays(rows[:, :5], "ns")
    with pytest.raises(RuntimeError):
        b.add_arrays(rows[:3], "ns", ids=mint_uuid4_bytes(2))


def test_non_finite_rows_are_refused_not_silently_tombstoned():
    i = idx()
    rows = np.ones((4, 3), np.float32)
    rows[2, 1] = np.nan
    with pytest.raises(RuntimeError, match="row 2"):
        i.add_arrays(rows, "ns")
    with pytest.raises(RuntimeError, match="non-finite"):
        i.add([Vector([1.0, np.inf, 0.0])], "ns2")
    assert i.namespace_counts("ns") == (0, 0)


def test_top_k_clamps_and_range_max_results_truncates():
    rows = np.random.default_rng(4).standard_normal((50, 4)).astype(np.float32)
    i = idx("l2")
    i.add_arrays(rows, "ns")
    q = VectorDTO(values=rows[0], metadata={})
    assert len(i.search(q, 10**9, "ns", "l2")) == 50          # clamps (index.py:107), never raises
    everything = i.range_search(q, 1e9, "ns", "l2", max_results=None)
    assert len(everything) == 50
    some = i.range_search(q, 1e9, "ns", "l2", max_results=7)
    assert some == everything[:7]                               # a cap: the nearest max_results
    assert i.range_search(q, 1e9, "ns", "l2") == everything    # default 1024 > 50


# ------------------------------------------------------------------------------------------------ ArrayStorage + QueryProcessor
def _twin_processors(n=120, d=10, seed=7, keep_host_copy=True):
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    metas = [{"i": j, "even": j % 2 == 0} for j in range(n)]
    fast = QueryProcessor(ArrayStorage(), idx())
    ids = fast.upsert_arrays(rows, "ns", metas, keep_host_copy=keep_host_copy)
    slow = QueryProcessor(InMemoryStorage(), idx())
    vs = [Vector(r, m) for r, m in zip(rows, metas)]
    for v, raw in zip(vs, ids):
        v._id = uuid.UUID(bytes=bytes(raw))
    slow._storage.write_vectors(vs, "ns")
    slow._index.add(vs, "ns")
    return fast, slow, rows, rng


def _same(a, b):
    assert len(a) == len(b)
    for ha, hb in zip(a, b):
        assert [h["id"] for h in ha] == [h["id"] for h in hb]
        assert [h["score"] for h in ha] == [h["score"] for h in hb]
        assert [h["metadata"] for h in ha] == [h["metadata"] for h in hb]
        assert all(np.array_equal(x["values"], y["values"]) and x["values"].dtype == np.float32 for x, y in zip(ha, hb))
        assert all(list(h) == ["id", "values", "metadata", "score"] for h in ha)   # query_processor.py:42-47 key order


@pytest.mark.parametrize("keep_host_copy", [True, False])
def test_batched_enrichment_equals_the_hit_by_hit_path(keep_host_copy):
    fast, slow, rows, rng = _twin_processors(keep_host_copy=keep_host_copy)
    qs = rng.standard_normal((9, rows.shape[1])).astype(np.float32)
    _same(fast.find_similar_many(qs, top_k=5, namespace="ns"), slow.find_similar_many(qs, top_k=5, namespace="ns"))
    one = fast.find_similar(VectorDTO(values=qs[0], metadata={}), top_k=5, namespace="ns")
    _same([one], [slow.find_similar(VectorDTO(values=qs[0], metadata={}), top_k=5, namespace="ns")])
    assert isinstance(one[0]["id"], uuid.UUID) and isinstance(one[0]["score"], float)
    # storage and index disagree: hits missing from the storage are dropped silently (query_processor.py:40-41)
    victim = one[0]["id"]
    assert fast._storage.delete(victim, "ns") and slow._storage.delete(victim, "ns")
    _same(fast.find_similar_many(qs, top_k=5, namespace="ns"), slow.find_similar_many(qs, top_k=5, namespace="ns"))
    assert victim not in [h["id"] for h in fast.find_similar_many(qs[:1], top_k=5, namespace="ns")[0]]
    # delete through the processor, metadata filter, unknown namespace
    gone = [h["id"] for h in one[1:3]]
    assert list(fast.delete(gone, "ns")) == list(slow.delete(gone, "ns")) == gone
    _same(fast.find_similar_many(qs, top_k=200, namespace="ns"), slow.find_similar_many(qs, top_k=200, namespace="ns"))
    _same(fast.find_similar_many(qs, 4, "ns", where=lambda m: m["even"]), slow.find_similar_many(qs, 4, "ns", where=lambda m: m["even"]))
    assert fast.find_similar_many(qs[:2], 3, "nowhere") == [[], []]
    near_f = fast.find_in_radius(VectorDTO(values=qs[0], metadata={}), 0.9, "ns", "cosine", max_results=6)
    near_s = slow.find_in_radius(VectorDTO(values=qs[0], metadata={}), 0.9, "ns", "cosine", max_results=6)
    _same([near_f], [near_s])
    assert 4 <= len(near_f) <= 6   # max_results caps the index hits; ids missing from the storage are dropped after that
    info = fast.get_storage_info()
    assert info["vectors_per_namespace"] == {"ns": rows.shape[0] - 3} == slow.get_storage_info()["vectors_per_namespace"]
    assert info["namespace_count"] == 1 and info["total_vectors"] == rows.shape[0] - 3
    assert (info["storage_size_bytes"] > 0) == keep_host_copy


def test_find_similar_stream_pipelines_without_changing_results():
    fast, slow, rows, rng = _twin_processors()
    batches = [rng.standard_normal((b, rows.shape[1])).astype(np.float32) for b in (3, 1, 5, 2)]
    got = list(fast.find_similar_stream(iter(batches), top_k=4, namespace="ns"))
    assert len(got) == 4
    for g, b in zip(got, batches):
        _same(g, slow.find_similar_many(b, top_k=4, namespace="ns"))
    assert list(fast.find_similar_stream([], top_k=4)) == []


def test_array_storage_speaks_the_object_surface_too():
    st = ArrayStorage()
    vs = [Vector([float(j), 1.0], {"j": j}) for j in range(5)]
    assert st.write_vectors(vs[:3], "a") == [True] * 3 and st.write(vs[3], "a") and st.write(vs[4], "b")
    got = st.read_vectors([vs[1].id, uuid.uuid4(), vs[3].id], "a")
    assert got[1] is None and got[0].id == vs[1].id and got[0].metadata == {"j": 1} and np.array_equal(got[2].values, vs[3].values)
    assert st.total_vectors == 5 and st.list_namespaces == ["a", "b"]
    assert st.delete(vs[0].id, "a") and not st.delete(vs[0].id, "a") and not st.delete(vs[0].id, "zzz")
    assert [r.metadata["j"] for r in st.namespace_map["a"]] == [1, 2, 3]
    qp = QueryProcessor(st, idx("l2"))                     # rebuild-from-storage (query_processor.py:58-61) works on it
    qp._index.add(vs[:4], "a")
    qp._index.remove([vs[0].id], "a")
    qp._index.rebuild({"a": st.namespace_map["a"]}, metric="l2")
    assert [h["metadata"]["j"] for h in qp.find_similar(VectorDTO([3.0, 1.0], {}), 2, "a", "l2")] == [3, 2]


# ------------------------------------------------------------------------------------------------ multi-device Index
def _sharded_and_single(space, n=400, d=16, g=4, seed=11):
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[n // 2] = rows[1]          # exact duplicates in different shards: the tie must go to the lower label
    rows[n - 1] = rows[1]
    many = Index(space=space, devices=[0] * g, engine_factory=OracleScanEngine)
    one = idx(space)
    ids = mint_uuid4_bytes(n)
    for lo, hi in ((0, 150), (150, 151), (151, 400)):   # bulk, single row, bulk
        many.add_arrays(rows[lo:hi], "ns", ids=ids[lo:hi])
        one.add_arrays(rows[lo:hi], "ns", ids=ids[lo:hi])
    assert isinstance(many._ns["ns"].engine, MultiDeviceEngine)
    return many, one, rows, ids, rng


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
def test_multi_device_index_equals_a_single_index(space):
    many, one, rows, ids, rng = _sharded_and_single(space)
    eng = many._ns["ns"].engine
    loads = [m.size for m in eng._l2g]
    assert sum(loads) == 400 and max(loads) - min(loads) <= 1              # appends level the shards
    qs = rng.standard_normal((12, rows.shape[1])).astype(np.float32)
    qs[0] = rows[1]
    metric = "cosine" if space == "cosine" else "l2"
    assert many.search_many(qs, 10, "ns", metric) == one.search_many(qs, 10, "ns", metric)
    got = many.search_many(qs[:1], 3, "ns", metric)
    assert got.labels[0].tolist() == [1, 200, 399]                          # cross-shard duplicates: ascending label
    want = exact_scan.knn(qs, rows, 10, space)
    assert np.array_equal(many.search_many(qs, 10, "ns", metric).labels, want[0])
    # rows come back bit-exact from whichever shard holds them
    assert np.array_equal(many.fetch_values("ns", np.array([399, 0, 150, 151, 7])), rows[[399, 0, 150, 151, 7]])
    assert np.array_equal(eng.get_rows(140, 20), rows[140:160])
    # tombstones, metadata masks, top_k beyond one shard's rows
    dead = [uuid.UUID(bytes=bytes(ids[j])) for j in (1, 5, 150, 200, 333)]
    many.remove(dead, "ns")
    one.remove(dead, "ns")
    assert many.namespace_counts("ns") == one.namespace_counts("ns") == (400, 5)
    assert many.search_many(qs, 150, "ns", metric) == one.search_many(qs, 150, "ns", metric)
    allowed = [uuid.UUID(bytes=bytes(ids[j])) for j in range(0, 400, 3)]
    assert many.search_many(qs, 5, "ns", metric, allowed_ids=allowed) == one.search_many(qs, 5, "ns", metric, allowed_ids=allowed)
    # range queries (merged by fp32 distance, cross-shard ties re-ranked in fp64)
    radius = float(np.median(exact_scan.exact_distances(qs[:1], rows, space)))
    assert many.range_search_many(qs, radius, "ns", metric, None) == one.range_search_many(qs, radius, "ns", metric, None)
    assert many.range_search_many(qs, radius, "ns", metric, 7) == one.range_search_many(qs, radius, "ns", metric, 7)
    # compaction renumbers like a rebuild from the survivors, then appends continue
    assert many.compact("ns") and one.compact("ns")
    assert many.namespace_counts("ns") == (395, 0)
    assert np.array_equal(many._ns["ns"].ids.raw[:395], one._ns["ns"].ids.raw[:395])
    extra = rng.standard_normal((9, rows.shape[1])).astype(np.float32)
    more = mint_uuid4_bytes(9)
    many.add_arrays(extra, "ns", ids=more)
    one.add_arrays(extra, "ns", ids=more)
    assert many.search_many(qs, 20, "ns", metric) == one.search_many(qs, 20, "ns", metric)
    assert np.array_equal(many._ns["ns"].engine.get_rows(390, 14), one._ns["ns"].engine.get_rows(390, 14))
    many.close()


def test_multi_device_save_load_and_query_processor(tmp_path):
    many, one, rows, ids, rng = _sharded_and_single("cosine", n=90, g=3)
    many.save_index(str(tmp_path / "snap"))
    back = Index(space="cosine", devices=[0, 0], engine_factory=OracleScanEngine)   # other shard count on reload
    assert back.load_index(str(tmp_path / "snap"))
    qs = rng.standard_normal((4, rows.shape[1])).astype(np.float32)
    assert back.search_many(qs, 6, "ns", "cosine") == one.search_many(qs, 6, "ns", "cosine")
    qp = QueryProcessor(ArrayStorage(), Index(space="cosine", devices=[0, 0, 0], engine_factory=OracleScanEngine))
    qp.upsert_arrays(rows, "ns", keep_host_copy=False)    # values come back from the shards
    hits = qp.find_similar_many(rows[:3], top_k=2, namespace="ns")
    assert all(np.array_equal(h[0]["values"], rows[j]) and abs(h[0]["score"] - 1.0) < 1e-6 for j, h in enumerate(hits))


@pytest.mark.parametrize("space", ["l2", "cosine"])
def test_search_stream_pipelines_shard_scans_and_merges_without_changing_results(space):
    """Index.search_stream (round 3, SURVEY 8e "overlaps the next query wave"): on a row-sharded namespace the shard scans of
    wave i+1 are queued before wave i is merged; results, wave for wave, are those of search_many -- also for batches that
    cannot be scanned (wrong dimensionality, empty) in the middle of the stream, and on a single engine."""
    import threading

    many, one, rows, ids, rng = _sharded_and_single(space)
    metric = "cosine" if space == "cosine" else "l2"
    d = rows.shape[1]
    batches = [rng.standard_normal((n, d)).astype(np.float32) for n in (5, 1, 12, 7, 3)]
    batches.insert(2, np.zeros((4, d + 1), np.float32))   # wrong dimensionality: [] per query, the stream goes on
    batches.insert(4, np.zeros((0, d), np.float32))        # an empty batch
    want = [one.search_many(q, 6, "ns", metric) for q in batches]
    for index in (many, one):
        got = list(index.search_stream(iter(batches), 6, "ns", metric))
        assert len(got) == len(batches)
        for g, w, q in zip(got, want, batches):
            assert len(g) == q.shape[0] and g == w
    assert [len(h) for h in many.search_stream([batches[0]], 6, "nope", metric)] == [5]   # unknown namespace: empty hit lists
    # the pipeline really runs ahead: while the consumer holds wave 0, wave 1's scans have been issued
    eng = many._ns["ns"].engine
    issued, real = [], eng.shards[0].search64

    def spy(q, k, m=None):
        issued.append(q.shape[0])
        return real(q, k, m)

    eng.shards[0].search64 = spy
    stream = many.search_stream(iter([batches[0], batches[1], batches[3]]), 6, "ns", metric)
    first = next(stream)
    deadline = threading.Event()
    for _ in range(200):  # (the shard's thread runs the queued call on its own)
        if len(issued) >= 2:
            break
        deadline.wait(0.01)
    assert first == want[0] and len(issued) >= 2, issued
    assert [h == w for h, w in zip(stream, (want[1], want[3]))] == [True, True]
    eng.shards[0].search64 = real
    # QueryProcessor.find_similar_stream rides on it
    from mlvectordb_amd import ArrayStorage, QueryProcessor
    from oracle.engine import OracleScanEngine

    qp = QueryProcessor(ArrayStorage(), Index(space=space, devices=[0, 0, 0], engine_factory=OracleScanEngine))
    qp.upsert_arrays(rows, "ns")
    a = list(qp.find_similar_stream(iter(batches[:2]), 4, "ns", metric))
    b = [qp.find_similar_many(q, 4, "ns", metric) for q in batches[:2]]
    assert [[h["id"] for h in hs] for wave in a for hs in wave] == [[h["id"] for h in hs] for wave in b for hs in wave]
    many.close()
    one.close()
