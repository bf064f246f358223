"""Host-side logic of Index / QueryProcessor with the oracle engine injected (CPU only).

Covers what the reference's Python owns (index.py:50-165, query_processor.py:26-62) and the quirks
recorded in SURVEY.md section 3.5.
"""
import numpy as np
import pytest

from mlvectordb_amd import Index, InMemoryStorage, QueryProcessor, Vector, VectorDTO
from mlvectordb_amd.index import SearchResult
from oracle.engine import OracleScanEngine


def idx(space="l2", **kw):
    return Index(space=space, engine_factory=OracleScanEngine, **kw)


def vecs(n, d=8, seed=0):
    rng = np.random.default_rng(seed)
    return [Vector(values=rng.standard_normal(d).tolist(), metadata={"i": i}) for i in range(n)]


def test_constructor_signature_matches_reference():
    i = Index.__new__(Index)
    Index.__init__(i, "cosine", 200, 16, 0.2, engine_factory=OracleScanEngine)  # positional, as index.py:18
    assert i._space == "cosine" and i._M == 16 and i._ef_construction == 200


def test_empty_and_unknown_namespace_return_empty_lists():
    i = idx()
    q = VectorDTO(values=[0.0] * 8, metadata={})
    assert i.search(q, 3, "nope", "l2") == []
    i.add([], "ns")  # no-op, does not create the namespace (index.py:52-53)
    assert i.search(q, 3, "ns", "l2") == []
    v = vecs(2)
    i.add(v, "ns")
    i.remove([x.id for x in v], "ns")
    assert i.search(q, 3, "ns", "l2") == []  # active count 0 (index.py:103-105)


def test_metric_argument_only_flips_the_score_q1_q2():
    v = vecs(20)
    i = idx("l2")
    i.add(v, "ns")
    q = VectorDTO(values=v[3].values, metadata={})
    as_l2 = i.search(q, 5, "ns", "l2")
    as_cos = i.search(q, 5, "ns", "cosine")
    assert [r.vector_id for r in as_l2] == [r.vector_id for r in as_cos]  # same space searched
    assert [r.score for r in as_cos] == [1 - r.score for r in as_l2]  # only the flip differs


def test_query_coercion_and_wrong_dim_query_returns_empty():
    v = vecs(10)
    i = idx()
    i.add(v, "ns")
    as_list = i.search(VectorDTO(values=list(map(float, v[0].values)), metadata={}), 1, "ns", "l2")
    as_f64 = i.search(VectorDTO(values=v[0].values.astype(np.float64), metadata={}), 1, "ns", "l2")
    assert as_list[0].vector_id == as_f64[0].vector_id == v[0].id
    assert i.search(VectorDTO(values=[1.0, 2.0], metadata={}), 1, "ns", "l2") == []  # swallowed RuntimeError


def test_wrong_dim_rows_raise_runtime_error():
    i = idx()
    i.add(vecs(3, d=8), "ns")
    with pytest.raises(RuntimeError):
        i.add(vecs(1, d=5), "ns")


def test_labels_continue_and_are_never_reused_until_rebuild():
    i = idx()
    a, b = vecs(5, seed=1), vecs(5, seed=2)
    i.add(a, "ns")
    i.remove([a[0].id], "ns")
    i.add(b, "ns")
    ns = i._ns["ns"]
    assert ns.ids.lookup([b[0].id])[0] == 5 and ns.total == 10 and ns.deleted == 1


def test_rebuild_threshold_and_flag():
    i = idx(rebuild_threshold=0.2)
    v = vecs(10)
    i.add(v, "ns")
    i.remove([v[0].id], "ns")
    assert not i.is_rebuild_required("ns")  # 1/10 < 0.2
    i.remove([v[1].id, v[1].id, v[2].id], "ns")  # duplicates / repeats are ignored after the first
    assert i.namespace_counts("ns") == (10, 3)
    assert i.is_rebuild_required("ns")
    assert not i.is_rebuild_required("other")


def test_rebuild_replaces_everything_and_uses_metric_as_space():
    i = idx("l2")
    a, b = vecs(4, seed=1), vecs(4, seed=2)
    i.add(a, "A")
    i.add(b, "B")
    i.rebuild({"A": a[:2]}, metric="cosine")
    assert i.search(VectorDTO(values=b[0].values, metadata={}), 1, "B", "l2") == []  # B wiped (index.py:136-143)
    assert i._ns["A"].engine.space == "cosine" and i._space == "l2"  # _space itself is untouched
    assert i._ns["A"].ids.lookup([a[1].id])[0] == 1


def test_query_processor_delete_rebuilds_all_namespaces_by_default_q4():
    qp = QueryProcessor(InMemoryStorage(), idx("cosine"))
    qp.upsert_many([VectorDTO([1, 0], {"l": "a1"}), VectorDTO([0, 1], {"l": "a2"})], namespace="A")
    qp.upsert_many([VectorDTO([1, 1], {"l": "b1"})], namespace="B")
    victim = qp.find_similar(VectorDTO([1, 0], {}), 1, namespace="A")[0]["id"]
    qp.delete([victim], namespace="A")
    assert len(qp.find_similar(VectorDTO([1, 1], {}), 1, namespace="B")) == 1  # divergence from the reference
    qp2 = QueryProcessor(InMemoryStorage(), idx("cosine"), rebuild_scope="namespace")
    qp2.upsert_many([VectorDTO([1, 0], {}), VectorDTO([0, 1], {})], namespace="A")
    qp2.upsert_many([VectorDTO([1, 1], {})], namespace="B")
    victim = qp2.find_similar(VectorDTO([1, 0], {}), 1, namespace="A")[0]["id"]
    qp2.delete([victim], namespace="A")
    assert qp2.find_similar(VectorDTO([1, 1], {}), 1, namespace="B") == []  # reference behaviour reproduced


def test_search_many_equals_repeated_search_and_find_similar_many():
    v = vecs(50, d=12)
    i = idx("cosine")
    i.add(v, "ns")
    rng = np.random.default_rng(5)
    qs = rng.standard_normal((7, 12)).astype(np.float32)
    batched = i.search_many(qs, 4, "ns", "cosine")
    single = [i.search(VectorDTO(values=q, metadata={}), 4, "ns", "cosine") for q in qs]
    assert batched == single
    assert all(isinstance(r, SearchResult) for hits in batched for r in hits)
    qp = QueryProcessor(InMemoryStorage(), idx("cosine"))
    qp.upsert_many([VectorDTO(values=x.values, metadata={"i": n}) for n, x in enumerate(v)])
    many = qp.find_similar_many(qs, top_k=3)
    one = [qp.find_similar(VectorDTO(values=q, metadata={}), top_k=3) for q in qs]
    assert [[h["id"] for h in hits] for hits in many] == [[h["id"] for h in hits] for hits in one]


def test_find_similar_drops_ids_missing_from_storage():
    qp = QueryProcessor(InMemoryStorage(), idx("cosine"))
    qp.upsert_many([VectorDTO([1, 0], {"l": "a"}), VectorDTO([0.9, 0.1], {"l": "b"})])
    top = qp.find_similar(VectorDTO([1, 0], {}), 2)
    qp._storage.delete(top[0]["id"], "default")  # storage and index now disagree
    assert [h["metadata"]["l"] for h in qp.find_similar(VectorDTO([1, 0], {}), 2)] == ["b"]


def test_range_search_and_euclidean_alias():
    i = idx("euclidean")
    v = [Vector(values=[0, 0]), Vector(values=[3, 4]), Vector(values=[6, 8])]
    i.add(v, "ns")
    hits = i.range_search(VectorDTO([0, 0], {}), 5.0, "ns", "euclidean")
    assert [h.vector_id for h in hits] == [v[0].id, v[1].id]
    assert [h.score for h in hits] == [0.0, 5.0]
    assert [h.score for h in i.search(VectorDTO([0, 0], {}), 3, "ns", "euclidean")] == [0.0, 5.0, 10.0]
    assert [h.score for h in i.search(VectorDTO([0, 0], {}), 3, "ns", "l2")] == [0.0, 25.0, 100.0]


def test_compact_equals_rebuild_from_the_survivors():
    """Index.compact (device-side in the product) leaves what a rebuild from the surviving vectors leaves:
    labels renumbered from 0 in insertion order, tombstones and the rebuild flag gone, other namespaces intact."""
    v = vecs(30, d=6, seed=5)
    other = vecs(4, d=6, seed=6)
    a, b = idx("l2", rebuild_threshold=0.2), idx("l2", rebuild_threshold=0.2)
    for i in (a, b):
        i.add(v, "ns")
        i.add(other, "other")
        i.remove([v[k].id for k in (0, 3, 4, 11, 17, 29)], "ns")
        assert i.is_rebuild_required("ns")
    survivors = [x for k, x in enumerate(v) if k not in (0, 3, 4, 11, 17, 29)]
    assert a.compact("ns") and not a.compact("unknown")
    b.rebuild({"ns": survivors, "other": other}, metric="l2")
    assert a.namespace_counts("ns") == b.namespace_counts("ns") == (24, 0)
    assert not a.is_rebuild_required("ns")
    assert np.array_equal(a._ns["ns"].ids.raw[:24], b._ns["ns"].ids.raw[:24])
    assert a._ns["ns"].ids.lookup([x.id for x in survivors]).tolist() == list(range(24))
    rng = np.random.default_rng(9)
    for _ in range(5):
        q = VectorDTO(values=rng.standard_normal(6).tolist(), metadata={})
        assert a.search(q, 7, "ns", "l2") == b.search(q, 7, "ns", "l2")
        assert a.search(q, 2, "other", "l2") == b.search(q, 2, "other", "l2")
    more = vecs(3, d=6, seed=7)  # labels continue after the compacted rows
    a.add(more, "ns")
    assert a._ns["ns"].ids.lookup([m.id for m in more]).tolist() == [24, 25, 26]


def test_query_processor_delete_compacts_instead_of_rebuilding():
    calls = []

    class Spy(Index):
        def rebuild(self, source, metric):
            calls.append("rebuild")
            return super().rebuild(source, metric)

        def compact(self, namespace):
            calls.append("compact")
            return super().compact(namespace)

    qp = QueryProcessor(InMemoryStorage(), Spy(space="cosine", engine_factory=OracleScanEngine))
    qp.upsert_many([VectorDTO([1, 0], {"l": "a1"}), VectorDTO([0, 1], {"l": "a2"}), VectorDTO([1, 1], {"l": "a3"})], namespace="A")
    victim = qp.find_similar(VectorDTO([1, 0], {}), 1, namespace="A")[0]["id"]
    assert qp.delete([victim], namespace="A") == [victim]      # 1/3 >= 0.2: the rebuild trigger fires
    assert calls == ["compact"]
    hits = qp.find_similar(VectorDTO([1, 0], {}), 3, namespace="A")
    assert [h["metadata"]["l"] for h in hits] == ["a3", "a2"]


def test_metadata_filtered_search_is_exact_among_the_matching_vectors():
    rng = np.random.default_rng(21)
    qp = QueryProcessor(InMemoryStorage(), idx("cosine"))
    rows = rng.standard_normal((200, 10)).astype(np.float32)
    qp.upsert_many([VectorDTO(r.tolist(), {"colour": "red" if i % 3 == 0 else "blue", "i": i}) for i, r in enumerate(rows)])
    q = VectorDTO(rng.standard_normal(10).tolist(), {})
    red = qp.find_similar_where(q, 5, where=lambda m: m["colour"] == "red")
    assert len(red) == 5 and all(h["metadata"]["colour"] == "red" for h in red)
    cos = rows @ np.asarray(q.values, np.float32) / np.linalg.norm(rows, axis=1) / np.linalg.norm(q.values)
    want = [i for i in np.argsort(-cos) if i % 3 == 0][:5]
    assert [h["metadata"]["i"] for h in red] == want                       # not a post-filter of the global top-5
    assert qp.find_similar_where(q, 5, where=lambda m: False) == []
    few = qp.find_similar_where(q, 50, where=lambda m: m["i"] < 7)        # k clamps to the matching count
    assert sorted(h["metadata"]["i"] for h in few) == list(range(7))
    qp.delete([red[0]["id"]])
    again = qp.find_similar_where(q, 5, where=lambda m: m["colour"] == "red")
    assert red[0]["id"] not in [h["id"] for h in again] and len(again) == 5


# ---- persistence (README.md:240-241 names save_index/load_index only: parity unpinned)
def _results(i, qs, ns, metric, k=5):
    return [[(r.vector_id, r.score) for r in hits] for hits in i.search_many(qs, k, ns, metric)]


def test_save_index_load_index_round_trip(tmp_path, monkeypatch):
    monkeypatch.setattr(Index, "_CHUNK_BYTES", 8 * 4 * 7)  # 7 rows per chunk: the streamed path
    i = idx("cosine", rebuild_threshold=0.3)
    a, b = vecs(40, seed=1), vecs(25, d=5, seed=2)
    i.add(a, "a")
    i.add(b, "b")
    i.remove([a[3].id, a[17].id, a[39].id], "a")
    i.remove([v.id for v in b[:10]], "b")  # 0.4 >= 0.3: flag raised
    qa = np.random.default_rng(5).standard_normal((4, 8)).astype(np.float32)
    qb = np.random.default_rng(6).standard_normal((3, 5)).astype(np.float32)
    want_a, want_b = _results(i, qa, "a", "cosine"), _results(i, qb, "b", "cosine")
    assert i.save_index(str(tmp_path / "snap")) is True

    j = idx("l2")  # space, threshold and contents all come from the snapshot
    j.add(vecs(3, seed=9), "stale")
    assert j.load_index(str(tmp_path / "snap")) is True
    assert j._space == "cosine" and j._rebuild_threshold == 0.3
    assert j.search(VectorDTO(values=[0.0] * 8, metadata={}), 1, "stale", "cosine") == []
    assert _results(j, qa, "a", "cosine") == want_a and _results(j, qb, "b", "cosine") == want_b
    assert j.namespace_counts("a") == (40, 3) and j.namespace_counts("b") == (25, 10)
    assert j.is_rebuild_required("b") and not j.is_rebuild_required("a")
    # the loaded index keeps working: labels continue, removed ids stay unknown, compaction still works
    more = vecs(5, seed=3)
    j.add(more, "a")
    assert j.namespace_counts("a") == (45, 3)
    hit = j.search(VectorDTO(values=more[2].values, metadata={}), 1, "a", "cosine")[0]
    assert hit.vector_id == more[2].id
    j.remove([a[3].id], "a")  # already gone: no state change (index.py:76-78)
    assert j.namespace_counts("a") == (45, 3)
    assert j.compact("b") and j.namespace_counts("b") == (15, 0)
    assert _results(j, qb, "b", "cosine") == want_b


def test_load_index_missing_or_inconsistent_snapshot(tmp_path):
    i = idx("l2")
    i.add(vecs(6), "a")
    assert i.load_index(str(tmp_path / "absent")) is False
    assert i.namespace_counts("a") == (6, 0)  # untouched
    i.save_index(str(tmp_path / "snap"))
    with open(tmp_path / "snap" / "ns0.rows.f32", "ab") as f:
        f.write(b"\0" * 4)
    with pytest.raises(RuntimeError):
        i.load_index(str(tmp_path / "snap"))
    assert i.namespace_counts("a") == (6, 0)  # validation happens before anything is dropped
    empty = idx("ip")
    assert empty.save_index(str(tmp_path / "empty")) and idx("l2").load_index(str(tmp_path / "empty"))


# ---- nothing is mutated by a batch the index refuses; rebuild never destroys the only copy of the rows
def _oracle_qp(scope="all"):
    from mlvectordb_amd import ArrayStorage

    index = Index(space="l2", engine_factory=OracleScanEngine)
    return QueryProcessor(ArrayStorage(), index, rebuild_scope=scope), index


def test_a_refused_batch_leaves_the_namespace_exactly_as_it_was():
    from mlvectordb_amd import SimpleVector

    index = Index(space="l2", engine_factory=OracleScanEngine)
    good = [Vector(values=[float(i), 1.0, 2.0]) for i in range(4)]
    index.add(good, "ns")
    with pytest.raises(RuntimeError, match="uuid.UUID"):  # a non-UUID id is refused before the rows reach the engine
        index.add([Vector(values=[9.0, 9.0, 9.0]), SimpleVector("abc", [1.0, 2.0, 3.0])], "ns")
    with pytest.raises(RuntimeError, match="non-finite"):
        index.add([Vector(values=[np.nan, 0.0, 0.0])], "ns")
    with pytest.raises(RuntimeError, match="handles"):
        index.add_arrays(np.ones((2, 3), np.float32), "ns", handles=np.arange(3))
    with pytest.raises(RuntimeError):  # a refused first batch does not even create the namespace
        index.add([SimpleVector("x", [1.0, 2.0])], "fresh")
    assert index.namespace_counts("fresh") == (0, 0) and "fresh" not in index._ns
    assert index.namespace_counts("ns") == (4, 0) and index._ns["ns"].engine.counts() == (4, 0)
    index.add([Vector(values=[5.0, 5.0, 5.0])], "ns")  # engine and host row counts still agree: adds keep working
    assert index.namespace_counts("ns") == (5, 0)


def test_upsert_arrays_validates_before_the_storage_is_written():
    qp, index = _oracle_qp()
    qp.upsert_arrays(np.ones((3, 4), np.float32), "ns")
    bad = np.ones((2, 4), np.float32)
    bad[1, 2] = np.inf
    with pytest.raises(RuntimeError, match="non-finite"):
        qp.upsert_arrays(bad, "ns")
    with pytest.raises(RuntimeError, match="dimensionality"):
        qp.upsert_arrays(np.ones((2, 5), np.float32), "ns")
    assert qp.get_namespace_count("ns") == 3 and index.namespace_counts("ns") == (3, 0)  # no ghost rows in the storage


def test_rebuild_stages_its_source_before_it_closes_anything():
    index = Index(space="l2", engine_factory=OracleScanEngine)
    rows = [Vector(values=[float(i), 0.0]) for i in range(5)]
    index.add(rows, "a")

    class NoValues:
        id, values, metadata = rows[0].id, None, {}

    for bad_source in ({"a": [NoValues()]}, {"a": rows, "b": [Vector(values=[np.nan, 1.0])]},
                       {"a": [Vector(values=[1.0, 2.0]), Vector(values=[1.0, 2.0, 3.0])]}):
        with pytest.raises(RuntimeError):
            index.rebuild(bad_source, metric="l2")
        assert index.namespace_counts("a") == (5, 0)  # untouched
        assert index.search(VectorDTO(values=[3.0, 0.0], metadata={}), 1, "a", "l2")[0].vector_id == rows[3].id
    with pytest.raises(RuntimeError, match="Space name"):
        index.rebuild({"a": rows}, metric="manhattan")
    assert index.namespace_counts("a") == (5, 0)


def test_delete_with_rows_kept_in_hbm_only_survives_the_rebuild_path():
    """ArrayStorage without a host copy of the values + an index without compact(): the rebuild source is read back
    from the index before its engines are closed (round 2: IndexError after the rows were already gone)."""
    qp, index = _oracle_qp()
    index.compact = None  # force the rebuild-from-storage path (what an index without device compaction takes)
    vals = np.arange(40, dtype=np.float32).reshape(10, 4)
    ids = qp.upsert_arrays(vals, "ns", keep_host_copy=False)
    other = qp.upsert_arrays(vals[:3] + 100.0, "other", keep_host_copy=False)
    from uuid import UUID

    victims = [UUID(bytes=ids[i].tobytes()) for i in (0, 1, 2)]  # 3/10 >= 0.2: the trigger fires
    assert list(qp.delete(victims, "ns")) == victims
    assert index.namespace_counts("ns") == (7, 0) and index.namespace_counts("other") == (3, 0)
    hit = qp.find_similar(VectorDTO(values=vals[5], metadata={}), top_k=1, namespace="ns", metric="l2")[0]
    assert hit["id"] == UUID(bytes=ids[5].tobytes()) and np.array_equal(hit["values"], vals[5])
    hit = qp.find_similar(VectorDTO(values=vals[1] + 100.0, metadata={}), top_k=1, namespace="other", metric="l2")[0]
    assert hit["id"] == UUID(bytes=other[1].tobytes())


def test_namespace_scoped_rebuild_refuses_rows_kept_in_hbm_only():
    qp, _ = _oracle_qp(scope="namespace")
    with pytest.raises(ValueError, match="rebuild_scope"):
        qp.upsert_arrays(np.ones((2, 4), np.float32), "ns", keep_host_copy=False)
    qp.upsert_arrays(np.ones((2, 4), np.float32), "ns")  # with a host copy the reference's Q4 behaviour is available


def test_remove_of_an_id_added_twice_is_a_no_op_the_second_time():
    index = Index(space="l2", engine_factory=OracleScanEngine, rebuild_threshold=2.0)
    v = Vector(values=[1.0, 0.0])
    index.add([v, Vector(values=[0.0, 1.0]), v], "ns")
    index.remove([v.id], "ns")
    assert index.namespace_counts("ns") == (3, 1)
    index.remove([v.id], "ns")  # reference: dict.pop already dropped the id (index.py:76-81)
    assert index.namespace_counts("ns") == (3, 1)
