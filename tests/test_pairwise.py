"""SURVEY 8(a) row a8': pairwise distance / similarity / normalize (host arithmetic; parity unpinned -- the
reference names the methods in README.md:30-41,178-181 and implements none).  Checked against the oracle's
distance matrix and against the reference's own hand case and its ``cosine_similarity`` helper
(reference tests/test_query_processor.py:29-33,52-67, restated)."""
import numpy as np
import pytest

from mlvectordb_amd import SimpleVector, Vector, pairwise
from oracle import exact_scan


def ref_cosine_similarity(a, b):  # what reference tests/test_query_processor.py:29-33 computes
    a, b = np.array(a), np.array(b)
    return np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))


HAND = {"A": [1, 0, 0], "B": [0, 1, 0], "C": [0.8, 0.2, 0]}
QUERY = [0.9, 0.1, 0]


def test_hand_case_orders_like_the_reference_expects():
    q = Vector(QUERY)
    sims = {k: q.similarity(Vector(v), "cosine") for k, v in HAND.items()}
    assert sorted(sims, key=sims.get, reverse=True) == ["A", "C", "B"]
    for k, v in HAND.items():
        assert sims[k] == pytest.approx(ref_cosine_similarity(np.float32(QUERY), np.float32(v)), rel=1e-6)
        assert q.distance(Vector(v), "cosine") == pytest.approx(1 - sims[k], abs=1e-12)


@pytest.mark.parametrize("space,metric", [("l2", "l2"), ("cosine", "cosine"), ("ip", "ip")])
def test_distance_equals_the_oracles_matrix_entry(space, metric):
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((50, 37), dtype=np.float32)
    qs = rng.standard_normal((4, 37), dtype=np.float32)
    want = exact_scan.exact_distances(qs, rows, space)
    for i in range(4):
        for j in (0, 7, 49):
            got = Vector(qs[i]).distance(Vector(rows[j]), metric)
            assert got == pytest.approx(want[i, j], rel=1e-12, abs=1e-12)
            assert np.float32(got) == np.float32(want[i, j]) or abs(got - want[i, j]) < 1e-12


def test_euclidean_is_sqrt_of_l2_and_readme_example():
    a = SimpleVector("doc1", np.array([1.0, 2.0, 3.0]), {"category": "text"})
    b = SimpleVector("doc2", np.array([2.0, 3.0, 4.0]), {"category": "text"})
    assert a.distance(b, metric="l2") == 3.0
    assert a.distance(b, metric="euclidean") == pytest.approx(np.sqrt(3.0), abs=1e-15)
    assert a.similarity(b, metric="cosine") == pytest.approx(20 / np.sqrt(14 * 29), abs=1e-12)
    assert a.similarity(b, "ip") == 20.0 and a.similarity(b, "dot") == 20.0
    assert a.similarity(b, "euclidean") == -a.distance(b, "euclidean")
    assert a.distance(a, "l2") == 0.0 and a.distance(a, "euclidean") == 0.0
    assert a.distance(a, "cosine") == pytest.approx(0.0, abs=1e-15)
    assert a.distance(b, "ip") == -19.0


def test_normalize_zero_vector_types_and_errors():
    v = Vector([3.0, 4.0])
    n = v.normalize()
    assert isinstance(n, Vector) and n.values.dtype == np.float32 and n.id != v.id
    assert np.allclose(n.values, [0.6, 0.8], atol=1e-7)
    z = SimpleVector("z", [0.0, 0.0, 0.0]).normalize()
    assert np.array_equal(z.data, np.zeros(3, np.float32))          # + 1e-30 keeps 0/0 out
    assert SimpleVector("z", [0.0, 0.0]).similarity(SimpleVector("y", [1.0, 0.0])) == 0.0
    assert np.array_equal(pairwise.normalize(np.float32([2, 0])), np.float32([1, 0]))
    with pytest.raises(ValueError):
        v.distance(Vector([1.0, 2.0, 3.0]))
    with pytest.raises(ValueError):
        v.distance(v, "manhattan")
    with pytest.raises(ValueError):
        v.similarity(v, "manhattan")


def test_simple_vector_dict_round_trip_and_protocol_shape():
    a = SimpleVector("doc1", [1.0, 2.0], {"k": 1})
    b = SimpleVector.from_dict(a.to_dict())
    assert a == b and b.values is b.data and b.shape() == (2,) and b.dimension == 2
    assert a.distance(np.float32([1.0, 2.0]), "l2") == 0.0  # raw arrays are accepted too
    assert Vector([1.0, 2.0]).distance(a, "l2") == 0.0


# ---- the same arithmetic behind the index: Index.distances (SURVEY 8a row a8' on the engine; HIP under -m gpu)
def _index_with(rows, space, factory):
    from mlvectordb_amd import Index

    index = Index(space=space, **({} if factory is None else {"engine_factory": factory}))
    vs = [Vector(r) for r in rows]
    index.add(vs, "ns")
    return index, vs


def _check_index_distances(space, factory, dim=37):
    from mlvectordb_amd import VectorDTO

    rng = np.random.default_rng(11)
    rows = rng.standard_normal((300, dim), dtype=np.float32)
    rows[17] = rows[3]  # a duplicate row: equal distances
    qs = rng.standard_normal((5, dim), dtype=np.float32)
    index, vs = _index_with(rows, space, factory)
    metric = space
    try:
        # (1) a hit's score IS Index.distances of that pair, exactly (the same fp64 summation, rounded once to fp32)
        hits = index.search_many(qs, 8, "ns", metric)
        ids = [[h.vector_id for h in hits[i]] for i in range(5)]
        got = index.distances(qs, ids, "ns", metric)
        assert got.shape == (5, 8)
        assert np.array_equal(got, np.array([[h.score for h in hits[i]] for i in range(5)]))
        # (2) == the oracle's matrix entry (fp32 rounding), == Vector.distance / similarity of the stored vectors
        want = exact_scan.exact_distances(qs, rows, space)
        pick = np.array([[0, 3, 17, 299], [5, 6, 7, 8], [100, 1, 2, 3], [42, 42, 42, 42], [299, 298, 297, 296]])
        got = index.distances(qs, pick, "ns", metric)  # int64 labels are accepted too
        for i in range(5):
            for j in range(4):
                d = np.float32(want[i, pick[i, j]])
                score = 1.0 - np.float64(d) if metric == "cosine" else np.float64(d)
                assert abs(got[i, j] - score) <= 1e-6 * max(1.0, abs(score))
                pw = (Vector(qs[i]).similarity(vs[pick[i, j]], "cosine") if metric == "cosine"
                      else Vector(qs[i]).distance(vs[pick[i, j]], metric))
                assert abs(got[i, j] - pw) <= 1e-6 * max(1.0, abs(pw))  # the fp32 rounding of the returned score
        assert got[0, 1] == got[0, 2]  # the duplicated row
        # (3) "euclidean" = sqrt of the l2 namespace's score; unknown / removed ids -> NaN
        if space == "l2":
            e = index.distances(qs[:1], [[vs[9].id]], "ns", "euclidean")
            assert abs(e[0, 0] - Vector(qs[0]).distance(vs[9], "euclidean")) <= 1e-5 * max(1.0, e[0, 0])
        index.remove([vs[5].id], "ns")
        import uuid

        out = index.distances(qs[:1], [[vs[5].id, uuid.uuid4(), vs[6].id]], "ns", metric)
        assert np.isnan(out[0, 0]) and np.isnan(out[0, 1]) and np.isfinite(out[0, 2])
        with pytest.raises(RuntimeError):
            index.distances(qs[:, :3], [[vs[0].id]] * 5, "ns", metric)
        # the reference's hand case through the index (tests/test_query_processor.py:52-67 restated)
        one = index.search(VectorDTO(values=qs[0], metadata={}), 1, "ns", metric)[0]
        assert index.distances(qs[:1], [[one.vector_id]], "ns", metric)[0, 0] == one.score
    finally:
        index.close()


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
def test_index_distances_on_the_oracle_engine(space):
    from oracle.engine import OracleScanEngine

    _check_index_distances(space, OracleScanEngine)


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("dim", [37, 768])
def test_index_distances_on_hip_equal_search_scores_pairwise_and_oracle(space, dim):
    """Vector.distance / similarity == Index.distances (mlvdb_pair_distances: the rescoring kernel's
    accumulate_rows) == the score Index.search returns == oracle.exact_distances (fp32 rounding)."""
    _check_index_distances(space, None, dim)


@pytest.mark.gpu
def test_pair_distances_through_logical_shards_and_on_hand_case():
    from mlvectordb_amd import Index

    rng = np.random.default_rng(3)
    rows = rng.standard_normal((500, 64), dtype=np.float32)
    qs = rng.standard_normal((3, 64), dtype=np.float32)
    one, many = Index(space="cosine"), Index(space="cosine", devices=[0, 0, 0])
    try:
        one.add_arrays(rows, "ns")
        many.add_arrays(rows, "ns")
        lab = rng.integers(0, 500, (3, 40))
        lab[1, 5] = -1
        a = one._ns["ns"].engine.pair_distances(qs, lab)
        b = many._ns["ns"].engine.pair_distances(qs, lab)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.isinf(a[0][1, 5])
        want = exact_scan.exact_distances(qs, rows, "cosine")
        assert np.abs(a[0][lab >= 0] - np.take_along_axis(want, np.maximum(lab, 0), axis=1)[lab >= 0]).max() < 1e-12
    finally:
        one.close()
        many.close()
    hand = Index(space="cosine")
    try:
        vs = {k: Vector(v) for k, v in HAND.items()}
        hand.add(list(vs.values()), "h")
        got = hand.distances(np.float32([QUERY]), [[v.id for v in vs.values()]], "h", "cosine")[0]
        for g, (k, v) in zip(got, vs.items()):
            assert g == pytest.approx(ref_cosine_similarity(np.float32(QUERY), np.float32(HAND[k])), abs=1e-6)
            assert g == pytest.approx(Vector(QUERY).similarity(v, "cosine"), abs=1e-6)
    finally:
        hand.close()
