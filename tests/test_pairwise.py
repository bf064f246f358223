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
