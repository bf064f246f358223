"""Every known-answer test the reference holds for the hot path, restated against this repo.

Source of the behaviours (restated, not copied): reference tests/test_index.py:25-71 and
tests/test_query_processor.py:36-131.  They pin ordering, membership, counts, sign and type --
no reference test asserts a numeric score.  Each case runs twice: with the oracle engine (CPU,
pins the oracle + host logic) and with the HIP engine (``-m gpu``, the product path).
"""
from uuid import UUID

import numpy as np
import pytest

from mlvectordb_amd import Index, InMemoryStorage, QueryProcessor, Vector, VectorDTO
from oracle.engine import OracleScanEngine

ENGINES = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def make_index(kind: str, space: str) -> Index:
    if kind == "oracle":
        return Index(space=space, engine_factory=OracleScanEngine)
    return Index(space=space)  # HipScanEngine; raises if the library / GPU is missing


@pytest.fixture(params=[2, 5, 100])
def sample_vectors(request):
    np.random.seed(42)
    data = np.random.rand(request.param, 16).astype(np.float32)
    return [Vector(values=v.tolist(), metadata={"i": i}) for i, v in enumerate(data)]


# ---- reference tests/test_index.py, Index(space="l2")
@pytest.mark.parametrize("kind", ENGINES)
def test_add_and_search_various_sizes(kind, sample_vectors):
    index = make_index(kind, "l2")
    index.add(sample_vectors, "varied_ns")
    noisy = sample_vectors[0].values + np.random.normal(0, 0.01, size=sample_vectors[0].values.shape)  # float64 query
    results = index.search(VectorDTO(values=noisy, metadata={}), top_k=5, namespace="varied_ns", metric="l2")
    assert len(results) == min(5, len(sample_vectors))
    ids = {v.id for v in sample_vectors}
    for r in results:
        assert r.vector_id in ids
        assert isinstance(r.score, float)
        assert r.score >= 0.0  # squared l2
    assert results[0].vector_id == sample_vectors[0].id
    assert [r.score for r in results] == sorted(r.score for r in results)


@pytest.mark.parametrize("kind", ENGINES)
def test_remove_and_search_various_sizes(kind, sample_vectors):
    index = make_index(kind, "l2")
    index.add(sample_vectors, "remove_ns")
    to_remove = [v.id for v in sample_vectors[:2]]
    index.remove(to_remove, "remove_ns")
    results = index.search(VectorDTO(values=sample_vectors[0].values, metadata={}), top_k=5,
                           namespace="remove_ns", metric="l2")
    assert not (set(to_remove) & {r.vector_id for r in results})
    assert len(results) == min(5, len(sample_vectors) - 2)


@pytest.mark.parametrize("kind", ENGINES)
def test_rebuild_many(kind, sample_vectors):
    index = make_index(kind, "l2")
    half = len(sample_vectors) // 2 or 1
    source = {"ns1": sample_vectors[:half], "ns2": sample_vectors[half:]}
    index.rebuild(source, metric="l2")
    for ns, vecs in source.items():
        results = index.search(VectorDTO(values=vecs[0].values, metadata={}), top_k=3, namespace=ns, metric="l2")
        assert len(results) > 0
        assert results[0].vector_id == vecs[0].id
        assert results[0].score == 0.0


# ---- reference tests/test_query_processor.py, Index(space="cosine")
def make_processor(kind):
    return QueryProcessor(InMemoryStorage(), make_index(kind, "cosine"))


def dto(x, y, z, label):
    return VectorDTO(values=[x, y, z], metadata={"label": label})


def true_cosine(a, b):
    a, b = np.array(a, dtype=np.float64), np.array(b, dtype=np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


@pytest.mark.parametrize("kind", ENGINES)
def test_insert_and_storage_integrity(kind):
    qp = make_processor(kind)
    qp.insert(dto(1, 0, 0, "A"))
    qp.insert(dto(0, 1, 0, "B"))
    assert qp._storage.total_vectors == 2
    labels = {v.metadata["label"] for v in qp._storage.namespace_map["default"]}
    assert labels == {"A", "B"}


@pytest.mark.parametrize("kind", ENGINES)
def test_find_similar_correctness(kind):
    qp = make_processor(kind)
    qp.upsert_many([dto(1, 0, 0, "A"), dto(0, 1, 0, "B"), dto(0.8, 0.2, 0, "C")])
    query = VectorDTO(values=[0.9, 0.1, 0], metadata={})
    results = qp.find_similar(query, top_k=3)
    assert [r["metadata"]["label"] for r in results] == ["A", "C", "B"]
    sims = [true_cosine(query.values, r["values"]) for r in results]
    assert sims == pytest.approx(sorted(sims, reverse=True), rel=1e-4)
    # score = 1 - cosine distance (index.py:126-127): the similarity itself
    assert [r["score"] for r in results] == pytest.approx(sims, abs=1e-5)
    for r in results:
        assert set(r) == {"id", "values", "metadata", "score"}
        assert isinstance(r["id"], UUID) and isinstance(r["score"], float)
        assert isinstance(r["values"], np.ndarray) and r["values"].dtype == np.float32


@pytest.mark.parametrize("kind", ENGINES)
def test_namespace_isolation(kind):
    qp = make_processor(kind)
    qp.insert(dto(1, 0, 0, "X"), namespace="alpha")
    qp.insert(dto(0, 1, 0, "Y"), namespace="beta")
    q = VectorDTO(values=[1, 0, 0], metadata={})
    r1 = qp.find_similar(q, top_k=1, namespace="alpha")
    r2 = qp.find_similar(q, top_k=1, namespace="beta")
    assert r1[0]["metadata"]["label"] == "X" and r2[0]["metadata"]["label"] == "Y"
    assert r1[0]["id"] != r2[0]["id"]
    assert qp.find_similar(q, top_k=1, namespace="gamma") == []  # unknown namespace (index.py:98-99)


@pytest.mark.parametrize("kind", ENGINES)
def test_delete_removes_from_storage_and_index(kind):
    qp = make_processor(kind)
    qp.upsert_many([dto(1, 0, 0, "A"), dto(0, 1, 0, "B")])
    q = VectorDTO(values=[1, 0, 0], metadata={})
    before = qp.find_similar(q, top_k=2)
    assert len(before) == 2
    gone = before[0]["id"]
    assert qp.delete([gone]) == [gone]  # 1/2 deleted >= 0.2: this also takes the rebuild path
    assert gone not in [v.id for v in qp._storage.namespace_map["default"]]
    after = qp.find_similar(q, top_k=2)
    assert [r["id"] for r in after] == [before[1]["id"]]


@pytest.mark.parametrize("kind", ENGINES)
def test_search_with_many_vectors(kind):
    qp = make_processor(kind)
    np.random.seed(42)
    qp.upsert_many([VectorDTO(values=np.random.rand(10).tolist(), metadata={"label": f"V{i}"}) for i in range(100)])
    results = qp.find_similar(VectorDTO(values=np.random.rand(10).tolist(), metadata={}), top_k=5)
    assert len(results) == 5
    assert all(isinstance(r["id"], UUID) for r in results)
    assert [r["score"] for r in results] == sorted((r["score"] for r in results), reverse=True)


@pytest.mark.parametrize("kind", ENGINES)
def test_search_with_few_vectors(kind):
    qp = make_processor(kind)
    qp.upsert_many([dto(1, 0, 0, "A"), dto(0, 1, 0, "B")])
    results = qp.find_similar(VectorDTO(values=[1, 0, 0], metadata={}), top_k=5)
    assert len(results) == 2  # top_k clamps to the live count (index.py:107)
    assert results[0]["metadata"]["label"] == "A"
