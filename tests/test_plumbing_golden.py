"""SURVEY 8c fixture (3), plumbing part: tests/golden/plumbing_reference_qp.json was recorded by running the REFERENCE's own
QueryProcessor / StorageEngineInMemory / Vector (imported from /root/reference in the build container by
tests/golden/make_plumbing_golden.py) over this repo's Index.  Here the same calls go through THIS repo's QueryProcessor /
InMemoryStorage / Vector -- oracle engine on the CPU, HIP engine under -m gpu -- and must give what the reference's classes
gave: keys and their order, value types and dtypes, hit order, scores, dropped ids, and quirk Q4 of delete's rebuild.
Pins plumbing, not arithmetic: the reference's Index needs hnswlib, which is absent."""
import json
from pathlib import Path
from uuid import UUID

import numpy as np
import pytest

from mlvectordb_amd import Index, InMemoryStorage, QueryProcessor, VectorDTO

GOLD = json.loads((Path(__file__).parent / "golden" / "plumbing_reference_qp.json").read_text())["cases"]


def _factory(engine):
    if engine == "oracle":
        from oracle.engine import OracleScanEngine

        return {"engine_factory": OracleScanEngine}
    return {}


def _qp(engine, space="cosine", scope="namespace"):  # the reference's delete rebuilds from the affected namespace only (Q4)
    return QueryProcessor(InMemoryStorage(), Index(space=space, **_factory(engine)), rebuild_scope=scope)


def _dto(values, i, **meta):
    return VectorDTO(values=values, metadata={"i": i, **meta})


def _check(hits, want, exact):
    assert len(hits) == len(want)
    for h, w in zip(hits, want):
        assert list(h.keys()) == w["keys"]
        assert type(h["id"]).__name__ == w["id_type"] and isinstance(h["id"], UUID)
        assert type(h["values"]).__name__ == w["values_type"] and str(np.asarray(h["values"]).dtype) == w["values_dtype"]
        assert np.asarray(h["values"]).tolist() == w["values"]
        assert dict(h["metadata"]) == w["metadata"] and h["metadata"]["i"] == w["i"]
        assert type(h["score"]).__name__ == w["score_type"]
        assert h["score"] == w["score"] if exact else abs(h["score"] - w["score"]) <= 1e-5


def _replay(engine):
    exact = engine == "oracle"
    c = GOLD["hand_correctness"]
    qp = _qp(engine)
    qp.upsert_many([_dto(v, i, label=l) for i, (v, l) in enumerate(zip(c["rows"], c["labels"]))])
    _check(qp.find_similar(VectorDTO(values=c["query"], metadata={}), top_k=c["top_k"]), c["hits"], exact)

    c = GOLD["namespace_isolation"]
    qp = _qp(engine)
    qp.insert(_dto([1, 0, 0], 0, label="X"), namespace="alpha")
    qp.insert(_dto([0, 1, 0], 1, label="Y"), namespace="beta")
    q = VectorDTO(values=[1, 0, 0], metadata={})
    _check(qp.find_similar(q, top_k=1, namespace="alpha"), c["alpha"], exact)
    _check(qp.find_similar(q, top_k=1, namespace="beta"), c["beta"], exact)
    _check(qp.find_similar(q, top_k=1, namespace="gamma"), c["unknown"], exact)
    assert qp.list_namespaces() == c["namespaces"]

    c = GOLD["few_vectors"]
    qp = _qp(engine)
    qp.upsert_many([_dto([1, 0, 0], 0, label="A"), _dto([0, 1, 0], 1, label="B")])
    _check(qp.find_similar(q, top_k=c["top_k"]), c["hits"], exact)

    c = GOLD["delete"]
    qp = _qp(engine)
    qp.upsert_many([_dto([1, 0, 0], 0, label="A"), _dto([0, 1, 0], 1, label="B")])
    qp.upsert_many([_dto([0, 0, 1], 2, label="Z")], namespace="other")
    before = qp.find_similar(q, top_k=2)
    _check(before, c["before"], exact)
    assert len(qp.delete([before[0]["id"]])) == c["deleted_count"]
    _check(qp.find_similar(q, top_k=2), c["after"], exact)
    _check(qp.find_similar(VectorDTO(values=[0, 0, 1], metadata={}), top_k=1, namespace="other"), c["other_namespace_after"], exact)
    assert qp.get_namespace_count("other") == c["other_namespace_count_in_storage"]
    assert qp.get_namespace_count("default") == c["default_count_in_storage"]

    c = GOLD["dropped_ids"]
    qp = _qp(engine)
    qp.upsert_many([_dto([1, 0, 0], 0), _dto([0.9, 0.1, 0], 1), _dto([0, 1, 0], 2)])
    first = qp.find_similar(q, top_k=3)
    _check(first, c["before"], exact)
    qp._storage.delete(first[0]["id"], "default")
    _check(qp.find_similar(q, top_k=3), c["after"], exact)

    rows = np.random.default_rng(1234).standard_normal((10_000, 128), dtype=np.float32)
    queries = np.random.default_rng(4321).standard_normal((8, 128), dtype=np.float32)
    for space in ("cosine", "l2"):
        c = GOLD[f"random_10k_128_k5_{space}_space"]
        qp = _qp(engine, space)
        qp.upsert_many([_dto(r, i) for i, r in enumerate(rows)])
        for qv, want in zip(queries, c["hits"]):
            got = qp.find_similar(VectorDTO(values=qv, metadata={}), top_k=c["top_k"], metric=c["metric"])
            assert [h["metadata"]["i"] for h in got] == [w["i"] for w in want]
            for h, w in zip(got, want):
                assert h["score"] == w["score"] if exact else abs(h["score"] - w["score"]) <= 1e-5 * max(1.0, abs(w["score"]))
        _check(qp.find_similar(VectorDTO(values=queries[0], metadata={}), top_k=1, metric="cosine"), c["first_hit"], exact)
        qp._index.close()


def test_this_repos_query_processor_replays_the_reference_query_processors_recorded_outputs():
    _replay("oracle")


@pytest.mark.gpu
def test_the_hip_engine_replays_the_reference_query_processors_recorded_outputs():
    _replay("hip")
