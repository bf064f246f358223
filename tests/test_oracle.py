"""The oracle against hand-computed answers and the committed golden vectors (CPU only)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import exact_scan

GOLDEN = Path(__file__).parent / "golden"


def test_l2_is_squared_and_exact_zero_for_stored_vector():
    rows = np.array([[1, 2, 3], [4, 6, 3], [1, 2, 3.5]], np.float32)
    labels, dist, counts = exact_scan.knn(rows[0], rows, 3, "l2")
    assert labels.tolist() == [[0, 2, 1]]
    assert dist[0].tolist() == [0.0, 0.25, 25.0]  # hnswlib l2 = squared distance, no sqrt
    assert counts.tolist() == [3]


def test_cosine_and_ip_definitions():
    rows = np.array([[1, 0, 0], [0, 1, 0], [0.8, 0.2, 0]], np.float32)
    q = np.array([[0.9, 0.1, 0]], np.float32)
    labels, dist, _ = exact_scan.knn(q, rows, 3, "cosine")
    sims = [float(np.dot(q[0].astype(np.float64), r.astype(np.float64)) /
                  (np.linalg.norm(q[0].astype(np.float64)) * np.linalg.norm(r.astype(np.float64)))) for r in rows]
    assert labels.tolist() == [[0, 2, 1]]
    assert np.allclose(1 - dist[0], [sims[0], sims[2], sims[1]], atol=1e-7)
    labels, dist, _ = exact_scan.knn(q, rows, 3, "ip")
    assert labels.tolist() == [[0, 2, 1]]
    assert np.allclose(dist[0], [1 - 0.9, 1 - np.float64(np.float32(0.8)) * np.float32(0.9) - np.float64(np.float32(0.2)) * np.float32(0.1), 1 - 0.1], atol=1e-7)


def test_ties_break_by_ascending_label_and_tombstones_are_skipped():
    rows = np.array([[1, 0], [0, 1], [1, 0], [1, 0]], np.float32)
    labels, dist, counts = exact_scan.knn([[1, 0]], rows, 4, "l2")
    assert labels.tolist() == [[0, 2, 3, 1]]
    labels, dist, counts = exact_scan.knn([[1, 0]], rows, 4, "l2", deleted=[0, 3])
    assert labels.tolist() == [[2, 1, -1, -1]] and counts.tolist() == [2]
    assert np.isinf(dist[0, 2:]).all()


def test_zero_vector_follows_hnswlib_normalisation():
    # x / (|x| + 1e-30): a zero row normalises to zero, similarity 0, distance 1
    rows = np.array([[0, 0], [1, 0], [-1, 0]], np.float32)
    labels, dist, _ = exact_scan.knn([[1, 0]], rows, 3, "cosine")
    assert labels.tolist() == [[1, 0, 2]]
    assert np.allclose(dist[0], [0.0, 1.0, 2.0])


def test_range_query_semantics():
    rows = np.array([[0, 0], [1, 0], [2, 0], [1, 0]], np.float32)
    (labels, dist), = exact_scan.range_query([[0, 0]], rows, 1.0, "l2")
    assert labels.tolist() == [0, 1, 3] and dist.tolist() == [0.0, 1.0, 1.0]
    (labels, dist), = exact_scan.range_query([[0, 0]], rows, 1.0, "l2", deleted=[1])
    assert labels.tolist() == [0, 3]


def test_postprocess_score_matches_index_search_rule():
    assert exact_scan.postprocess_score(0.25, "l2") == 0.25
    assert exact_scan.postprocess_score(0.25, "cosine") == 0.75  # index.py:126-127 flips on the metric ARGUMENT


@pytest.mark.parametrize("name", sorted(p.stem for p in GOLDEN.glob("knn_*.npz")))
def test_oracle_reproduces_committed_golden_vectors(name):
    from tests.golden.make_golden import regenerate_inputs

    g = np.load(GOLDEN / f"{name}.npz")
    meta = json.loads(str(g["meta"]))
    rows, qs, deleted = regenerate_inputs(meta)
    labels, dist, counts = exact_scan.knn(qs, rows, meta["k"], meta["space"], deleted=deleted)
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(counts, g["counts"])
    assert np.array_equal(dist, g["dist"])


def test_blas_baseline_engine_through_the_protocol_surface_agrees_with_the_oracle():
    """bench.py's cpu_baseline leg: QueryProcessor.find_similar_many -> Index.search_many -> BlasScanEngine (fp32 sgemm).
    A speed baseline, ranked on fp32 scores -- on well-separated data its ids are the oracle's."""
    import uuid

    from mlvectordb_amd import ArrayStorage, Index, QueryProcessor
    from oracle.blas_scan import BlasScanEngine

    rng = np.random.default_rng(0)
    rows = rng.standard_normal((5000, 48), dtype=np.float32)
    q = rng.standard_normal((9, 48), dtype=np.float32)
    for space in ("cosine", "l2"):
        qp = QueryProcessor(ArrayStorage(), Index(space=space, engine_factory=BlasScanEngine))
        ids = qp.upsert_arrays(rows, "cpu")
        hits = qp.find_similar_many(q, top_k=5, namespace="cpu", metric=space)
        want = exact_scan.knn(q, rows, 5, space)
        assert all([h["id"] for h in hits[i]] == [uuid.UUID(bytes=ids[l].tobytes()) for l in want[0][i]] for i in range(9))
        assert all(np.array_equal(h["values"], rows[l]) for h, l in zip(hits[3], want[0][3]))
