"""BASELINE-size checks on the GPU through size-independent properties (no CPU scan of 1M+ rows).

Config 2 shape (1M x 768): the filter path must agree with the exact fp64 scan id for id (the
exact scan is itself pinned against the oracle at small sizes in test_gpu_parity.py), results
must be sorted, planted neighbours must be found, and tombstoning the winners must promote the
runners-up.
"""
import numpy as np
import pytest

from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 768, 10


@pytest.fixture(scope="module")
def big_engine():
    eng = HipScanEngine(D, "cosine", device=0, capacity_hint=N)
    for off, rows in synth.iter_corpus(0, N, D, threads=8):
        eng.append(rows)
    yield eng
    eng.close()


def test_million_rows_filter_equals_exact_and_properties(big_engine):
    eng = big_engine
    qs = synth.queries(64, D)
    planted = synth.corpus_rows(123_456, 4, D)
    qs[:4] = planted + 0.01 * qs[:4]  # near-copies of rows 123456..123459
    eng.set_strategy("exact")
    le, de, ce = eng.search(qs[:16], K)
    eng.set_strategy("filter")
    lf, df, cf = eng.search(qs, K)
    stats = eng.last_stats()
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0
    assert np.array_equal(lf[:16], le) and np.array_equal(df[:16], de)
    assert (cf == K).all() and (lf >= 0).all() and (lf < N).all()
    assert (np.diff(df, axis=1) >= 0).all()                       # nearest first
    assert [len(set(r)) for r in lf.tolist()] == [K] * len(lf)    # no row twice
    assert lf[:4, 0].tolist() == [123_456, 123_457, 123_458, 123_459]
    # batch=1 latency path (config 2): AUTO -> exact scan, same answer
    eng.set_strategy("auto")
    l1, d1, _ = eng.search(qs[5:6], K)
    assert np.array_equal(l1, lf[5:6]) and np.array_equal(d1, df[5:6])
    # tombstone every winner of query 7: the next search returns none of them, and what
    # comes back is no nearer than the old 10th
    eng.tombstone(lf[7])
    eng.set_strategy("filter")
    l2, d2, _ = eng.search(qs[7:40], K)
    assert not (set(l2[0].tolist()) & set(lf[7].tolist()))
    assert d2[0, 0] >= df[7, -1]
    assert np.array_equal(l2[1:], lf[8:40]) or np.intersect1d(lf[7], lf[8:40]).size > 0
