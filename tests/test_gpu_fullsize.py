"""BASELINE-size checks on the GPU (SURVEY 8d sizes; no CPU scan of 10M rows).

* configs[1] shape (1M x 768): filter == exact fp64 scan, properties, and a 1M-row comparison with the NumPy
  oracle itself (32 queries).
* configs[2] (10M x 768 cosine, batch 256, the headline): every one of the 256 queries, int8 body, three scan
  rounds (the third only exists beyond 1,572,864 rows), ids and fp32 distances equal to the exact fp64 scan
  (itself pinned against the oracle at small sizes in test_gpu_parity.py and at 1M rows here) AND, for 8 of the queries, to
  the NumPy fp64 oracle over the whole 10M rows (chunk-wise scan + merge); planted near-copies; a 1024-query (four-pass) wave;
  again after tombstoning 10 % of the rows with default_rng(99).
* configs[3] (10M x 768 squared-l2): kNN the same way, and the range query at the mean 10th-neighbour radius
  against the exact range scan for every query, including the ones with more hits than a candidate list holds.

The reference pins "many vectors" at 50 rows (reference tests/test_query_processor.py:108-119); this is that
pin at the size the north star asks for.
"""
import numpy as np
import pytest

from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine
from tests.helpers import assert_knn_matches, oracle_knn

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 768, 10
N10, B = 10_000_000, 256
PLANT = [123_456, 5_000_001, 9_999_999, 1_572_863, 1_572_864, 65_279, 65_280]  # incl. both sides of the round-2 / round-3 boundaries


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
    # batch=1 latency path (config 2): AUTO -> narrow filter kernel, same answer
    eng.set_strategy("auto")
    l1, d1, _ = eng.search(qs[5:6], K)
    assert np.array_equal(l1, lf[5:6]) and np.array_equal(d1, df[5:6])
    # the NumPy oracle itself on the whole 1M rows (fp64, ~10 s of host BLAS), 32 queries
    rows = synth.corpus_rows(0, N, D)
    assert_knn_matches((lf[:32], df[:32], cf[:32]), oracle_knn(qs[:32], rows, K, "cosine"), "1M/oracle")
    del rows
    # tombstone every winner of query 7: the next search returns none of them, and what
    # comes back is no nearer than the old 10th
    eng.tombstone(lf[7])
    eng.set_strategy("filter")
    l2, d2, _ = eng.search(qs[7:40], K)
    assert not (set(l2[0].tolist()) & set(lf[7].tolist()))
    assert d2[0, 0] >= df[7, -1]
    assert np.array_equal(l2[1:], lf[8:40]) or np.intersect1d(lf[7], lf[8:40]).size > 0


# ------------------------------------------------------------------------------------------------ 10M rows
def _load_10m(space):
    eng = HipScanEngine(D, space, device=0, capacity_hint=N10)
    for off, rows in synth.iter_corpus(0, N10, D, threads=16):
        eng.append(rows)
    return eng


def _queries_with_plants():
    qs = synth.queries(B, D)
    for i, row in enumerate(PLANT):
        qs[i] = synth.corpus_rows(row, 1, D)[0] + 0.01 * qs[i]
    return qs


def _both_strategies(eng, qs, k):
    eng.set_strategy("exact")
    want = eng.search(qs, k)
    assert eng.last_stats()["strategy_used"] == 1
    eng.set_strategy("auto")
    got = eng.search(qs, k)
    return got, want, eng.last_stats()


def _assert_identical(got, want, tag):
    (gl, gd, gc), (wl, wd, wc) = got, want
    bad = np.nonzero((gl != wl).any(axis=1))[0]
    assert bad.size == 0, f"{tag}: ids differ from the exact fp64 scan for queries {bad[:8]}: {gl[bad[0]]} vs {wl[bad[0]]}"
    assert np.array_equal(gc, wc) and np.array_equal(gd, wd), f"{tag}: fp32 distances differ"


@pytest.fixture(scope="module")
def engine_10m_cosine():
    eng = _load_10m("cosine")
    yield eng
    eng.close()


def test_config3_10m_cosine_all_256_queries_equal_the_exact_scan(engine_10m_cosine):
    eng = engine_10m_cosine
    qs = _queries_with_plants()
    got, want, st = _both_strategies(eng, qs, K)
    assert st["strategy_used"] == 2 and st["bound_dtype"] == 2 and st["scan_launches"] == 3, st
    assert st["rows_scanned"] in (N10, N10 - 3840) and st["fallback_queries"] == 0  # exact seed: the prefix rows join round 1
    _assert_identical(got, want, "10M/cosine")
    lf, df, cf = got
    assert (cf == K).all() and (lf >= 0).all() and (lf < N10).all()
    assert (np.diff(df, axis=1) >= 0).all()
    assert lf[:len(PLANT), 0].tolist() == PLANT
    # batches of 1 and 2 queries on the same 10M rows (round 3: exact prefix seed, three narrow scan rounds, fused finish):
    # a query's answer does not depend on the batch it rides in
    eng.set_strategy("auto")
    for lo, hi in ((200, 201), (3, 5), (0, 1)):
        l1, d1, c1 = eng.search(qs[lo:hi], K)
        st1 = eng.last_stats()
        assert st1["strategy_used"] == 2 and st1["bound_dtype"] == 2 and st1["fallback_queries"] == 0, st1
        assert np.array_equal(l1, lf[lo:hi]) and np.array_equal(d1, df[lo:hi]) and (c1 == K).all()
    # ... and against the NumPy ORACLE itself at the full 10M rows, for 8 of the queries (the planted ones + one plain): the fp64
    # oracle scans the regenerated 1M-row chunks one by one and the per-chunk top-k are merged by (fp64 distance, label), so the
    # headline-size answer is oracle-pinned too, not only pinned through the GPU's exact scan (about a minute of host NumPy)
    from oracle import exact_scan

    sub = np.concatenate([np.arange(len(PLANT)), [200]])
    parts_l, parts_d = [], []

    def block_topk(rows, base):  # (cache-sized blocks: the fp64 copy of a block stays out of DRAM)
        d64 = exact_scan.exact_distances(qs[sub], rows, "cosine")
        part = np.argpartition(d64, K - 1, axis=1)[:, :K]
        return part + base, np.take_along_axis(d64, part, axis=1)

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(8) as pool:  # (NumPy releases the GIL in the conversions and products: eight blocks at a time)
        for off, rows in synth.iter_corpus(0, N10, D, threads=16):
            los = range(0, rows.shape[0], 50_000)
            for pl, pd in pool.map(lambda lo: block_topk(rows[lo:lo + 50_000], off + lo), los):
                parts_l.append(pl)
                parts_d.append(pd)
            del rows
    from mlvectordb_amd.sharded import merge_topk

    ol, od, _ = merge_topk(parts_l, parts_d, K)
    assert np.array_equal(lf[sub], ol), "10M rows: ids differ from the NumPy oracle"
    assert np.abs(df[sub] - od).max() <= 1e-5
    # BASELINE configs[4]'s per-GPU shape: a 1024-query wave = four 256-query passes over the 10M-row shard (what every rank of
    # the 8-GPU job runs; the cross-shard merge is covered by tests/test_sharded.py and bench.py's sharded gate)
    big = np.concatenate([qs, np.random.default_rng(1024).standard_normal((1024 - qs.shape[0], qs.shape[1])).astype(np.float32)])
    got4, want4, st4 = _both_strategies(eng, big, K)
    assert st4["strategy_used"] == 2 and st4["scan_launches"] == 12 and st4["fallback_queries"] == 0, st4
    _assert_identical(got4, want4, "10M/cosine/batch1024")
    assert np.array_equal(got4[0][:qs.shape[0]], lf)  # a query's answer does not depend on the pass it rides in
    # top_k = 100 and 1000 at the full size: big-k passes on the filter path against the paged exact scan
    for kk, nqq in ((100, B), (1000, 32)):
        gotk, wantk, stk = _both_strategies(eng, qs[:nqq], kk)
        assert stk["strategy_used"] == 2 and stk["bound_dtype"] == 2 and stk["fallback_queries"] == 0, stk
        assert stk["candidates_rescored"] < 2 * kk * nqq, stk
        _assert_identical(gotk, wantk, f"10M/cosine/k{kk}")
        assert np.array_equal(gotk[0][:, :K], lf[:nqq])  # the first 10 of the 100 are the k = 10 answer
    # SURVEY 8d's secondary run: 10 % random tombstones
    dead = np.nonzero(np.random.default_rng(99).random(N10) < 0.1)[0].astype(np.int64)
    assert eng.tombstone(dead) == dead.size
    got2, want2, st2 = _both_strategies(eng, qs, K)
    assert st2["strategy_used"] == 2 and st2["bound_dtype"] == 2 and st2["scan_launches"] == 3
    _assert_identical(got2, want2, "10M/cosine/tombstones")
    assert not np.isin(got2[0], dead).any()
    alive = ~np.isin(lf, dead)
    for i in range(B):  # surviving old winners keep their relative order at the head of the new list
        kept = lf[i][alive[i]]
        assert np.array_equal(got2[0][i, :kept.size], kept)


@pytest.fixture(scope="module")
def engine_10m_l2(engine_10m_cosine):
    engine_10m_cosine.close()  # 38 GB each (54 with MLVDB_SHADOW=bf16): one at a time
    eng = _load_10m("l2")
    yield eng
    eng.close()


def test_config4_10m_l2_knn_and_range_equal_the_exact_scans(engine_10m_l2):
    eng = engine_10m_l2
    qs = _queries_with_plants()
    got, want, st = _both_strategies(eng, qs, K)
    assert st["strategy_used"] == 2 and st["scan_launches"] == 3 and st["fallback_queries"] == 0
    _assert_identical(got, want, "10M/l2")
    assert got[0][:len(PLANT), 0].tolist() == PLANT
    for lo, hi in ((200, 201), (5, 7)):  # batches of 1 and 2 queries (the small-batch steps on the l2 arithmetic)
        l1, d1, c1 = eng.search(qs[lo:hi], K)
        assert eng.last_stats()["strategy_used"] == 2 and eng.last_stats()["fallback_queries"] == 0
        assert np.array_equal(l1, got[0][lo:hi]) and np.array_equal(d1, got[1][lo:hi])
    # range query at the mean 10th-neighbour distance: filter path vs the exact range scan
    radius = float(got[1][len(PLANT):, K - 1].mean())
    eng.set_strategy("auto")
    hits = eng.range(qs, radius, 1024)
    st_r = eng.last_stats()
    assert st_r["strategy_used"] == 2
    eng.set_strategy("exact")
    hits_exact = eng.range(qs, radius, 1024)
    assert eng.last_stats()["strategy_used"] == 1
    n_hits = np.array([len(h[0]) for h in hits])
    assert n_hits.max() > 1024, "the workload is supposed to contain queries with large hit lists"
    for i, ((hl, hd), (el, ed)) in enumerate(zip(hits, hits_exact)):
        assert np.array_equal(hl, el) and np.array_equal(hd, ed), f"range hits of query {i} differ ({len(hl)} vs {len(el)})"
        assert (hd <= np.float32(radius)).all() and (np.diff(hd) >= 0).all()
        m = min(len(hl), K)
        assert np.array_equal(hl[:m], got[0][i, :m])
        if len(hl) < K:
            assert got[1][i, len(hl)] > np.float32(radius)
