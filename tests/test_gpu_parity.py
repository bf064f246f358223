"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bar (north star): ids bit-exact, scores within 1e-5.  Sizes are chosen so the NumPy oracle
finishes in seconds; BASELINE-size behaviour is covered by properties in test_gpu_fullsize.py.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from mlvectordb_amd.engine import HipScanEngine
from oracle import exact_scan
from tests.helpers import SCORE_ATOL, assert_knn_matches, deleted_mask, make_case, oracle_knn

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).parent / "golden"


def run_hip(rows, qs, k, space, strategy, deleted=None, append_chunks=1):
    eng = HipScanEngine(rows.shape[1], space, device=0, strategy=strategy)
    try:
        for part in np.array_split(rows, append_chunks):
            if len(part):
                eng.append(part)
        if deleted is not None and deleted.any():
            changed = eng.tombstone(np.nonzero(deleted)[0])
            assert changed == int(deleted.sum())
            assert eng.tombstone(np.nonzero(deleted)[0][:3]) == 0  # already deleted: no state change
        assert eng.counts() == (rows.shape[0], int(deleted.sum()) if deleted is not None else 0)
        return eng.search(qs, k), eng.last_stats()
    finally:
        eng.close()


EXACT_CASES = [
    # seed, n, d, nq, k, deleted_frac, dup
    (1, 1, 3, 1, 1, 0.0, False),
    (2, 15, 3, 2, 5, 0.0, False),
    (3, 17, 10, 3, 5, 0.2, True),
    (4, 100, 16, 5, 5, 0.1, True),
    (5, 1000, 16, 8, 5, 0.1, True),
    (6, 1037, 100, 9, 7, 0.0, True),
    (7, 4096, 128, 16, 5, 0.1, True),
    (8, 5000, 130, 4, 64, 0.3, False),
    (9, 2048, 768, 8, 10, 0.1, True),
    (10, 30000, 64, 2, 10, 0.0, False),
    (11, 20000, 256, 1, 1, 0.5, False),
]


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("case", EXACT_CASES, ids=lambda c: f"n{c[1]}d{c[2]}q{c[3]}k{c[4]}")
def test_exact_scan_matches_oracle(case, space):
    seed, n, d, nq, k, frac, dup = case
    rows, qs = make_case(seed, n, d, nq, dup=dup)
    deleted = deleted_mask(seed, n, frac)
    got, stats = run_hip(rows, qs, k, space, "exact", deleted, append_chunks=3 if n > 50 else 1)
    assert stats["strategy_used"] == 1
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"exact/{space}/n{n}d{d}")


FILTER_CASES = [
    # seed, n, d, nq, k, deleted_frac, dup, clustered
    (21, 500, 64, 3, 5, 0.0, True, False),      # far fewer rows than one tile
    (22, 3000, 192, 24, 10, 0.05, True, False),  # 3 chunks: the R=3 kernel
    (23, 4096, 128, 16, 5, 0.1, True, False),    # 2 chunks: the R=4 kernel
    (24, 2500, 64, 40, 10, 0.1, False, False),   # 1 chunk: the R=2 kernel
    (25, 10000, 128, 256, 5, 0.1, True, False),  # a full 256-query pass
    (26, 9000, 768, 64, 10, 0.0, True, False),   # the benchmark dimension
    (27, 70000, 64, 300, 10, 0.02, False, False),  # two passes (256 + 44), two scan rounds
    (28, 6000, 256, 32, 64, 0.0, False, False),  # k = MAX_TOPK
    (29, 5000, 128, 16, 10, 0.0, True, True),    # near-ties everywhere: candidate lists overflow -> fallback
    (30, 3000, 320, 8, 3, 0.6, False, False),    # 5 chunks (R=2), most rows tombstoned
]


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("case", FILTER_CASES, ids=lambda c: f"n{c[1]}d{c[2]}q{c[3]}k{c[4]}")
def test_filter_path_matches_oracle(case, space):
    seed, n, d, nq, k, frac, dup, clustered = case
    rows, qs = make_case(seed, n, d, nq, dup=dup, clustered=clustered)
    deleted = deleted_mask(seed, n, frac)
    got, stats = run_hip(rows, qs, k, space, "filter", deleted, append_chunks=2)
    assert stats["strategy_used"] == 2
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"filter/{space}/n{n}d{d}")


def test_filter_bound_prunes_on_random_data():
    """On random-normal data the bf16 bounds must discard almost everything (else the path is pointless)."""
    rows, qs = make_case(31, 60000, 128, 64)
    got, stats = run_hip(rows, qs, 10, "cosine", "filter")
    assert_knn_matches(got, oracle_knn(qs, rows, 10, "cosine"), "filter/prune")
    assert stats["fallback_queries"] == 0
    assert stats["candidates_rescored"] < 64 * 60000 * 0.02


def test_zero_rows_zero_queries_and_scale_extremes():
    rng = np.random.default_rng(41)
    rows = rng.standard_normal((3000, 64)).astype(np.float32)
    rows[5] = 0.0          # hnswlib: zero row normalises to zero -> cosine distance 1
    rows[6] *= 1e-18       # tiny but non-zero norm
    rows[7] *= 1e12        # huge norm
    qs = rng.standard_normal((12, 64)).astype(np.float32)
    qs[1] = 0.0
    qs[2] *= 1e-15
    for space in ("cosine", "l2", "ip"):
        for strategy in ("exact", "filter"):
            got, _ = run_hip(rows, qs, 8, space, strategy)
            assert_knn_matches(got, oracle_knn(qs, rows, 8, space), f"{strategy}/{space}/extremes")


@pytest.mark.parametrize("name", sorted(p.stem for p in GOLDEN.glob("knn_*.npz")))
@pytest.mark.parametrize("strategy", ["exact", "filter"])
def test_committed_golden_vectors(name, strategy):
    from tests.golden.make_golden import regenerate_inputs

    g = np.load(GOLDEN / f"{name}.npz")
    meta = json.loads(str(g["meta"]))
    rows, qs, deleted = regenerate_inputs(meta)
    (labels, dist, counts), _ = run_hip(rows, qs, meta["k"], meta["space"], strategy, deleted)
    assert np.array_equal(labels, g["labels"]) and np.array_equal(counts, g["counts"])
    assert np.abs(dist - g["dist"]).max() <= SCORE_ATOL


@pytest.mark.parametrize("space", ["l2", "cosine"])
@pytest.mark.parametrize("n,d,nq,k", [(3000, 64, 3, 100), (5000, 20, 2, 333), (700, 128, 9, 700), (2000, 64, 1, 1000)])
def test_topk_above_64_is_served_in_rank_ordered_pages(n, d, nq, k, space):
    rows, qs = make_case(55, n, d, nq, dup=True)
    deleted = deleted_mask(55, n, 0.1)
    got, stats = run_hip(rows, qs, k, space, "auto", deleted)
    assert stats["strategy_used"] == 1 and stats["scan_launches"] == -(-k // 64)
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"paged/{space}/n{n}k{k}")


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("k", [65, 100, 333, 1000])
def test_topk_65_to_1000_stays_on_the_filter_path(space, k):
    """the reference clamps top_k only to the live count (index.py:107) and its REST model admits 1..1000
    (rest_api.py:24).  top_k in (64, 1024] runs as big-k passes: dense seed of 65,280 rows, scan rounds on the int8 body,
    mid (fp16) bounds, exact rescoring of the survivors -- ids of the oracle, tombstones and duplicates included."""
    d, n = 256, 150_001
    rows, qs = make_case(1300 + k, n, d, 40, dup=True)
    deleted = deleted_mask(23, n, 0.05)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        for part in np.array_split(rows, 3):
            eng.append(part)
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, k)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["bound_dtype"] == 2 and st["fallback_queries"] == 0 and st["scan_launches"] >= 1, st
        assert st["candidates_rescored"] < 3 * k * qs.shape[0], st   # the mid bounds leave ~k rows per query for the exact gather
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"bigk/{space}/k{k}")
        few = eng.search(qs[:3], k)     # 1-8 queries: the narrow kernel does the rounds
        assert eng.last_stats()["strategy_used"] == 2
        assert_knn_matches(few, oracle_knn(qs[:3], rows, k, space, deleted), f"bigk-narrow/{space}/k{k}")
    finally:
        eng.close()


@pytest.mark.parametrize("space,d,n,k", [("cosine", 100, 80_001, 100), ("l2", 128, 40_001, 100), ("ip", 768, 70_003, 200),
                                         ("l2", 768, 40_001, 100), ("cosine", 1000, 33_001, 65)])
def test_bigk_on_other_shapes_masks_and_fallbacks(space, d, n, k, monkeypatch):
    """Big-k passes on a padded int8 shadow (d = 100, 1000), on the bf16 shadow (d = 128), on a corpus smaller than the dense
    seed (no scan round at all), under a row mask, with more than 256 queries; L2_SHADOW=0 (no mid shadow) takes the paged
    exact scan."""
    rows, qs = make_case(1500 + d, n, d, 300 if d == 128 else 20, dup=True)
    deleted = deleted_mask(29, n, 0.05)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        eng.append(rows)
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, k)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["fallback_queries"] == 0, st
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"bigk-shapes/{space}/d{d}")
        mask = (np.arange(n) % 3 != 0).astype(np.uint8)
        gm = eng.search(qs[:10], k, mask)
        assert_knn_matches(gm, oracle_knn(qs[:10], rows, k, space, deleted | (mask == 0)), f"bigk-mask/{space}/d{d}")
        eng.set_tuning(L2_SHADOW=0)
        paged = eng.search(qs[:10], k)
        assert eng.last_stats()["strategy_used"] == 1
        assert_knn_matches(paged, oracle_knn(qs[:10], rows, k, space, deleted), f"bigk-paged/{space}/d{d}")
    finally:
        eng.close()


def test_bigk_with_thousands_of_equal_rows_falls_back_exactly():
    """12,000 copies of one row next to the query: every copy ties into the top k, more than the 8,192 hits one block ranks --
    the query is flagged and served by the paged exact scan; the answer is still the oracle's (lowest labels first)."""
    n, d, k = 120_000, 256, 100
    rows, qs = make_case(4343, n, d, 4)
    near = qs[0] + 0.01 * np.random.default_rng(5).standard_normal(d).astype(np.float32)
    where = np.random.default_rng(6).choice(np.arange(5_000, n), size=12_000, replace=False)
    rows[where] = near
    got, stats = run_hip(rows, qs, k, "cosine", "filter", None, append_chunks=2)
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] >= 1, stats
    assert_knn_matches(got, oracle_knn(qs, rows, k, "cosine"), "bigk equal rows")
    assert got[0][0][0] == 1 and np.array_equal(got[0][0][1:], np.sort(where)[:k - 1])  # (make_case: query 0 is row 1 itself)


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
def test_filter_without_bf16_shadow_matches_oracle(space, monkeypatch):
    """MLVDB_NO_SHADOW=1: the scan converts the fp32 panels in registers instead of reading the shadow."""
    monkeypatch.setenv("MLVDB_NO_SHADOW", "1")
    for seed, n, d, nq, k in [(61, 9000, 768, 40, 10), (62, 5000, 128, 256, 5), (63, 3000, 192, 24, 10)]:
        rows, qs = make_case(seed, n, d, nq, dup=True)
        deleted = deleted_mask(seed, n, 0.1)
        got, stats = run_hip(rows, qs, k, space, "filter", deleted)
        assert stats["strategy_used"] == 2
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"noshadow/{space}/n{n}d{d}")


@pytest.mark.parametrize("space,shadow", [("l2", "default"), ("cosine", "default"), ("ip", "default"), ("cosine", "bf16")])
def test_int8_only_shadow_matches_oracle(space, shadow, monkeypatch):
    """dim % 256 == 0: by default (round 3) the index keeps NO bf16 shadow in HBM (1.25x instead of 1.75x the corpus): seeding
    pass, small batches of every space, scans, range and row-mask searches all run on the int8 shadow.  MLVDB_SHADOW=bf16 keeps
    both shadows (round 2's default).  Same answers."""
    if shadow == "bf16":
        monkeypatch.setenv("MLVDB_SHADOW", "bf16")
    else:
        monkeypatch.delenv("MLVDB_SHADOW", raising=False)
    for seed, n, d, nq, k in [(64, 40000, 256, 40, 10), (65, 9000, 768, 300, 5), (66, 36000, 512, 3, 10)]:
        rows, qs = make_case(seed, n, d, nq, dup=True)
        deleted = deleted_mask(seed, n, 0.1)
        eng = HipScanEngine(d, space, device=0, strategy="filter")
        try:
            eng.append(rows[: n // 2])
            eng.append(rows[n // 2:])
            eng.tombstone(np.nonzero(deleted)[0])
            got = eng.search(qs, k)
            st = eng.last_stats()
            assert st["strategy_used"] == 2 and st["bound_dtype"] == 2, st
            assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"int8only/{space}/n{n}d{d}")
            mask = (np.arange(n) % 3 != 0).astype(np.uint8)     # row-mask search: no shadow serves it, still exact
            gm = eng.search(qs[:8], k, mask)
            assert_knn_matches(gm, oracle_knn(qs[:8], rows, k, space, deleted | (mask == 0)), f"int8only-mask/{space}/n{n}")
            radius = float(got[1][:, min(k, 5) - 1].mean())
            hits = eng.range(qs[:6], radius, 64)
            want = exact_scan.range_query(qs[:6], rows, radius, space, deleted=deleted)
            assert all(np.array_equal(h[0], w[0]) for h, w in zip(hits, want))
            old = eng.compact()
            assert old.size == n - int(deleted.sum())
            assert_knn_matches(eng.search(qs[:16], k), oracle_knn(qs[:16], rows[old], k, space), f"int8only-compact/{space}")
        finally:
            eng.close()


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("d", [3, 20, 48, 65, 100, 300, 384, 1000])
def test_any_dim_runs_on_the_zero_padded_int8_shadow(space, d):
    """the reference takes any dim from the first vector (index.py:54).  The int8 shadow is padded with zero
    columns to a multiple of 256 (they change neither a dot product nor a norm), so d = 100 / 300 / 384 / 1000 -- and d < 64,
    where the shadow is wider than the rows -- run the int8 body: 256-query passes, 1-8 queries (narrow kernel), row masks,
    range queries, appends in pieces, tombstones, compaction."""
    n = 60_001 if d <= 384 else 30_001
    rows, qs = make_case(700 + d, n, d, 44, dup=True)
    deleted = deleted_mask(17, n, 0.07)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        eng.append(rows[: n // 2])
        eng.append(rows[n // 2:])
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, 10)
        st = eng.last_stats()
        # l2 / ip rows of a few columns quantise too coarsely for the index-wide error criterion (a short row beside a long one
        # in its scale group: relative error > 0.03): such an index keeps to the exact scan, whatever strategy was asked for
        i8 = st["strategy_used"] == 2
        assert i8 or (d < 64 and space != "cosine"), st
        # (d = 3: nearly every row of the top 10 ties within the int8 band -- a query may take its exact fallback)
        assert not i8 or (st["bound_dtype"] == 2 and (st["fallback_queries"] == 0 or d < 16)), st
        assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"padded-int8/{space}/d{d}")
        small = eng.search(qs[:3], 7)
        st = eng.last_stats()
        assert not i8 or (st["strategy_used"] == 2 and st["bound_dtype"] == 2), st
        assert_knn_matches(small, oracle_knn(qs[:3], rows, 7, space, deleted), f"padded-int8-narrow/{space}/d{d}")
        mask = (np.arange(n) % 4 != 1).astype(np.uint8)
        gm = eng.search(qs[:12], 10, mask)
        assert_knn_matches(gm, oracle_knn(qs[:12], rows, 10, space, deleted | (mask == 0)), f"padded-int8-mask/{space}/d{d}")
        radius = float(got[1][:, 4].mean())
        hits = eng.range(qs[:6], radius, 64)
        want = exact_scan.range_query(qs[:6], rows, radius, space, deleted=deleted)
        assert all(np.array_equal(h[0], w[0]) for h, w in zip(hits, want))
        old = eng.compact()
        assert_knn_matches(eng.search(qs[:16], 10), oracle_knn(qs[:16], rows[old], 10, space), f"padded-int8-compact/{space}/d{d}")
    finally:
        eng.close()


@pytest.mark.parametrize("d", [16, 32, 40])
def test_auto_routes_small_dims_by_measured_cost(d):
    """dim < 64: the padded int8 shadow (256 B per row) is wider than the fp32 rows and the exact scan of short rows is cheap per
    pass, so AUTO keeps small batches on the exact scan -- without ever building the shadow for them -- and sends a batch
    through the int8 body once its exact passes (one per 8 queries) would cover about a million rows
    (profiles/r04/dim_ab_small_dims_4m.txt).  Both answers are the oracle's."""
    n = 40_000
    rows, qs = make_case(900 + d, n, d, 256, dup=True)
    for space in ("cosine", "l2"):
        eng = HipScanEngine(d, space, device=0)
        try:
            eng.append(rows)
            for nq in (3, 16, 64):  # 1, 2, 8 exact passes over 40k rows
                few = eng.search(qs[:nq], 10)
                st = eng.last_stats()
                assert st["strategy_used"] == 1, (nq, st)
                assert_knn_matches(few, oracle_knn(qs[:nq], rows, 10, space), f"auto-small-dim-exact/{space}/d{d}/nq{nq}")
            many = eng.search(qs, 10)   # 32 passes x 40k rows
            st = eng.last_stats()
            # (l2 rows of 16 columns may fail the index-wide quantisation criterion: then the exact scan keeps every batch)
            assert (st["strategy_used"] == 2 and st["bound_dtype"] == 2) or (space == "l2" and d < 32), st
            assert_knn_matches(many, oracle_knn(qs, rows, 10, space), f"auto-small-dim-int8/{space}/d{d}")
        finally:
            eng.close()


@pytest.mark.parametrize("d", [256, 100])
def test_l2_index_with_a_few_badly_quantising_rows_stays_on_the_int8_shadow(d):
    """A row with one 40-sigma component quantises at a relative error of ~0.06 and drags the 7 rows of its scale group along.
    Until round 4 the index-wide criterion (0.03) then took the WHOLE l2 index off the int8 shadow (3.3 x slower on 768
    columns, 54 x where no other filter body exists: profiles/r04/outlier_row_ab_4m.txt).  l2 bounds carry per-group errors, so a
    few such rows now cost only their groups -- the pass seeds its thresholds exactly, because the dense int8 seeding pass
    bounds with the index-wide error.  Odd rows among the seed rows, in both rounds, one of them tombstoned; every path the
    index serves; I8_ERR_L2=30 is the old behaviour; an index where 2 % of the rows are odd leaves the shadow as before."""
    n = 120_001
    rows, qs = make_case(4100 + d, n, d, 40, dup=True)
    odd = [1_001, 30_003, 90_007, 110_011]
    for i in odd:
        rows[i, i % d] = 40.0
    rows[110_011, 7] = 1.0e6   # (its 7 neighbours quantise to zeros: relative error 1, the bound is Cauchy-Schwarz's)
    eng = HipScanEngine(d, "l2", device=0, strategy="filter")
    try:
        eng.append(rows[:70_000])
        eng.append(rows[70_000:])
        eng.tombstone(np.array([90_007]))
        deleted = np.zeros(n, bool)
        deleted[90_007] = True
        got = eng.search(qs, 10)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["bound_dtype"] == 2 and st["fallback_queries"] == 0, st
        assert st["candidates_rescored"] < 400 * qs.shape[0], st   # (a stalled threshold leaves thousands per query)
        assert_knn_matches(got, oracle_knn(qs, rows, 10, "l2", deleted), f"odd-rows/l2/d{d}")
        for nq, k in ((2, 10), (9, 33), (12, 100)):   # small-batch chain, another k, a big-k pass
            sub = eng.search(qs[:nq], k)
            assert eng.last_stats()["bound_dtype"] == 2
            assert_knn_matches(sub, oracle_knn(qs[:nq], rows, k, "l2", deleted), f"odd-rows/l2/d{d}/nq{nq}k{k}")
        radius = float(got[1][:, 5].mean())
        hits = eng.range(qs[:6], radius, 64)
        want = exact_scan.range_query(qs[:6], rows, radius, "l2", deleted=deleted)
        assert all(np.array_equal(h[0], w[0]) for h, w in zip(hits, want))
        mask = (np.arange(n) % 3 != 1).astype(np.uint8)
        gm = eng.search(qs[:12], 10, mask)
        assert_knn_matches(gm, oracle_knn(qs[:12], rows, 10, "l2", deleted | (mask == 0)), f"odd-rows-mask/l2/d{d}")
        eng.set_tuning(I8_ERR_L2=30)
        eng.last_stats()   # (the statistics accumulate between reads)
        old = eng.search(qs, 10)
        assert eng.last_stats()["bound_dtype"] != 2
        assert np.array_equal(old[0], got[0]) and np.array_equal(old[1], got[1])
    finally:
        eng.close()
    # rows trickling in one at a time with a search after each: the partly filled last slab -- here with an odd row in it -- is
    # converted again by every update, and its odd group must be counted once, not 31 times (the index would leave the shadow)
    eng = HipScanEngine(d, "l2", device=0, strategy="filter")
    try:
        base = 50_016
        trickle = rows[:base + 31].copy()
        trickle[base + 1, 3] = 40.0
        eng.append(trickle[:base])
        for i in range(31):
            eng.append(trickle[base + i: base + i + 1])
            got = eng.search(qs[:9], 10)
        assert eng.last_stats()["bound_dtype"] == 2
        assert_knn_matches(got, oracle_knn(qs[:9], trickle, 10, "l2"), f"odd-rows-trickle/l2/d{d}")
    finally:
        eng.close()
    many = rows.copy()
    for i in range(0, n, 50):
        many[i, i % d] = 40.0
    eng = HipScanEngine(d, "l2", device=0, strategy="filter")
    try:
        eng.append(many)
        got = eng.search(qs[:16], 10)
        assert eng.last_stats()["bound_dtype"] != 2
        assert_knn_matches(got, oracle_knn(qs[:16], many, 10, "l2"), f"many-odd-rows/l2/d{d}")
    finally:
        eng.close()


def test_l2_batch_with_odd_query_norms_keeps_the_rest_on_the_filter():
    """l2 quantises every query of a pass with one step and bounds every lane with one pair of error coefficients (the folded
    admission test needs them in registers).  The step follows the largest component of the pass's TYPICAL queries (<= 4 x the
    median of the per-query maxima); a query whose image then measures an error > 4 x the pass's median (one 100 x the others
    clips, one 1000 x smaller has no levels left) is taken off the filter before the scans and answered by the exact fallback,
    so the other 38 keep their bounds.  kNN at k = 10, k = 100 (big-k pass) and a range query: the oracle's answer for all 40."""
    n, d = 120_001, 256
    rows, qs = make_case(3131, n, d, 40, dup=True)
    qs[7] *= 100.0
    qs[21] *= 1e-3
    eng = HipScanEngine(d, "l2", device=0, strategy="filter")
    try:
        eng.append(rows)
        got = eng.search(qs, 10)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["bound_dtype"] == 2 and st["fallback_queries"] <= 2, st
        assert st["candidates_rescored"] < 40 * 400, st   # the typical queries keep their tight bounds
        assert_knn_matches(got, oracle_knn(qs, rows, 10, "l2"), "l2 odd norms")
        got = eng.search(qs, 100)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["fallback_queries"] <= 2, st
        assert_knn_matches(got, oracle_knn(qs, rows, 100, "l2"), "l2 odd norms, k = 100")
        dmat = exact_scan.exact_distances(qs, rows, "l2")
        typical = np.delete(np.arange(40), [7, 21])
        radius = float(np.float32(np.sort(dmat[typical], axis=1)[:, 9].mean()))
        hits = eng.range(qs, radius, 256)
        st = eng.last_stats()
        assert st["strategy_used"] == 2, st
        want = exact_scan.range_query(qs, rows, radius, "l2")
        assert sum(len(w[0]) for w in want) > 100
        for i, ((gl, gd), (wl, wd)) in enumerate(zip(hits, want)):
            # (the tiny query is within the radius of EVERY row: one call ranks at most 16384 hits per query, the nearest)
            assert len(gl) == min(len(wl), 16384), (i, len(gl), len(wl))
            assert np.array_equal(gl, wl[:len(gl)]), f"range, odd norms, query {i}: {gl} vs {wl}"
            assert np.abs(gd - wd[:len(gl)]).max(initial=0.0) <= SCORE_ATOL * max(1.0, float(np.abs(wd).max(initial=0.0)))
    finally:
        eng.close()


def test_l2_offsets_plane_is_reused_across_passes_and_forgotten_on_every_mutation():
    """The l2 offsets plane (filter_l2_offsets_kernel) is kept from one pass to the next while the pass scale stays in its grid
    bin and the row pairs are untouched.  One engine, a sequence that would expose a stale plane: other queries of the same scale,
    tombstones of current winners (a dead row changes its lane group's P0), queries at other scales (up, down, back), a row-mask
    search between unmasked ones, appends, compaction -- every answer against the oracle, and the same sequence once more with
    the plane recomputed per pass (L2_OFFSET_CACHE=0)."""
    n, d, k = 50_001, 256, 10
    rows, qs = make_case(5150, n, d, 24, dup=True)
    extra = np.random.default_rng(5151).standard_normal((4_000, d)).astype(np.float32)
    more_q = np.random.default_rng(5152).standard_normal((24, d)).astype(np.float32)
    for cache in (1, 0):
        eng = HipScanEngine(d, "l2", device=0, strategy="filter")
        try:
            eng.set_tuning(L2_OFFSET_CACHE=cache)
            eng.append(rows)
            dead = np.zeros(n, dtype=bool)

            def check(q, tag, mask=None):
                got = eng.search(q, k) if mask is None else eng.search(q, k, mask)
                st = eng.last_stats()
                assert st["strategy_used"] == 2 and st["bound_dtype"] == 2, (tag, st)
                gone = dead.copy()
                if mask is not None:
                    gone |= mask == 0
                assert_knn_matches(got, oracle_knn(q, rows, k, "l2", gone), f"l2 offsets cache={cache}: {tag}")
                return got

            first = check(qs, "first pass")
            check(more_q, "other queries, same scale")
            winners = np.unique(first[0][:, 0])
            eng.tombstone(winners)
            dead[winners] = True
            check(qs, "after tombstoning the winners")
            check(qs * 1.7, "scale up")
            check(qs * 0.4, "scale down")
            check(qs, "back")
            mask = (np.random.default_rng(5153).random(n) < 0.5).astype(np.uint8)
            check(qs, "row mask", mask)
            check(more_q, "unmasked again")
            check(more_q, "row mask, other queries", mask)
            eng.append(extra)
            rows2 = np.concatenate([rows, extra])
            got = eng.search(qs, k)
            assert_knn_matches(got, oracle_knn(qs, rows2, k, "l2", np.concatenate([dead, np.zeros(extra.shape[0], bool)])),
                               f"l2 offsets cache={cache}: after an append")
            old = eng.compact()
            got = eng.search(more_q, k)
            assert_knn_matches(got, oracle_knn(more_q, rows2[old], k, "l2"), f"l2 offsets cache={cache}: after compaction")
        finally:
            eng.close()


def test_k_larger_than_live_rows_pads():
    rows, qs = make_case(51, 6, 64, 2)
    got, _ = run_hip(rows, qs, 10, "l2", "exact", deleted_mask(51, 6, 0.4))
    assert_knn_matches(got, oracle_knn(qs, rows, 10, "l2", deleted_mask(51, 6, 0.4)), "pad")


def test_get_rows_roundtrip_and_reset():
    rng = np.random.default_rng(61)
    rows = rng.standard_normal((777, 50)).astype(np.float32)
    eng = HipScanEngine(50, "l2", device=0)
    try:
        eng.append(rows[:300])
        eng.append(rows[300:])
        assert np.array_equal(eng.get_rows(0, 777), rows)
        assert np.array_equal(eng.get_rows(290, 30), rows[290:320])
        eng.reset("cosine")
        assert eng.counts() == (0, 0)
        labels, dist, counts = eng.search(rows[:2], 3)
        assert counts.tolist() == [0, 0] and (labels == -1).all()
        eng.append(rows[:10])
        got = eng.search(rows[:2], 3)
        assert_knn_matches(got, oracle_knn(rows[:2], rows[:10], 3, "cosine"), "after-reset")
    finally:
        eng.close()


def test_device_pointer_entry_runs_the_exact_fallback_for_overflowed_queries():
    """Two tight clusters of 20,000 rows: a query near a centre has more candidates than a list holds (8192), its list
    overflows and the query is re-run on the exact scan.  On the device-pointer entry that decision stays on the device
    (the ranking kernel compacts the flagged queries, the exact kernels read the list); queries far from both centres
    keep the filter's answer."""
    import torch

    rng = np.random.default_rng(404)
    n, d, nq, k = 40_000, 128, 24, 10
    centres = rng.standard_normal((2, d)).astype(np.float32)
    rows = (centres[rng.integers(0, 2, n)] + 1e-4 * rng.standard_normal((n, d))).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs[::3] = centres[0] + 1e-3 * qs[::3]
    qs[1::3] = centres[1] + 1e-3 * qs[1::3]
    eng = HipScanEngine(d, "cosine", device=0, strategy="filter")
    try:
        eng.append(rows)
        t_q = torch.from_numpy(qs).cuda()
        lab = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        eng.search_device(t_q.data_ptr(), nq, k, lab.data_ptr(), dist.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        stats = eng.last_stats()
        assert stats["strategy_used"] == 2 and stats["fallback_queries"] >= 8, stats
        assert_knn_matches((lab.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy()), oracle_knn(qs, rows, k, "cosine"),
                           "device-entry/fallback")
        got = eng.search(qs, k)  # host-pointer entry: the same decision, taken on the host after its own sync
        assert eng.last_stats()["fallback_queries"] == stats["fallback_queries"]
        assert_knn_matches(got, oracle_knn(qs, rows, k, "cosine"), "host-entry/fallback")
        # the same entry again with 24, 13 and 1 queries (full and partly filled query tiles of the fallback scan), fp64 distances
        # asked for: every call leaves the flags, the selected-query list and the partial lists ready for the next one
        want_all = oracle_knn(qs, rows, k, "cosine")
        d64 = torch.empty((nq, k), dtype=torch.float64, device="cuda")
        for m in (nq, 13, 1, nq):
            lab.fill_(-7)
            eng.search_device(t_q.data_ptr(), m, k, lab.data_ptr(), dist.data_ptr(), cnt.data_ptr(), d64.data_ptr(), stream)
            torch.cuda.synchronize()
            st = eng.last_stats()
            assert 1 <= st["fallback_queries"] <= m, (m, st)
            got_m = (lab[:m].cpu().numpy(), dist[:m].cpu().numpy(), cnt[:m].cpu().numpy())
            assert_knn_matches(got_m, tuple(w[:m] for w in want_all), f"device-entry/fallback m={m}")
            assert np.abs(d64[:m].cpu().numpy() - got_m[1]).max() < 1e-6
    finally:
        eng.close()


def test_device_pointer_entry_and_fp64_distances():
    import torch

    rows, qs = make_case(71, 5000, 128, 32)
    eng = HipScanEngine(128, "cosine", device=0, strategy="filter")
    try:
        t_rows = torch.from_numpy(rows).cuda()
        eng.append_device(t_rows.data_ptr(), rows.shape[0])
        t_q = torch.from_numpy(qs).cuda()
        k = 10
        lab = torch.empty((32, k), dtype=torch.int64, device="cuda")
        dist = torch.empty((32, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(32, dtype=torch.int32, device="cuda")
        d64 = torch.empty((32, k), dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        eng.search_device(t_q.data_ptr(), 32, k, lab.data_ptr(), dist.data_ptr(), cnt.data_ptr(), d64.data_ptr(), stream)
        torch.cuda.synchronize()
        want = oracle_knn(qs, rows, k, "cosine")
        assert_knn_matches((lab.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy()), want, "device-entry")
        want64 = np.take_along_axis(exact_scan.exact_distances(qs, rows, "cosine"), want[0], axis=1)
        assert np.abs(d64.cpu().numpy() - want64).max() < 1e-12
    finally:
        eng.close()


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("strategy,n,d", [("exact", 2000, 20), ("exact", 3000, 128), ("filter", 40000, 128), ("filter", 5000, 192)])
def test_range_query_matches_oracle(space, strategy, n, d):
    rows, qs = make_case(81, n, d, 20, dup=True)
    deleted = deleted_mask(81, n, 0.1)
    dmat = exact_scan.exact_distances(qs, rows[~deleted], space)
    radius = float(np.float32(np.sort(dmat, axis=1)[:, 9].mean()))  # ~10 hits per query
    eng = HipScanEngine(d, space, device=0, strategy=strategy)
    try:
        eng.append(rows)
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.range(qs, radius, 64)
    finally:
        eng.close()
    want = exact_scan.range_query(qs, rows, radius, space, deleted=deleted)
    assert sum(len(w[0]) for w in want) > 0
    for i, ((gl, gd), (wl, wd)) in enumerate(zip(got, want)):
        assert np.array_equal(gl, wl), f"range {space}/{strategy} query {i}: {gl} vs {wl}"
        assert np.abs(gd - wd).max(initial=0.0) <= SCORE_ATOL


@pytest.mark.parametrize("strategy,d", [("filter", 64), ("exact", 24)])
def test_range_with_more_hits_than_candidate_slots(strategy, d):
    """A radius that admits > 8192 rows for some queries: exact counts + nearest hits via the paged exact scan."""
    rows, qs = make_case(93, 40000, d, 6)
    dmat = exact_scan.exact_distances(qs, rows, "l2")
    radius = float(np.float32(np.sort(dmat, axis=1)[:, 9000].min()))  # one query has 9001 hits, the rest fewer
    eng = HipScanEngine(d, "l2", device=0, strategy=strategy)
    try:
        eng.append(rows)
        got = eng.range(qs, radius, 100)  # capacity 100 < hits: the engine retries with the reported counts
    finally:
        eng.close()
    want = exact_scan.range_query(qs, rows, radius, "l2")
    assert max(len(w[0]) for w in want) > 8192
    for (gl, gd), (wl, wd) in zip(got, want):
        assert np.array_equal(gl, wl)
        assert np.abs(gd - wd).max(initial=0.0) <= SCORE_ATOL


def test_range_packed_entry_equals_the_dense_one_and_reports_the_sizes_it_needs():
    """mlvdb_range_batch_packed (ABI 6) against mlvdb_range_batch on the same queries: same hits, packed; a total_capacity that is
    too small writes nothing but the counts and the layout (out_offsets[nq] = entries needed); a counting call passes NULL
    outputs; a per-query capacity below a query's count returns its nearest hits and MLVDB_ERR_OVERFLOW."""
    import ctypes as C

    from mlvectordb_amd import _native

    rows, qs = make_case(97, 50_000, 128, 24, dup=True)
    dmat = exact_scan.exact_distances(qs, rows, "l2")
    radius = float(np.float32(np.sort(dmat, axis=1)[:, 40].mean()))
    want = exact_scan.range_query(qs, rows, radius, "l2")
    sizes = np.array([len(w[0]) for w in want])
    assert sizes.max() > 3 * max(1, sizes.min()) and sizes.sum() > 500
    eng = HipScanEngine(128, "l2", device=0, strategy="filter")
    try:
        eng.append(rows)
        lib, h, nq = eng._lib, eng.handle, qs.shape[0]
        cap = int(sizes.max())
        dl = np.full((nq, cap), -9, dtype=np.int64)
        dd = np.zeros((nq, cap), dtype=np.float32)
        dc = np.zeros(nq, dtype=np.int64)
        assert lib.mlvdb_range_batch(h, qs.ctypes.data, nq, C.c_float(radius), cap, dl.ctypes.data, dd.ctypes.data, dc.ctypes.data) == 0
        assert np.array_equal(dc, sizes)
        total = int(sizes.sum())
        pl = np.full(total, -9, dtype=np.int64)
        pd = np.zeros(total, dtype=np.float32)
        po = np.zeros(nq + 1, dtype=np.int64)
        pc = np.zeros(nq, dtype=np.int64)
        rc = lib.mlvdb_range_batch_packed(h, qs.ctypes.data, nq, C.c_float(radius), cap, total, pl.ctypes.data, pd.ctypes.data,
                                          po.ctypes.data, pc.ctypes.data)
        assert rc == 0 and np.array_equal(pc, sizes) and np.array_equal(po, np.concatenate([[0], np.cumsum(sizes)]))
        for i in range(nq):
            assert np.array_equal(pl[po[i]:po[i + 1]], dl[i, :sizes[i]]) and np.array_equal(pd[po[i]:po[i + 1]], dd[i, :sizes[i]])
            assert np.array_equal(pl[po[i]:po[i + 1]], want[i][0])
        # too small in all: nothing written, the layout reported
        pl[:] = -9
        po[:] = -1
        rc = lib.mlvdb_range_batch_packed(h, qs.ctypes.data, nq, C.c_float(radius), cap, total - 1, pl.ctypes.data, pd.ctypes.data,
                                          po.ctypes.data, pc.ctypes.data)
        assert rc == _native.ERR_OVERFLOW and (pl == -9).all() and po[nq] == total and np.array_equal(pc, sizes)
        # counting call
        po[:] = -1
        rc = lib.mlvdb_range_batch_packed(h, qs.ctypes.data, nq, C.c_float(radius), cap, 0, None, None, po.ctypes.data, pc.ctypes.data)
        assert rc == _native.ERR_OVERFLOW and po[nq] == total and np.array_equal(pc, sizes)
        # per-query capacity 5: the nearest 5 of each, exact counts, overflow status
        rc = lib.mlvdb_range_batch_packed(h, qs.ctypes.data, nq, C.c_float(radius), 5, total, pl.ctypes.data, pd.ctypes.data,
                                          po.ctypes.data, pc.ctypes.data)
        assert rc == _native.ERR_OVERFLOW and np.array_equal(pc, sizes) and np.array_equal(np.diff(po), np.minimum(sizes, 5))
        for i in range(nq):
            assert np.array_equal(pl[po[i]:po[i + 1]], want[i][0][:5])
        # the engine wrapper: every hit (it resizes by itself), and truncated
        hits = eng.range(qs, radius, 3)
        assert len(hits) == nq and all(np.array_equal(a[0], b[0]) for a, b in zip(hits, want))
        cut = eng.range(qs, radius, 3, truncate=True)
        assert all(np.array_equal(a[0], b[0][:3]) for a, b in zip(cut, want)) and cut[-1][0].size == min(3, sizes[-1])
        assert [len(x[0]) for x in hits[2:5]] == sizes[2:5].tolist()
    finally:
        eng.close()


def test_range_capacity_overflow_is_reported_then_resolved():
    rows, qs = make_case(91, 3000, 64, 4)
    eng = HipScanEngine(64, "l2", device=0)
    try:
        eng.append(rows)
        radius = float(np.sort(exact_scan.exact_distances(qs, rows, "l2"), axis=1)[:, 49].max()) * 1.001
        got = eng.range(qs, radius, 8)  # capacity 8 < hits: the engine retries with the reported counts
        want = exact_scan.range_query(qs, rows, radius, "l2")
        for (gl, _), (wl, _) in zip(got, want):
            assert np.array_equal(gl, wl) and len(gl) >= 50
    finally:
        eng.close()


# one case per live code path of the DEFAULT library (the tuning variants of `make AB=1` are in tests/test_ab_variants.py, marked `ab`)
SCAN_VARIANTS = [
    {},                                              # defaults: the int8 body (shadow zero-padded to a multiple of 256 columns); bf16 body for d = 64 / 128
    {"MLVDB_I8": "0", "MLVDB_SHADOW": "bf16"},       # bf16 body: one 8-wave workgroup per CU, Q by LDS-DMA (rings of 4 and of 2 k-steps)
    {"MLVDB_SCAN_VAR": "237"},                       # (cosine, int8) round 2's default body: the A/B reference kept in the default library
    {"MLVDB_SCAN_XCD": "1"},                         # every XCD scans one contiguous eighth of the tile range
    {"MLVDB_SHADOW": "bf16"},                        # both shadows kept (round 2's default): int8 scans, bf16 seeding / narrow passes off
    {"MLVDB_SEED_I8": "0"},                          # int8-only index, seeding pass by the compiler kernel on the fp32 rows
    {"MLVDB_I8_PAD": "0"},                           # round 3: int8 shadow only where dim % 256 == 0
]

NARROW_CASES = [
    # space, d, nq, n  (batches of <= 64 queries: the narrow kernel, query image resident in LDS)
    ("cosine", 768, 1, 70_003), ("cosine", 768, 16, 70_003), ("cosine", 768, 17, 70_003), ("l2", 768, 32, 40_001),
    ("ip", 128, 5, 150_001), ("l2", 192, 9, 150_001), ("cosine", 64, 31, 150_001), ("ip", 1536, 3, 33_001),
    ("l2", 1536, 20, 33_001), ("cosine", 320, 2, 9_000), ("cosine", 768, 33, 70_003), ("l2", 128, 64, 150_001),
    ("ip", 1536, 40, 33_001), ("l2", 1088, 64, 20_000), ("ip", 192, 50, 9_000),
]


@pytest.mark.parametrize("space,d,nq,n,narrow", [c + (nw,) for c in NARROW_CASES for nw in (("1", "0") if c[1] in (64, 128) else ("1",))])
def test_small_batches_through_the_filter_agree_with_oracle(space, d, nq, n, narrow, monkeypatch):
    """Batches of 1..64 queries forced through the filter: the narrow kernel (MLVDB_SCAN_NARROW=1, default) and the
    same batch padded into a 256-query pass (=0) return the oracle's ids; ragged tiles, tombstones, duplicates."""
    monkeypatch.setenv("MLVDB_SCAN_NARROW", narrow)
    rows, qs = make_case(500 + d + nq, n, d, nq, dup=True)
    deleted = deleted_mask(11, n, 0.05)
    got, stats = run_hip(rows, qs, 10, space, "filter", deleted, append_chunks=3)
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0
    assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"narrow={narrow}/{space}/d{d}/nq{nq}")


@pytest.mark.parametrize("space", ["cosine", "l2", "ip"])
@pytest.mark.parametrize("nq", [9, 64, 65, 128, 129])
def test_passes_of_9_to_128_queries_compute_only_their_query_tiles(space, nq):
    """a pass of <= 64 / <= 128 queries runs the int8 body generated for 4 / 8 of the 16 query tiles (no MFMAs,
    B reads, Q transfers or admission tests for the empty ones).  Same ids as the oracle and as the padded 16-tile body
    (SCAN_NQT=16); ragged tiles, tombstones, duplicates, appends in pieces."""
    d, n = 768, (70_003 if space != "l2" else 30_001)  # (l2: the NumPy oracle is element-wise, not a GEMM)
    rows, qs = make_case(2100 + nq, n, d, nq, dup=True)
    deleted = deleted_mask(31, n, 0.05)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        for part in np.array_split(rows, 3):
            eng.append(part)
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, 10)
        st = eng.last_stats()
        assert st["strategy_used"] == 2 and st["bound_dtype"] == 2 and st["fallback_queries"] == 0, st
        assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"nqt/{space}/nq{nq}")
        eng.set_tuning(SCAN_NQT=16)
        padded = eng.search(qs, 10)
        assert np.array_equal(padded[0], got[0]) and np.array_equal(padded[1], got[1])
    finally:
        eng.close()


@pytest.mark.parametrize("space,d,nq", [("cosine", 4096, 40), ("l2", 2560, 70), ("ip", 8192, 20), ("cosine", 8192, 3),
                                        # 1-2 queries: the prefix seed with a 64 KB query in LDS; the fused finish at and beyond its ld = 2048 limit
                                        ("l2", 8192, 1), ("cosine", 4096, 2), ("ip", 2048, 1), ("cosine", 2304, 2)])
def test_wide_rows_through_the_filter_agree_with_oracle(space, d, nq):
    """dim up to the 8192 limit on the int8 path: the dense seeding pass groups fewer queries per workgroup when their int8 image
    no longer fits LDS at 64 (32 beyond ld = 2304, 16 beyond 4736), the refine cannot fuse the pruning beyond ld = 2048."""
    n = (33_000 if d <= 4096 else 17_000) if space != "l2" else 12_000  # (l2: the NumPy oracle is element-wise, not a GEMM)
    rows, qs = make_case(900 + d, n, d, nq, dup=True)
    deleted = deleted_mask(5, n, 0.05)
    got, stats = run_hip(rows, qs, 10, space, "filter", deleted, append_chunks=2)
    assert stats["strategy_used"] == 2 and stats["bound_dtype"] == 2 and stats["fallback_queries"] == 0
    assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"wide/{space}/d{d}/nq{nq}")


# (the default structure only: each step switched off in turn -- SMALL_SEED=0, SMALL_FINISH=0, SMALL_NQ=0 / 8 -- is in
# tests/test_ab_variants.py, marked `ab`: alternative paths, not live ones)
SMALL_KNOBS = [
    {},                                 # 1-2 queries: exact prefix seed + fused last refine / rescoring / ranking; one query on a corpus
                                        # of <= 11.5M / k rows: ONE scan round after an 11,520-row exact prefix (round 4)
    {"MLVDB_SMALL_BATCH": "0"},         # the scan rounds for every corpus size (what 10M rows take)
]
SMALL_CASES = [
    # space, d, nq, n, k, deleted_frac  (d % 256 == 0: the int8 shadow, which the small-batch steps need)
    ("cosine", 768, 1, 70_003, 10, 0.05), ("cosine", 768, 2, 70_003, 10, 0.05), ("l2", 768, 1, 40_001, 10, 0.3),
    ("ip", 256, 2, 150_001, 1, 0.0), ("l2", 1536, 1, 33_001, 64, 0.05), ("cosine", 256, 1, 150_001, 33, 0.9),
    ("ip", 768, 1, 3_000, 10, 0.05),     # fewer rows than the seed prefix
    ("l2", 256, 2, 9_000, 64, 0.995),    # fewer than k live rows in the prefix (and 45 in all)
    ("cosine", 768, 5, 70_003, 10, 0.05), ("l2", 768, 8, 40_001, 10, 0.05),  # 3-8 queries: default structure unless SMALL_NQ=8
    ("l2", 768, 1, 70_003, 10, 0.05), ("ip", 512, 1, 120_001, 64, 0.5),       # one round: l2 offsets, k = 64 with half the rows dead
]


@pytest.mark.parametrize("knobs", SMALL_KNOBS, ids=lambda v: ",".join(f"{k[6:]}={x}" for k, x in v.items()) or "default")
@pytest.mark.parametrize("space,d,nq,n,k,frac", SMALL_CASES)
def test_batches_of_one_or_two_queries_agree_with_oracle(space, d, nq, n, k, frac, knobs, monkeypatch):
    """The small-batch steps of the filter path (api.hip run_filter_pass: prefix_exact_kernel + filter_prefix_thr_kernel seed,
    launch_filter_finish_small) against the oracle, each switched off in turn: ragged tiles, tombstones up to 99.5 %, duplicates,
    k = 1 / 33 / 64, corpora smaller than the prefix."""
    for key, val in knobs.items():
        monkeypatch.setenv(key, val)
    rows, qs = make_case(900 + d + nq + k, n, d, nq, dup=True)
    deleted = deleted_mask(13, n, frac)
    got, stats = run_hip(rows, qs, k, space, "filter", deleted, append_chunks=3)
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0 and stats["bound_dtype"] == 2
    one_round = not knobs and nq == 1 and n > 4 * 11_520 and n * k * 6 <= 6000 * 11_520  # (api.hip run_filter_pass)
    assert (stats["scan_launches"] == 1) == (one_round or n <= 65_280), stats  # (<= 65,280 rows are one round anyway)
    assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), f"small {knobs}/{space}/d{d}/nq{nq}/k{k}")


@pytest.mark.parametrize("copies,expect_fallback", [(3_000, False), (12_000, True)])
def test_one_query_among_thousands_of_equal_rows(copies, expect_fallback):
    """A corpus holding `copies` identical rows next to the query: 3,000 of them survive every bound (the fused finish ranks a
    list longer than its LDS table: the per-wave top-k path), 12,000 overflow the candidate list (the query goes to the exact
    scan, decided on the device).  Either way the answer is the oracle's: the lowest labels among the equal rows."""
    n, d, k = 60_000, 256, 10
    rows, qs = make_case(4242, n, d, 1)
    near = qs[0] + 0.01 * np.random.default_rng(5).standard_normal(d).astype(np.float32)
    where = np.random.default_rng(6).choice(np.arange(5_000, n), size=copies, replace=False)
    rows[where] = near
    got, stats = run_hip(rows, qs, k, "cosine", "filter", None, append_chunks=2)
    assert stats["strategy_used"] == 2 and (stats["fallback_queries"] > 0) == expect_fallback, stats
    want = oracle_knn(qs, rows, k, "cosine")
    assert_knn_matches(got, want, f"equal rows x{copies}")
    assert np.array_equal(got[0][0], np.sort(where)[:k])


VARIANT_SHAPES = [("cosine", 128), ("l2", 192), ("ip", 64), ("cosine", 768), ("l2", 1536), ("ip", 768), ("l2", 256)]


@pytest.mark.parametrize("variant,space,d", [(v, sp, d) for i, v in enumerate(SCAN_VARIANTS)
                                             for j, (sp, d) in enumerate(VARIANT_SHAPES) if i == 0 or (i + j) % 3 == 0],
                         ids=lambda v: (",".join(f"{k[6:]}={x}" for k, x in v.items()) or "default") if isinstance(v, dict) else str(v))
def test_scan_kernel_variants_agree_with_oracle(variant, space, d, monkeypatch):
    """Every generated geometry of the filter scan (the variable is read per launch), on corpora with more
    tiles than resident workgroups (persistent tile loop, prefetch across tile boundaries), ragged last
    tile, tombstones and duplicated rows."""
    for key, val in variant.items():
        monkeypatch.setenv(key, val)
    n = 150_001 if d <= 192 else (70_003 if d <= 768 else 33_001)
    rows, qs = make_case(300 + d, n, d, 40, dup=True)
    deleted = deleted_mask(7, n, 0.05)
    got, stats = run_hip(rows, qs, 10, space, "filter", deleted, append_chunks=3)
    assert stats["strategy_used"] == 2 and stats["fallback_queries"] == 0
    ld8 = 0 if d in (64, 128) or (variant.get("MLVDB_I8_PAD") == "0" and d % 256) else -(-d // 256) * 256
    assert stats["bound_dtype"] == (2 if ld8 and variant.get("MLVDB_I8") != "0" else 1)
    assert_knn_matches(got, oracle_knn(qs, rows, 10, space, deleted), f"variant {variant}/{space}/d{d}")


@pytest.mark.parametrize("space,d,strategy", [("cosine", 128, "filter"), ("l2", 20, "exact"), ("ip", 192, "filter")])
def test_device_compaction_equals_a_fresh_index_of_the_survivors(space, d, strategy):
    """mlvdb_index_compact: live rows keep their order, labels renumber from 0, rows / shadow / norms move with
    them (searches on the compacted index == oracle on the surviving rows; get_rows returns them bit for bit),
    appends continue after them, and compacting twice or with nothing deleted is the identity."""
    n = 20_011
    rows, qs = make_case(500 + d, n, d, 24, dup=True)
    deleted = deleted_mask(3, n, 0.37)
    eng = HipScanEngine(d, space, device=0, strategy=strategy)
    try:
        for part in np.array_split(rows, 3):
            eng.append(part)
        eng.tombstone(np.nonzero(deleted)[0])
        old = eng.compact()
        keep = np.nonzero(~deleted)[0]
        assert np.array_equal(old, keep)
        assert eng.counts() == (keep.size, 0)
        assert np.array_equal(eng.get_rows(0, keep.size), rows[keep])
        got = eng.search(qs, 10)
        assert_knn_matches(got, oracle_knn(qs, rows[keep], 10, space), f"compacted/{space}")
        assert np.array_equal(eng.compact(), np.arange(keep.size))       # nothing deleted: identity
        extra, _ = make_case(77, 1000, d, 1)
        assert eng.append(extra) == keep.size                            # labels continue
        eng.tombstone(np.arange(0, keep.size, 2))                        # second generation of tombstones
        old2 = eng.compact()
        live2 = np.concatenate([np.arange(1, keep.size, 2), np.arange(keep.size, keep.size + 1000)])
        assert np.array_equal(old2, live2)
        both = np.concatenate([rows[keep], extra])[live2]
        assert_knn_matches(eng.search(qs, 10), oracle_knn(qs, both, 10, space), f"compacted twice/{space}")
    finally:
        eng.close()


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("strategy,n,d,nq", [("filter", 60_000, 128, 40), ("exact", 3000, 20, 5), ("filter", 9000, 768, 300),
                                             ("filter", 30_000, 256, 1), ("filter", 30_000, 768, 2)])  # (1-2 queries: the fused finish under a mask)
def test_row_mask_search_matches_oracle(space, strategy, n, d, nq):
    """mlvdb_search_batch_filtered: the exact top-k among the allowed, live rows (mask of a metadata filter),
    including masks that allow fewer than k rows and masks that allow nothing; the index itself is unchanged."""
    rows, qs = make_case(900 + d, n, d, nq, dup=True)
    deleted = deleted_mask(11, n, 0.1)
    rng = np.random.default_rng(5)
    eng = HipScanEngine(d, space, device=0, strategy=strategy)
    try:
        eng.append(rows)
        eng.tombstone(np.nonzero(deleted)[0])
        for frac in (0.5, 0.01, 3.0 / n, 0.0):
            mask = (rng.random(n) < frac).astype(np.uint8)
            got = eng.search(qs, 10, mask=mask)
            want = oracle_knn(qs, rows, 10, space, deleted | (mask == 0))
            assert_knn_matches(got, want, f"masked {frac}/{strategy}/{space}")
        assert_knn_matches(eng.search(qs, 10), oracle_knn(qs, rows, 10, space, deleted), "unmasked afterwards")
        with pytest.raises(RuntimeError):
            eng.search(qs, 10, mask=np.ones(n - 1, np.uint8))
    finally:
        eng.close()


def test_save_index_load_index_round_trip_on_device(tmp_path, monkeypatch):
    """Snapshot: rows come back off the device bit-exact, the reloaded index (ingest kernels run
    again: layout, norms, shadow) answers with the same ids and the same fp32 scores, tombstones included."""
    from mlvectordb_amd import Index, Vector, VectorDTO

    monkeypatch.setattr(Index, "_CHUNK_BYTES", 64 * 4 * 3000)  # several chunks
    rng = np.random.default_rng(77)
    n, d = 40_000, 64
    rows = rng.standard_normal((n, d), dtype=np.float32)
    vs = [Vector(values=r, metadata={}) for r in rows]
    a = Index(space="cosine", strategy="filter")
    a.add(vs, "ns")
    a.remove([vs[i].id for i in range(0, n, 17)], "ns")
    qs = rng.standard_normal((20, d), dtype=np.float32)
    want = [[(r.vector_id, r.score) for r in hits] for hits in a.search_many(qs, 10, "ns", "cosine")]
    assert a.save_index(str(tmp_path / "snap"))
    on_disk = np.fromfile(tmp_path / "snap" / "ns0.rows.f32", dtype=np.float32).reshape(n, d)
    assert (on_disk == rows).all()
    a.close()
    b = Index(space="l2", strategy="filter")
    assert b.load_index(str(tmp_path / "snap"))
    got = [[(r.vector_id, r.score) for r in hits] for hits in b.search_many(qs, 10, "ns", "cosine")]
    assert got == want
    assert b.namespace_counts("ns") == (n, len(range(0, n, 17)))
    one = b.search(VectorDTO(values=rows[5], metadata={}), 1, "ns", "cosine")[0]
    assert one.vector_id == vs[5].id
    b.close()


@pytest.mark.parametrize("space,d,nq,k", [("cosine", 768, 40, 10), ("cosine", 256, 256, 64), ("cosine", 1536, 7, 1),
                                          ("l2", 768, 40, 10), ("ip", 512, 300, 5), ("l2", 256, 3, 10)])
def test_int8_shadow_follows_appends_tombstones_compaction_and_masks(space, d, nq, k):
    """The int8 shadow is maintained lazily: rows appended, tombstoned or compacted away after a
    search must be reflected in the next one (ids == oracle each time); a row-mask search masks a copy of its row pairs."""
    n = 30_000
    rows, qs = make_case(900 + d, n, d, nq, dup=True)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        eng.append(rows[:20_000])
        got = eng.search(qs, k)
        assert eng.last_stats()["bound_dtype"] == 2
        assert_knn_matches(got, oracle_knn(qs, rows[:20_000], k, space), "i8/first")
        eng.append(rows[20_000:])  # not a multiple of the panel: the shared panel is rewritten
        deleted = deleted_mask(3, n, 0.1)
        eng.tombstone(np.nonzero(deleted)[0])
        got = eng.search(qs, k)
        assert eng.last_stats()["bound_dtype"] == 2 and eng.last_stats()["fallback_queries"] == 0
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), "i8/append+tombstone")
        # tombstone the current best hit of every query: it must disappear
        best = np.unique(got[0][:, 0])
        deleted[best] = True
        eng.tombstone(best)
        got = eng.search(qs, k)
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), "i8/tombstoned best")
        mask = (np.arange(n) % 3 != 0).astype(np.uint8)
        got = eng.search(qs, k, mask=mask)
        assert eng.last_stats()["bound_dtype"] == 2  # round 2: the mask is applied to a copy of the int8 row pairs too
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted | (mask == 0)), "i8/mask")
        got = eng.search(qs, k)  # ... and the unmasked shadow is untouched by it
        assert_knn_matches(got, oracle_knn(qs, rows, k, space, deleted), "i8/after mask")
        old = eng.compact()
        live = rows[old]
        got = eng.search(qs, k)
        assert eng.last_stats()["bound_dtype"] == 2
        assert_knn_matches(got, oracle_knn(qs, live, k, space), "i8/compacted")
        eng.reset()
        eng.append(rows[5_000:9_000])
        got = eng.search(qs, k)
        assert_knn_matches(got, oracle_knn(qs, rows[5_000:9_000], k, space), "i8/reset")
    finally:
        eng.close()


@pytest.mark.parametrize("space,dtype", [("cosine", 2), ("l2", 2), ("ip", 1)])
def test_int8_shadow_and_rows_with_outlier_components(space, dtype):
    """One scale per row: a row with one component ~30x the others has a large int8 error (its scale is set by the
    outlier, its norm is not).  Cosine bounds carry every row's own error, so only those rows are admitted more often; l2 bounds
    carry per-group errors (round 4: until then l2 left the int8 shadow here, bound_dtype 1); ip still uses the index-wide
    maximum and keeps to the fp32-in-register / bf16 bounds (relative per component).  Same ids either way."""
    rows, qs = make_case(77, 20_000, 768, 24)
    rows[123, 5] = 30.0
    rows[9_000, 700] = -25.0
    eng = HipScanEngine(768, space, device=0, strategy="filter")
    try:
        eng.append(rows)
        got = eng.search(qs, 10)
        st = eng.last_stats()
        assert st["bound_dtype"] == dtype and st["fallback_queries"] == 0
        assert st["candidates_rescored"] < 24 * 2000
        assert_knn_matches(got, oracle_knn(qs, rows, 10, space), "i8/outliers")
    finally:
        eng.close()


@pytest.mark.parametrize("space,d,copies", [("cosine", 768, 100), ("l2", 256, 40), ("cosine", 256, 3000), ("ip", 512, 70)])
def test_many_identical_rows_at_the_top_of_the_lists(space, d, copies):
    """`copies` exact copies of one row, spread over the corpus, are the nearest rows of some queries: their bounds are
    bit-identical, so the exact-threshold refine finds more equal keys at its selection boundary than it may pick (it then
    takes the ones with the lowest list indices) and the rescoring ranks long runs of equal distances by label.  The
    answer is the oracle's: the copies with the lowest labels first."""
    n, nq, k = 60_000, 24, 10
    rows, qs = make_case(4100 + copies, n, d, nq)
    rng = np.random.default_rng(copies)
    where = np.sort(rng.choice(np.arange(5000, n), copies, replace=False))  # past the seeding pass too
    rows[where] = rows[7]
    qs[1] = rows[7] + 1e-3 * qs[1]
    qs[2] = -rows[7]           # ... and the farthest rows of another
    qs[3] = rows[7]
    (gl, gd, gc), stats = run_hip(rows, qs, k, space, "filter", append_chunks=3)
    assert stats["strategy_used"] == 2 and stats["bound_dtype"] == 2
    assert_knn_matches((gl, gd, gc), oracle_knn(qs, rows, k, space), f"copies/{space}/{copies}")
    if space != "ip":
        want = np.sort(np.concatenate([[7], where]))[:k]
        assert gl[3].tolist() == want.tolist() and gl[1].tolist() == want.tolist()


@pytest.mark.parametrize("space", ["cosine", "l2", "ip"])
def test_int8_bounds_hold_for_queries_with_a_dominant_component(space):
    """A query with one dominant component quantises its small components badly (int8 error ~0.1 of its norm, and
    the norm of its int8 image exceeds 1): the error term must follow (1 + eq8), not a constant.  Rows of the same
    kind make the top of the ranking a field of near-ties."""
    rng = np.random.default_rng(5)
    n, d, nq = 40_000, 768, 48
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = (0.004 * rng.standard_normal((nq, d))).astype(np.float32)
    axes = rng.integers(0, d, nq)
    qs[np.arange(nq), axes] = 1.0
    peaky = rng.choice(n, 4000, replace=False)  # rows dominated by one of the queries' axes, with individual noise
    rows[peaky] *= 0.05
    rows[peaky, axes[rng.integers(0, nq, peaky.size)]] = rng.uniform(2.0, 3.0, peaky.size).astype(np.float32)
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    try:
        eng.append(rows)
        got = eng.search(qs, 10)
        assert eng.last_stats()["strategy_used"] == 2
        assert_knn_matches(got, oracle_knn(qs, rows, 10, space), f"i8/peaky/{space}")
    finally:
        eng.close()
