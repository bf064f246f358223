"""Randomised model checks of the array-backed host structures (CPU, hypothesis):

* ``IdTable`` against the two dicts the reference keeps (index.py:21-22,56-63,76-81);
* ``MultiDeviceEngine`` (logical shards over oracle engines) against ONE oracle engine through arbitrary sequences of
  appends, tombstones, compactions, masked searches and row reads -- labels, ids and fp64-ranked answers must agree.
"""
import uuid

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from mlvectordb_amd.idtable import IdTable, uuids_to_bytes
from mlvectordb_amd.multi_device import MultiDeviceEngine
from oracle.engine import OracleScanEngine

_ops = st.lists(st.tuples(st.sampled_from(["add_obj", "add_raw", "kill", "lookup", "take"]), st.integers(0, 2**31 - 1)),
                min_size=1, max_size=25)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(_ops)
def test_idtable_behaves_like_the_two_dicts(ops):
    # model: the reference's two dicts -- {label: id} of the live rows and {id: label}, whose entry is overwritten when an
    # id is added again (index.py:62) and popped when it is removed (index.py:76-81): an older row of a removed id is
    # never found again
    t, live, known, u2l = IdTable(), {}, [], {}

    def model_lookup(u):
        return u2l.get(u, -1)

    for op, seed in ops:
        rng = np.random.default_rng(seed)
        if op in ("add_obj", "add_raw"):
            batch = [uuid.UUID(bytes=rng.bytes(16)) for _ in range(int(rng.integers(1, 12)))]
            if known and rng.random() < 0.3:
                batch[0] = known[int(rng.integers(len(known)))]   # the same id again: the newest row wins (index.py:62)
            rows_before = t.n
            first = t.append_uuids(batch) if op == "add_obj" else t.append_raw(uuids_to_bytes(batch))
            assert first == rows_before and t.n == rows_before + len(batch)
            live.update({first + i: u for i, u in enumerate(batch)})
            u2l.update({u: first + i for i, u in enumerate(batch)})
            known += batch
        elif op == "kill" and known:
            victims = [known[int(i)] for i in rng.integers(0, len(known), 3)] + [uuid.UUID(bytes=rng.bytes(16))]
            labels = t.lookup(victims)
            assert labels.tolist() == [model_lookup(v) for v in victims]
            hit = np.unique(labels[labels >= 0])
            t.kill(hit)
            for lab in hit.tolist():
                del u2l[live.pop(lab)]
            assert set(t.dead_labels().tolist()) == set(range(t.n)) - set(live)
        elif op == "lookup" and known:
            probe = [known[int(i)] for i in rng.integers(0, len(known), 5)] + [None, "not-an-id"]
            assert t.lookup(probe).tolist() == [model_lookup(v) for v in probe]
            assert t.lookup_raw(uuids_to_bytes(probe[:5])).tolist() == [model_lookup(v) for v in probe[:5]]
        elif op == "take" and live:
            keep = np.array(sorted(live), dtype=np.int64)
            t = t.take(keep)
            live = {new: live[old] for new, old in enumerate(keep.tolist())}
            u2l = {u: lab for lab, u in live.items()}  # a rebuild re-adds the survivors in order: the newest row wins again
    got = t.uuids_at(np.arange(-1, t.n + 1))
    assert got[0] is None and got[-1] is None
    assert {i: u for i, u in enumerate(got[1:-1].tolist()) if u is not None} == live


_eng_ops = st.lists(st.tuples(st.sampled_from(["append", "tombstone", "compact", "search", "masked", "rows"]),
                              st.integers(0, 2**31 - 1)), min_size=2, max_size=14)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.sampled_from(["l2", "cosine", "ip"]), st.integers(2, 5), _eng_ops)
def test_logical_shards_equal_one_engine_through_any_history(space, shards, ops):
    d = 6
    many = MultiDeviceEngine(d, space, [0] * shards, shard_factory=lambda dev: OracleScanEngine(d, space))
    one = OracleScanEngine(d, space)
    try:
        for op, seed in ops:
            rng = np.random.default_rng(seed)
            total = one.counts()[0]
            if op == "append" or total == 0:
                rows = rng.standard_normal((int(rng.integers(1, 30)), d)).astype(np.float32)
                if total and rng.random() < 0.4:
                    rows[0] = one.get_rows(int(rng.integers(total)), 1)[0]     # an exact duplicate somewhere else
                assert many.append(rows) == one.append(rows) == total
            elif op == "tombstone":
                labels = rng.integers(0, total, int(rng.integers(1, 6)))
                assert many.tombstone(labels) == one.tombstone(labels)
            elif op == "compact":
                assert np.array_equal(many.compact(), one.compact())
            elif op in ("search", "masked"):
                qs = rng.standard_normal((3, d)).astype(np.float32)
                k = int(rng.integers(1, 9))
                mask = (rng.random(total) < 0.6).astype(np.uint8) if op == "masked" else None
                gl, gd, gc = many.search(qs, k, mask)
                wl, wd, wc = one.search(qs, k, mask)
                assert np.array_equal(gl, wl) and np.array_equal(gc, wc) and np.allclose(gd, wd, rtol=0, atol=1e-6, equal_nan=True)
            else:
                pick = rng.integers(0, total, 4)
                assert np.array_equal(many.get_rows_at(pick), one.get_rows_at(pick))
            assert many.counts() == one.counts()
    finally:
        many.close()
