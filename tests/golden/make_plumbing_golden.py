#!/usr/bin/env python3
"""Golden fixture for the PLUMBING of the hot path (SURVEY 8c fixture (3), the part that can run here).

Runs, in the BUILD CONTAINER ONLY (it needs /root/reference; nothing of the reference travels -- only the JSON this
script writes), the reference's own ``QueryProcessor`` (src/mlvectordb/implementations/query_processor.py:11-82:
``insert`` / ``upsert_many`` / ``find_similar`` / ``delete``), its ``StorageEngineInMemory`` and its ``Vector`` -- their real
code, imported from /root/reference -- over THIS repo's ``Index`` with the oracle engine, and records what comes back:
keys, value types and dtypes, order, scores, the ids dropped by ``find_similar`` when storage no longer has them, and what
``delete``'s rebuild trigger does to the other namespaces.

What this pins: this repo's ``QueryProcessor`` / storage / ``Vector`` against the reference's real ones (tests/
test_plumbing_golden.py replays the same calls through this repo's classes -- oracle engine on the CPU, HIP engine under
-m gpu -- and compares with the file).  What it does NOT pin: the reference's ``Index`` (index.py:17-165) and any
arithmetic.  ``Index`` needs ``import hnswlib``; hnswlib is absent from this image and stays absent -- no module of that
name is put in its place -- so ``Index.search``'s post-processing (index.py:121-129) stays pinned by the restated reference
tests only (tests/test_reference_behaviour.py), and numeric scores stay parity-unpinned.

How the import works: the reference's package ``__init__`` (src/mlvectordb/__init__.py:19) eagerly imports its
hnswlib-backed ``Index``, which fails.  The parent packages are therefore registered as bare namespace modules (their
``__init__`` files are not executed) and the three names the reference's own modules import from the package root are
bound to the reference's own classes, as its ``__init__`` lines 11-17 do.
"""
import json
import sys
import types
from pathlib import Path
from uuid import UUID

import numpy as np

REF = Path("/root/reference")
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF))

for name, rel in (("src", "src"), ("src.mlvectordb", "src/mlvectordb"), ("src.mlvectordb.interfaces", "src/mlvectordb/interfaces"),
                  ("src.mlvectordb.implementations", "src/mlvectordb/implementations")):
    mod = types.ModuleType(name)
    mod.__path__ = [str(REF / rel)]
    sys.modules[name] = mod
from src.mlvectordb.interfaces.storage_engine import StorageEngine as _RefStorageEngine  # noqa: E402
from src.mlvectordb.interfaces.vector import VectorDTO as RefDTO  # noqa: E402

sys.modules["src.mlvectordb"].StorageEngine = _RefStorageEngine  # what src/mlvectordb/__init__.py:13 re-exports
from src.mlvectordb.implementations.query_processor import QueryProcessor as RefQueryProcessor  # noqa: E402
from src.mlvectordb.implementations.storage_engine_in_memory import StorageEngineInMemory as RefStorage  # noqa: E402

assert "hnswlib" not in sys.modules

from mlvectordb_amd import Index  # noqa: E402
from oracle.engine import OracleScanEngine  # noqa: E402


def new_processor(space="cosine"):
    return RefQueryProcessor(RefStorage(), Index(space=space, engine_factory=OracleScanEngine))


def record(hits):
    """What a caller of find_similar sees, JSON-able: ids replaced by the inserted row's tag (metadata["i"])."""
    out = []
    for h in hits:
        assert isinstance(h["id"], UUID)
        out.append({"keys": list(h.keys()), "i": h["metadata"]["i"], "metadata": dict(h["metadata"]),
                    "values": [float(x) for x in np.asarray(h["values"]).tolist()],
                    "values_type": type(h["values"]).__name__, "values_dtype": str(np.asarray(h["values"]).dtype),
                    "score": h["score"], "score_type": type(h["score"]).__name__, "id_type": type(h["id"]).__name__})
    return out


def dto(values, i, **meta):
    return RefDTO(values=values, metadata={"i": i, **meta})


cases = {}

# reference tests/test_query_processor.py:52-67 (three hand vectors, all three back, descending true cosine)
qp = new_processor()
hand = [([1, 0, 0], "A"), ([0, 1, 0], "B"), ([0.8, 0.2, 0], "C")]
qp.upsert_many([dto(v, i, label=l) for i, (v, l) in enumerate(hand)])
cases["hand_correctness"] = {"space": "cosine", "rows": [v for v, _ in hand], "labels": [l for _, l in hand],
                             "query": [0.9, 0.1, 0], "top_k": 3, "metric": "cosine",
                             "hits": record(qp.find_similar(RefDTO(values=[0.9, 0.1, 0], metadata={}), top_k=3))}

# :70-85 (namespace isolation, insert one by one)
qp = new_processor()
qp.insert(dto([1, 0, 0], 0, label="X"), namespace="alpha")
qp.insert(dto([0, 1, 0], 1, label="Y"), namespace="beta")
q = RefDTO(values=[1, 0, 0], metadata={})
cases["namespace_isolation"] = {"space": "cosine", "alpha": record(qp.find_similar(q, top_k=1, namespace="alpha")),
                                "beta": record(qp.find_similar(q, top_k=1, namespace="beta")),
                                "unknown": record(qp.find_similar(q, top_k=1, namespace="gamma")),
                                "namespaces": qp.list_namespaces()}

# :122-131 (top_k above the row count clamps)
qp = new_processor()
qp.upsert_many([dto([1, 0, 0], 0, label="A"), dto([0, 1, 0], 1, label="B")])
cases["few_vectors"] = {"space": "cosine", "top_k": 5, "hits": record(qp.find_similar(RefDTO(values=[1, 0, 0], metadata={}), top_k=5))}

# :88-105 + SURVEY quirk Q4: delete -> remove -> rebuild trigger (1/2 >= 0.2) -> rebuild from ONLY the affected namespace
qp = new_processor()
qp.upsert_many([dto([1, 0, 0], 0, label="A"), dto([0, 1, 0], 1, label="B")])
qp.upsert_many([dto([0, 0, 1], 2, label="Z")], namespace="other")
q = RefDTO(values=[1, 0, 0], metadata={})
before = qp.find_similar(q, top_k=2)
deleted = qp.delete([before[0]["id"]])
cases["delete"] = {"space": "cosine", "before": record(before), "deleted_count": len(deleted),
                   "after": record(qp.find_similar(q, top_k=2)),
                   "other_namespace_after": record(qp.find_similar(RefDTO(values=[0, 0, 1], metadata={}), top_k=1, namespace="other")),
                   "other_namespace_count_in_storage": qp.get_namespace_count("other"),
                   "default_count_in_storage": qp.get_namespace_count("default")}

# find_similar silently drops hits whose id the storage no longer has (query_processor.py:41-43): storage-only delete
qp = new_processor()
qp.upsert_many([dto([1, 0, 0], 0), dto([0.9, 0.1, 0], 1), dto([0, 1, 0], 2)])
first = qp.find_similar(q, top_k=3)
qp._storage.delete(first[0]["id"], "default")
cases["dropped_ids"] = {"space": "cosine", "before": record(first), "after": record(qp.find_similar(q, top_k=3))}

# BASELINE configs[0] shape: 10k x 128, cosine, k = 5, through upsert_many / find_similar (l2 space scored as cosine too: Q1)
rng = np.random.default_rng(1234)
rows = rng.standard_normal((10_000, 128), dtype=np.float32)
queries = np.random.default_rng(4321).standard_normal((8, 128), dtype=np.float32)
for space in ("cosine", "l2"):
    qp = new_processor(space)
    qp.upsert_many([dto(r, i) for i, r in enumerate(rows)])
    res = [qp.find_similar(RefDTO(values=qv, metadata={}), top_k=5, metric="cosine") for qv in queries]
    cases[f"random_10k_128_k5_{space}_space"] = {
        "space": space, "rows": "np.random.default_rng(1234).standard_normal((10000, 128), float32)",
        "queries": "np.random.default_rng(4321).standard_normal((8, 128), float32)", "top_k": 5, "metric": "cosine",
        "hits": [[{"i": h["metadata"]["i"], "score": h["score"]} for h in hits] for hits in res],
        "first_hit": record(res[0][:1])}

out = Path(__file__).parent / "plumbing_reference_qp.json"
out.write_text(json.dumps({"generator": "tests/golden/make_plumbing_golden.py",
                           "pins": "this repo's QueryProcessor / storage / Vector against the reference's real classes; "
                                   "NOT the reference's Index (needs hnswlib: absent, stays absent) and no arithmetic",
                           "cases": cases}, indent=1) + "\n")
print("wrote", out, {k: (len(v.get("hits", [])) if isinstance(v.get("hits"), list) else "-") for k, v in cases.items()})
