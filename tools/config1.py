#!/usr/bin/env python3
"""BASELINE configs[0]: 10k x 128 random vectors, cosine k=5, through QueryProcessor.find_similar.

The same Protocol-level flow (upsert_many -> find_similar, reference query_processor.py:19-49) is timed on the HIP
engine and on the NumPy oracle engine (the CPU plumbing case the config names; the reference itself cannot run here,
hnswlib is absent), and the hits are compared: ids identical, scores within 1e-5.
Prints one JSON object."""
import json, os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mlvectordb_amd import Index, InMemoryStorage, QueryProcessor, VectorDTO
from oracle.engine import OracleScanEngine

N, D, K, NQ = 10_000, 128, 5, 200
rng = np.random.default_rng(1234)
rows = rng.standard_normal((N, D), dtype=np.float32)
queries = np.random.default_rng(4321).standard_normal((NQ, D), dtype=np.float32)


def run(index):
    qp = QueryProcessor(InMemoryStorage(), index)
    t0 = time.perf_counter()
    qp.upsert_many([VectorDTO(values=r, metadata={"i": i}) for i, r in enumerate(rows)], namespace="bench")
    load = time.perf_counter() - t0
    lat, out = [], []
    for i in range(NQ):
        t0 = time.perf_counter()
        hits = qp.find_similar(VectorDTO(values=queries[i], metadata={}), top_k=K, namespace="bench", metric="cosine")
        lat.append(time.perf_counter() - t0)
        out.append([(h["metadata"]["i"], h["score"]) for h in hits])
    lat = np.array(lat[20:])
    return {"upsert_s": round(load, 3), "p50_ms": round(float(np.median(lat)) * 1e3, 4), "p99_ms": round(float(np.quantile(lat, 0.99)) * 1e3, 4),
            "qps": round(1.0 / float(np.mean(lat)), 1)}, out


res = {"config": "BASELINE configs[0]: 10k x 128, cosine, k=5, batch 1 through QueryProcessor.find_similar", "host_cores": os.cpu_count()}
res["hip"], got = run(Index(space="cosine"))
res["numpy_oracle_engine"], want = run(Index(space="cosine", engine_factory=OracleScanEngine))
res["ids_equal"] = all([g[0] for g in a] == [w[0] for w in b] for a, b in zip(got, want))
res["max_abs_score_err"] = max(abs(g[1] - w[1]) for a, b in zip(got, want) for g, w in zip(a, b))
print(json.dumps(res))
