#!/usr/bin/env python3
"""Protocol-level stream throughput (QueryProcessor.find_similar_stream) on an N x 768 corpus: tuning aid."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mlvectordb_amd import ArrayStorage, Index, QueryProcessor, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
index = Index(space="cosine", capacity_hint=n)
qp = QueryProcessor(ArrayStorage(), index)
for off, rows in synth.iter_corpus(0, n, 768, threads=16):
    qp.upsert_arrays(rows, "bench", keep_host_copy=False)
q = synth.queries(256, 768)
qp.find_similar_many(q, top_k=10, namespace="bench", metric="cosine")
eng = index._ns["bench"].engine
t = []
for _ in range(10):
    ts = time.perf_counter(); eng.search(q, 10); t.append(time.perf_counter() - ts)
print(f"engine host-io p50 {np.median(t)*1e3:.3f} ms")
t = []
for _ in range(8):
    ts = time.perf_counter(); qp.find_similar_many(q, top_k=10, namespace="bench", metric="cosine"); t.append(time.perf_counter() - ts)
print(f"find_similar_many p50 {np.median(t)*1e3:.3f} ms")
for rep in range(3):
    ts = time.perf_counter(); nh = 0
    for hits in qp.find_similar_stream((q for _ in range(20)), top_k=10, namespace="bench", metric="cosine"):
        nh += sum(len(h) for h in hits)
    print(f"stream: {(time.perf_counter()-ts)/20*1e3:.3f} ms per wave ({nh//20} hits)")
