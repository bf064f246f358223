#!/usr/bin/env python3
"""Where a wave's time goes on the Protocol surface: QueryProcessor.find_similar_stream over a resident N x d namespace
(ArrayStorage, values in HBM only), with the pieces of the enrichment timed on the consumer thread and the searches on the
worker thread.

    python tools/protocol_profile.py --rows 10000000 --waves 40
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--waves", type=int, default=40)
    ap.add_argument("--space", default="cosine")
    args = ap.parse_args()
    import gc

    from mlvectordb_amd import ArrayStorage, Index, QueryProcessor, synth

    index = Index(space=args.space, device=0, capacity_hint=args.rows)
    qp = QueryProcessor(ArrayStorage(), index)
    for _, rows in synth.iter_corpus(0, args.rows, args.dim, threads=16):
        qp.upsert_arrays(rows, "bench", keep_host_copy=False)
    qs = [synth.queries(args.batch, args.dim, i) for i in range(8)]
    gc.collect()
    gc.freeze()
    qp.find_similar_many(qs[0], top_k=args.k, namespace="bench", metric=args.space)

    spans = {"search": [], "fetch_values": [], "enrich": [], "read_rows_at": []}

    def timed(obj, name, key):
        fn = getattr(obj, name)

        def wrapper(*a, **kw):
            ts = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                spans[key].append(time.perf_counter() - ts)

        setattr(obj, name, wrapper)

    eng = index._ns["bench"].engine
    timed(eng, "search", "search")
    timed(index, "fetch_values", "fetch_values")
    timed(qp, "_enrich_many", "enrich")
    timed(qp._storage, "read_rows_at", "read_rows_at")

    lat = []
    for i in range(8):
        ts = time.perf_counter()
        qp.find_similar_many(qs[i % 8], top_k=args.k, namespace="bench", metric=args.space)
        lat.append(time.perf_counter() - ts)
    for key, v in spans.items():
        if v:
            print(f"one wave at a time  {key:14s} p50 {np.median(v) * 1e3:7.3f} ms  (n = {len(v)})")
        v.clear()
    print(f"find_similar_many p50 {np.median(lat) * 1e3:.3f} ms")
    n_hits, ts = 0, time.perf_counter()
    for hits in qp.find_similar_stream((qs[i % 8] for i in range(args.waves)), top_k=args.k, namespace="bench", metric=args.space):
        n_hits += sum(len(h) for h in hits)
    t_stream = time.perf_counter() - ts
    for key, v in spans.items():
        if v:
            print(f"stream              {key:14s} p50 {np.median(v) * 1e3:7.3f} ms  mean {np.mean(v) * 1e3:7.3f} ms (n = {len(v)})")
    print(f"stream: {t_stream / args.waves * 1e3:.3f} ms per wave over {args.waves} waves, {n_hits} hits")


if __name__ == "__main__":
    main()
