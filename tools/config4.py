#!/usr/bin/env python3
"""BASELINE configs[3]: 10M x 768, k=10 euclidean (squared-l2 space) kNN + range query, batch 256, 1 x MI355X.

Radius = mean 10th-neighbour distance of the batch (~10 hits per query).  Prints one JSON
line with wave times and the parity checks (filter vs exact fp64 scan on a query subset; range hits
vs the kNN distances).  Not the headline bench: results are committed under profiles/.
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--waves", type=int, default=10)
    args = ap.parse_args()
    import torch
    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine

    n, d, b, k = args.rows, args.dim, args.batch, 10
    eng = HipScanEngine(d, "l2", device=0, capacity_hint=n)
    t0 = time.perf_counter()
    for _, rows in synth.iter_corpus(0, n, d, threads=16):
        eng.append(rows)
    load_s = time.perf_counter() - t0
    q = synth.queries(b, d)
    labels, dist, counts = eng.search(q, k)                      # warm-up + reference answer
    eng.set_strategy("exact")
    le, de, _ = eng.search(q[:8], k)
    eng.set_strategy("auto")
    knn_ok = bool(np.array_equal(labels[:8], le) and np.array_equal(dist[:8], de))
    t = []
    for _ in range(args.waves):
        torch.cuda.synchronize(); ts = time.perf_counter(); eng.search(q, k); t.append(time.perf_counter() - ts)
    knn_ms = float(np.median(t)) * 1e3
    radius = float(dist[:, k - 1].mean())
    hits = eng.range(q, radius, 8192)                             # warm-up + answer
    t = []
    for _ in range(args.waves):
        torch.cuda.synchronize(); ts = time.perf_counter(); eng.range(q, radius, 8192); t.append(time.perf_counter() - ts)
    range_ms = float(np.median(t)) * 1e3
    # every range hit is within the radius, sorted, and is a prefix-consistent superset/subset of the kNN answer
    ok = True
    nh = []
    for i, (hl, hd) in enumerate(hits):
        nh.append(len(hl))
        ok &= bool((hd <= np.float32(radius)).all() and (np.diff(hd) >= 0).all())
        m = min(len(hl), k)
        ok &= bool(np.array_equal(hl[:m], labels[i, :m]))
        if len(hl) < k:
            ok &= bool(dist[i, len(hl)] > np.float32(radius))
    alg = n * d * 4 + n * 4
    print(json.dumps({
        "config": f"BASELINE configs[3]: {n} x {d}, l2 (squared) kNN k={k} + range, batch {b}",
        "knn_ms_per_wave_host_inclusive": round(knn_ms, 3), "knn_qps": round(b / knn_ms * 1e3, 1),
        "knn_whole_wave_frac_of_8TBs": round(alg / (knn_ms * 1e-3) / 8e12, 4),
        "range_ms_per_wave_host_inclusive": round(range_ms, 3), "range_qps": round(b / range_ms * 1e3, 1),
        "range_whole_wave_frac_of_8TBs": round(alg / (range_ms * 1e-3) / 8e12, 4),
        "radius_squared_l2": radius, "mean_hits_per_query": float(np.mean(nh)), "max_hits": int(max(nh)),
        "parity": {"knn_filter_equals_exact_scan": knn_ok, "range_consistent_with_knn": ok},
        "load_s": round(load_s, 1)}))


if __name__ == "__main__":
    main()
