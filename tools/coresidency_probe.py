#!/usr/bin/env python3
"""Does a second resident 10M x 768 index slow the first one's waves?  (bench.py keeps the cosine and the l2 engine resident
together; tools/wave_profile.py measures one engine alone.)  Host-pointer kNN waves, p50 of 30, in this order: l2 alone; l2 with
a cosine engine resident beside it; l2 again after the cosine engine was closed."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine

N, D, K = 10_000_000, 768, 10


def p50(eng, qs, n=30):
    t = []
    for i in range(n + 3):
        ts = time.perf_counter()
        eng.search(qs[i % len(qs)], K)
        t.append(time.perf_counter() - ts)
    return float(np.median(t[3:])) * 1e3


def main():
    qs = [synth.queries(256, D, i) for i in range(4)]
    l2 = HipScanEngine(D, "l2", device=0, capacity_hint=N)
    cos = HipScanEngine(D, "cosine", device=0, capacity_hint=N)
    for _, rows in synth.iter_corpus(0, N, D, threads=16):
        l2.append(rows)
        cos.append(rows)
    print(f"both resident (filled alternately, like bench.py): l2 {p50(l2, qs):.3f} ms, cosine {p50(cos, qs):.3f} ms, l2 {p50(l2, qs):.3f} ms")
    cos.close()
    print(f"cosine engine closed: l2 {p50(l2, qs):.3f} ms")
    l2.close()
    l2 = HipScanEngine(D, "l2", device=0, capacity_hint=N)
    for _, rows in synth.iter_corpus(0, N, D, threads=16):
        l2.append(rows)
    print(f"l2 alone, fresh: {p50(l2, qs):.3f} ms")
    cos = HipScanEngine(D, "cosine", device=0, capacity_hint=N)
    for _, rows in synth.iter_corpus(0, N, D, threads=16):
        cos.append(rows)
    print(f"cosine filled afterwards, both resident: l2 {p50(l2, qs):.3f} ms, cosine {p50(cos, qs):.3f} ms")


if __name__ == "__main__":
    main()
