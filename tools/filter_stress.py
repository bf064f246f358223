#!/usr/bin/env python3
"""Randomised exactness campaign for the bound filters (int8 / bf16) on one GPU: for many (distribution, shape, space, batch, k,
tombstone) draws, the filter strategy must return exactly the ids (and float32 distances) of the exact fp64 scan.
Distributions are chosen to stress the error bounds: Gaussian, clustered (near-ties), heavy-tailed, sparse, rows and
queries with dominant components, wide dynamic range of norms.  Prints one line per case and a summary; exits non-zero
on any mismatch.  Not part of the test suite (minutes of GPU time)."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mlvectordb_amd.engine import HipScanEngine

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-rows", type=int, default=200_000)
ap.add_argument("--batches", default="1,3,8,9,40,70,130,256,300", help="batch sizes drawn from")
ap.add_argument("--dims", default="256,512,768,1024,320,128,100,300,384,1000", help="dimensions drawn from (round 4: any dim >= 64 runs the int8 body on a zero-padded shadow)")
ap.add_argument("--ks", default="1,5,10,64,100,300", help="top_k drawn from (round 4: > 64 = big-k passes)")
ap.add_argument("--spaces", default="cosine,l2,ip")
ap.add_argument("--kinds", default="gauss,clustered,heavy,sparse,peaky,norms",
                help="distributions drawn from; fewodd = Gaussian with a handful of rows that carry one 10-100 sigma component "
                     "(an l2 index keeps its int8 bounds with few such rows since round 4: exact seed, per-group errors)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)


def draw(kind, n, d):
    if kind == "gauss":
        return rng.standard_normal((n, d), dtype=np.float32)
    if kind == "clustered":
        c = rng.standard_normal((32, d), dtype=np.float32)
        return (c[rng.integers(0, 32, n)] + 0.02 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    if kind == "heavy":
        return rng.standard_t(2.5, (n, d)).astype(np.float32)
    if kind == "sparse":
        x = rng.standard_normal((n, d), dtype=np.float32)
        x[rng.random((n, d)) < 0.9] = 0.0
        return x
    if kind == "peaky":
        x = (0.01 * rng.standard_normal((n, d))).astype(np.float32)
        x[np.arange(n), rng.integers(0, d, n)] = rng.uniform(0.5, 3.0, n).astype(np.float32)
        return x
    if kind == "norms":
        return (rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 6, (n, 1)))).astype(np.float32)
    if kind == "fewodd":
        x = rng.standard_normal((n, d), dtype=np.float32)
        m = int(rng.integers(1, 1 + max(1, min(60, n // 3000))))
        at = rng.integers(0, min(n, 3840) if rng.random() < 0.3 else n, m)   # (sometimes among the seed rows)
        mag = rng.uniform(10, 100, m) * rng.choice([1.0, 1.0, 1.0, 30.0, 1.0e4], m)   # (the neighbours of a 1e6 row quantise to zeros)
        x[at, rng.integers(0, d, m)] = (mag * rng.choice([-1.0, 1.0], m)).astype(np.float32)
        return x
    raise ValueError(kind)


KINDS = args.kinds.split(",")
bad = 0
t0 = time.time()
for case in range(args.cases):
    kind, qkind = KINDS[rng.integers(len(KINDS))], KINDS[rng.integers(len(KINDS))]
    d = int(rng.choice([int(x) for x in args.dims.split(",")]))
    n = int(rng.integers(20_000, args.max_rows))
    n = max(n, 2 * 300)
    nq = int(rng.choice([int(x) for x in args.batches.split(",")]))
    k = int(rng.choice([int(x) for x in args.ks.split(",")]))
    space = str(rng.choice(args.spaces.split(",")))
    rows = draw(kind, n, d)
    qs = draw(qkind, nq, d)
    if rng.random() < 0.3:  # queries that are (noisy) copies of rows: exact and near matches
        src = rng.integers(0, n, nq)
        qs = (rows[src] * (1 + 0.001 * rng.standard_normal((nq, d)))).astype(np.float32)
    i8only = d % 256 == 0 and rng.random() < 0.75     # no bf16 shadow (the default since round 3; else MLVDB_SHADOW=bf16)
    if d % 256 == 0 and not i8only:
        os.environ["MLVDB_SHADOW"] = "bf16"
    eng = HipScanEngine(d, space, device=0, strategy="filter")
    os.environ.pop("MLVDB_SHADOW", None)
    eng.append(rows[: n // 2])
    eng.append(rows[n // 2:])                          # the int8 shadow is extended lazily
    if rng.random() < 0.5:
        eng.tombstone(np.nonzero(rng.random(n) < 0.1)[0])
    mask = (rng.random(n) < rng.uniform(0.05, 0.95)).astype(np.uint8) if rng.random() < 0.3 else None  # row-mask search
    fl, fd, fc = eng.search(qs, k, mask)
    st = eng.last_stats()
    eng.set_strategy("exact")
    el, ed, ec = eng.search(qs, k, mask)
    eng.close()
    kind = kind + ("*" if i8only else "") + ("+m" if mask is not None else "")
    ok = np.array_equal(fl, el) and np.array_equal(fc, ec) and np.array_equal(fd, ed)
    bad += not ok
    print(f"case {case:3d} {'ok ' if ok else 'BAD'} rows {kind:12s} queries {qkind:9s} {space:6s} n {n:6d} d {d:4d} nq {nq:3d} k {k:2d} "
          f"dtype {st['bound_dtype']} fallback {st['fallback_queries']:3d} rescored/q {st['candidates_rescored'] / max(1, nq):8.1f}", flush=True)
print(f"{args.cases - bad} of {args.cases} cases identical to the exact scan in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
