#!/usr/bin/env python3
"""One kind of query wave, repeated, on a resident N x d corpus -- the target of `rocprofv3 --kernel-trace --stats`
when a per-kernel breakdown of that wave is wanted (kNN at some top_k, or a range query at the mean k-th neighbour distance).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k100 -- python3 tools/wave_profile.py --k 100
    python tools/wave_profile.py --space l2 --range          # prints the wave time only

Prints p50 / min wave time (device-resident for kNN, host-pointer entry for range) and the call's statistics."""
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
    ap.add_argument("--space", default="cosine")
    ap.add_argument("--range", action="store_true", help="range query at the mean k-th neighbour distance instead of kNN")
    ap.add_argument("--waves", type=int, default=12)
    ap.add_argument("--tune", default="", help="KEY=VAL,KEY=VAL tuning of the handle")
    ap.add_argument("--check", action="store_true", help="compare with the exact scan")
    ap.add_argument("--host", action="store_true", help="kNN through the host-pointer entry (queries in, results out over PCIe) as well")
    args = ap.parse_args()
    import torch

    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine

    eng = HipScanEngine(args.dim, args.space, device=0, capacity_hint=args.rows)
    for _, rows in synth.iter_corpus(0, args.rows, args.dim, threads=16):
        eng.append(rows)
    if args.tune:
        eng.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",")})
    qh = [synth.queries(args.batch, args.dim, i) for i in range(4)]
    k = args.k
    t = []
    if args.range:
        _, dist, _ = eng.search(qh[0], k)
        radius = float(dist[:, k - 1].mean())
        hits = eng.range(qh[0], radius, 8192)
        eng.last_stats()
        for i in range(args.waves):
            ts = time.perf_counter()
            eng.range(qh[0], radius, 8192)
            t.append(time.perf_counter() - ts)
        st = eng.last_stats()
        print(f"range radius {radius:.4f}: mean hits {np.mean([len(h[0]) for h in hits]):.1f}")
        if args.check:
            eng.set_strategy("exact")
            want = eng.range(qh[0], radius, 8192)
            print("hits equal the exact range scan:", all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(hits, want)))
    else:
        qd = [torch.from_numpy(q).cuda() for q in qh]
        lab = torch.empty((args.batch, k), dtype=torch.int64, device="cuda")
        dst = torch.empty((args.batch, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(args.batch, dtype=torch.int32, device="cuda")
        for i in range(3):
            eng.search_device(qd[i % 4].data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
        torch.cuda.synchronize()
        eng.last_stats()
        for i in range(args.waves):
            torch.cuda.synchronize()
            ts = time.perf_counter()
            eng.search_device(qd[i % 4].data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
            torch.cuda.synchronize()
            t.append(time.perf_counter() - ts)
        st = eng.last_stats()
        if args.host:
            th = []
            for i in range(args.waves + 3):
                ts = time.perf_counter()
                eng.search(qh[i % 4], k)
                th.append(time.perf_counter() - ts)
            th = np.array(th[3:]) * 1e3
            print(f"host-pointer entry: wave p50 {np.median(th):.3f} ms, min {th.min():.3f} ms")
            eng.last_stats()
        if args.check:
            eng.search_device(qd[0].data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
            torch.cuda.synchronize()
            got = lab.cpu().numpy().copy()
            eng.set_strategy("exact")
            want, _, _ = eng.search(qh[0], k)
            print("ids equal the exact scan:", bool(np.array_equal(got, want)))
    t = np.array(t) * 1e3
    print(f"{args.rows} x {args.dim} {args.space} batch {args.batch} {'range' if args.range else f'k={k}'}: wave p50 {np.median(t):.3f} ms, "
          f"min {t.min():.3f} ms; per wave: {st['scan_launches'] / args.waves:.1f} scan launches, "
          f"{st['candidates_rescored'] / args.waves / args.batch:.1f} rows rescored per query, {st['fallback_queries']} fallbacks")
    eng.close()


if __name__ == "__main__":
    main()
