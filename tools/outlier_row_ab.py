#!/usr/bin/env python3
"""One badly quantising row and an l2 / ip index: int8 bounds are used while the index-wide largest relative row error stays
under a limit (tuning I8_ERR_L2 / I8_ERR_IP, thousandths; typical Gaussian rows: 8-15).  A row with one 40-sigma component
quantises at ~60: with the limit at 30 the whole index leaves the int8 shadow (fp32 rows converted in registers, or the exact
scan where dim % 64 != 0).  l2 bounds carry per-row-group errors since round 4, so only that row's group should pay.
Prints the wave time for a clean corpus and one with a few such rows, at several limits; ids are checked against the exact scan."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dims", default="768,300")
    ap.add_argument("--spaces", default="l2,ip")
    ap.add_argument("--limits", default="30,250")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--waves", type=int, default=10)
    ap.add_argument("--k", type=int, default=10, help="top_k (> 64: big-k passes -- their dense seed is the open item of DESIGN 10)")
    ap.add_argument("--odd-value", type=float, default=40.0, help="the component planted in the odd rows (1e6: their neighbours quantise to zeros)")
    args = ap.parse_args()
    import torch

    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine

    k = args.k
    for d in [int(x) for x in args.dims.split(",")]:
        pieces = [rows for _, rows in synth.iter_corpus(0, args.rows, d, threads=16)]
        odd = np.random.default_rng(5).integers(0, args.rows, 5)
        for space in args.spaces.split(","):
            for corpus in ("clean", f"5 rows with a {args.odd_value:g}-sigma component"):
                eng = HipScanEngine(d, space, device=0, capacity_hint=args.rows)
                off = 0
                for rows in pieces:
                    if corpus != "clean":
                        rows = rows.copy()
                        for r in odd[(odd >= off) & (odd < off + len(rows))]:
                            rows[r - off, int(r) % d] = args.odd_value
                    eng.append(rows)
                    off += len(rows)
                eng.set_profiling(True)
                q = torch.from_numpy(synth.queries(args.batch, d)).cuda()
                lab = torch.empty((args.batch, k), dtype=torch.int64, device="cuda")
                dst = torch.empty((args.batch, k), dtype=torch.float32, device="cuda")
                cnt = torch.empty(args.batch, dtype=torch.int32, device="cuda")
                eng.set_strategy("exact")
                eng.search_device(q.data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
                torch.cuda.synchronize()
                want = lab.cpu().numpy().copy()
                eng.set_strategy("auto")
                for lim in [int(x) for x in args.limits.split(",")]:
                    eng.set_tuning(**{"I8_ERR_L2": lim, "I8_ERR_IP": lim})
                    t = []
                    for i in range(args.waves + 3):
                        torch.cuda.synchronize()
                        ts = time.perf_counter()
                        eng.search_device(q.data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
                        torch.cuda.synchronize()
                        t.append(time.perf_counter() - ts)
                        if i == 2:
                            eng.last_stats()
                    st = eng.last_stats()
                    same = np.array_equal(lab.cpu().numpy(), want)
                    print(f"d {d:4d} {space:3s} {corpus:34s} limit {lim:4d}: wave p50 {np.median(t[3:]) * 1e3:7.3f} ms, strategy {st['strategy_used']} "
                          f"bound dtype {st['bound_dtype']}, rescored/q {st['candidates_rescored'] / args.waves / args.batch:7.1f}, "
                          f"fallbacks {st['fallback_queries'] / args.waves:5.1f}{'' if same else '  IDS DIFFER'}", flush=True)
                eng.close()


if __name__ == "__main__":
    main()
