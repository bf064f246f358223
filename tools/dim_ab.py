#!/usr/bin/env python3
"""A/B for dimensions that are not multiples of 256: the int8 body on a shadow zero-padded to the next
multiple of 256 columns against what served such a corpus before -- the bf16 body (dim % 64 == 0: MLVDB_SHADOW=bf16 + I8=0)
or the exact fp64 scan (any other dim).  One engine per variant (the shadow is a creation-time decision), same rows, same
queries; prints the wave time (device-resident, synchronised), the scan-kernel time and checks that the ids agree."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dims", default="384,1000,100,300,640")
    ap.add_argument("--batches", default="256,32,1")
    ap.add_argument("--space", default="cosine")
    ap.add_argument("--waves", type=int, default=12)
    args = ap.parse_args()
    import torch

    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine

    k = 10
    for d in [int(x) for x in args.dims.split(",")]:
        variants = [("int8 shadow padded to %d" % (-(-d // 256) * 256), {}, "auto")]
        if d % 64 == 0:
            variants.append(("bf16 body (round 3)", {"MLVDB_SHADOW": "bf16", "MLVDB_I8": "0"}, "auto"))
        else:
            variants.append(("exact fp64 scan (round 3)", {}, "exact"))
        ref = {}
        for name, env, strategy in variants:
            for key, val in env.items():
                os.environ[key] = val
            eng = HipScanEngine(d, args.space, device=0, capacity_hint=args.rows, strategy=strategy)
            for key in env:
                os.environ.pop(key)
            for _, rows in synth.iter_corpus(0, args.rows, d, threads=16):
                eng.append(rows)
            eng.set_profiling(True)
            for batch in [int(b) for b in args.batches.split(",")]:
                q = torch.from_numpy(synth.queries(batch, d)).cuda()
                lab = torch.empty((batch, k), dtype=torch.int64, device="cuda")
                dst = torch.empty((batch, k), dtype=torch.float32, device="cuda")
                cnt = torch.empty(batch, dtype=torch.int32, device="cuda")
                t = []
                for i in range(args.waves + 3):
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    eng.search_device(q.data_ptr(), batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
                    torch.cuda.synchronize()
                    t.append(time.perf_counter() - ts)
                    if i == 2:
                        eng.last_stats()
                st = eng.last_stats()
                ids = lab.cpu().numpy().copy()
                same = np.array_equal(ids, ref.setdefault(batch, ids))
                print(f"d {d:5d} batch {batch:4d} {name:32s}: wave p50 {np.median(t[3:]) * 1e3:7.3f} ms, scan kernels {st['scan_ms'] / args.waves:7.3f} ms, "
                      f"strategy {st['strategy_used']} bound dtype {st['bound_dtype']}{'' if same else '  IDS DIFFER'}", flush=True)
            eng.close()


if __name__ == "__main__":
    main()
