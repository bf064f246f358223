#!/usr/bin/env python3
"""A/B the filter-scan kernel variants in ONE process on one resident corpus (tuning aid).

Variants are compile-time instantiations picked per launch from the handle's tuning state (mlvdb_index_set_tuning: the
library reads the environment only when a handle is created), given as --envs "A=1,B=2;A=0" with the tuning keys
(SCAN_VAR, SCAN_NW, I8, ...; an MLVDB_ prefix is accepted).  Most variants exist only in the AB build:
    make -C mlvectordb_amd/csrc AB=1 && MLVDB_HIP_LIBRARY=mlvectordb_amd/csrc/libmlvdb_hip_ab.so python tools/scan_ab.py ...
Prints per-variant median wave time and scan-kernel GB/s (HIP events), interleaved over rounds.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--waves", type=int, default=6)
    ap.add_argument("--envs", default=";SCAN_VAR=237")
    ap.add_argument("--space", default="cosine")
    ap.add_argument("--no-check", action="store_true", help="timing diagnostics that change the answer")
    args = ap.parse_args()
    import torch

    from mlvectordb_amd import synth
    from mlvectordb_amd.engine import HipScanEngine

    eng = HipScanEngine(args.dim, args.space, device=0, capacity_hint=args.rows, strategy="filter")
    for _, rows in synth.iter_corpus(0, args.rows, args.dim, threads=16):
        eng.append(rows)
    q = torch.from_numpy(synth.queries(args.batch, args.dim)).cuda()
    k = 10
    lab = torch.empty((args.batch, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((args.batch, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(args.batch, dtype=torch.int32, device="cuda")
    eng.set_profiling(True)
    combos = [c for c in args.envs.split(";")]
    keys = sorted({kv.split("=")[0] for c in combos for kv in c.split(",") if kv})
    defaults = {key: eng.get_tuning(key) for key in keys}
    res = {c: {"wave": [], "scan": [], "bytes": 0} for c in combos}
    ref = None
    for rnd in range(args.rounds):
        for c in combos:
            eng.set_tuning(**defaults)
            eng.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in c.split(",") if kv})
            for w in range(args.waves):
                t0 = time.perf_counter()
                eng.search_device(q.data_ptr(), args.batch, k, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                st = eng.last_stats()
                if w > 0 or rnd > 0:
                    res[c]["wave"].append(dt)
                    res[c]["scan"].append(st["scan_ms"])
                    res[c]["bytes"] = st["rows_scanned"] * (args.dim * 4 + 4)
            ids = lab.cpu().numpy().copy()
            if ref is None:
                ref = ids
            assert args.no_check or np.array_equal(ids, ref), f"variant {c} changed the answer"
    print(f"rows {args.rows} dim {args.dim} batch {args.batch}")
    for c in combos:
        scan = np.median(res[c]["scan"])
        print(f"{c:40s}: wave p50 {np.median(res[c]['wave'])*1e3:7.3f} ms  min {np.min(res[c]['wave'])*1e3:7.3f}  "
              f"scan {scan:7.3f} ms = {res[c]['bytes']/scan/1e6:7.1f} GB/s")
    eng.close()


if __name__ == "__main__":
    main()
