#!/usr/bin/env python3
"""Small query batches (1..64) on an N x 768 cosine corpus: exact fp32/fp64 scan vs the filter path with the narrow
kernel (query image resident in LDS, streams the bf16 shadow) vs the same batch padded into a 256-query pass.
Prints p50 call latency (device-pointer entry, synchronised) and checks that all three return the same ids."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--space", default="cosine")
ap.add_argument("--batches", default="1,2,4,8,12,16,32,64")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--modes", default="exact,narrow,pass256,auto")
ap.add_argument("--spin", action="store_true", help="wait for the call by polling the stream (hipStreamQuery) instead of hipStreamSynchronize")
args = ap.parse_args()
N, D, K = args.rows, args.dim, 10
eng = HipScanEngine(D, args.space, device=0, capacity_hint=N)
for _, rows in synth.iter_corpus(0, N, D, threads=16):
    eng.append(rows)
print(f"corpus {N} x {D} {args.space}", flush=True)
batches = [int(b) for b in args.batches.split(",")]
qmax = max(batches)
q = torch.from_numpy(synth.queries(qmax, D)).cuda()
lab = torch.empty((qmax, K), dtype=torch.int64, device="cuda"); dst = torch.empty((qmax, K), dtype=torch.float32, device="cuda")
cnt = torch.empty(qmax, dtype=torch.int32, device="cuda")
ALL = {"exact": ("exact", "exact", "1"), "narrow": ("filter narrow", "filter", "1"), "pass256": ("filter 256-pass", "filter", "0"), "auto": ("auto", "auto", "1")}
modes = [ALL[m] for m in args.modes.split(",")]
for nq in batches:
    ref = None
    line = [f"nq {nq:3d}:"]
    for name, strat, narrow in modes:
        eng.set_tuning(SCAN_NARROW=int(narrow))
        eng.set_strategy(strat)
        lat = []
        for i in range(args.iters + 5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.search_device(q.data_ptr(), nq, K, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
            if args.spin:
                st = torch.cuda.current_stream()
                while not st.query():
                    pass
            else:
                torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        ids = lab[:nq].cpu().numpy().copy()
        if ref is None:
            ref = ids
        same = bool((ids == ref).all())
        line.append(f"{name} {np.median(lat[5:]) * 1e3:.3f} ms{'' if same else ' IDS DIFFER'}")
    print("  ".join(line), flush=True)
