#!/usr/bin/env python3
"""Batch-1 exact scan (BASELINE configs[1]: 1M x 768): latency and scan-kernel GB/s for a few grid sizes."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine

N, D, K = 1_000_000, 768, 10
eng = HipScanEngine(D, "cosine", device=0, capacity_hint=N)
for _, rows in synth.iter_corpus(0, N, D, threads=16):
    eng.append(rows)
q = torch.from_numpy(synth.queries(8, D)).cuda()
lab = torch.empty((8, K), dtype=torch.int64, device="cuda"); dst = torch.empty((8, K), dtype=torch.float32, device="cuda"); cnt = torch.empty(8, dtype=torch.int32, device="cuda")
eng.set_profiling(True)
# eng.set_tuning(EXACT_NT=0) selects plain loads for batch 1
for nq in (1, 2, 4, 8):
    for nblk in ("256", "512"):
        eng.set_tuning(EXACT_NBLK=int(nblk))
        lat = []
        for i in range(40):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.search_device(q.data_ptr(), nq, K, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
            torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
            if i == 9: eng.last_stats()
        st = eng.last_stats()
        scan = st["scan_ms"] / 30
        print(f"nq {nq} nblk {nblk:5s}: p50 {np.median(lat[10:])*1e3:.3f} ms  scan {scan:.3f} ms = {N*(D*4+4)/scan/1e6:.0f} GB/s")
