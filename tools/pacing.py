#!/usr/bin/env python3
"""How a 10M x 768 wave's duration responds to a short idle before it: K synchronised waves with a host busy-wait of X us
between them; prints the wall time per wave INCLUDING the pause, and the wave alone.  (The synchronised loop of bench.py is X = 0.)"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine

N, D, K, B = 10_000_000, 768, 10, 256
eng = HipScanEngine(D, "cosine", device=0, capacity_hint=N)
for _, rows in synth.iter_corpus(0, N, D, threads=16):
    eng.append(rows)
q = torch.from_numpy(synth.queries(B, D)).cuda()
lab = torch.empty((B, K), dtype=torch.int64, device="cuda"); dst = torch.empty((B, K), dtype=torch.float32, device="cuda")
cnt = torch.empty(B, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
for _ in range(5):
    eng.search_device(q.data_ptr(), B, K, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
st.synchronize()
for rep in range(2):
    for pause_us in (0, 25, 50, 100, 200, 400, 0):
        waves, t_all0 = [], time.perf_counter()
        for i in range(40):
            t0 = time.perf_counter()
            eng.search_device(q.data_ptr(), B, K, lab.data_ptr(), dst.data_ptr(), cnt.data_ptr(), 0, 0)
            st.synchronize()
            t1 = time.perf_counter()
            waves.append(t1 - t0)
            while time.perf_counter() - t1 < pause_us * 1e-6:
                pass
        total = (time.perf_counter() - t_all0) / 40
        print(f"pause {pause_us:4d} us: wave alone p50 {np.median(waves) * 1e3:.3f} ms, per wave including the pause {total * 1e3:.3f} ms", flush=True)
