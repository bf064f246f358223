#!/bin/bash
out=gpurun_out/r3u; mkdir -p $out
bash tools/gpu_suite.sh r3u || exit 1
python - > $out/batch1_host_io.txt 2> $out/batch1.err <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from mlvectordb_amd import synth
from mlvectordb_amd.engine import HipScanEngine
eng = HipScanEngine(768, "cosine", device=0, capacity_hint=1_000_000)
for _, rows in synth.iter_corpus(0, 1_000_000, 768, threads=16):
    eng.append(rows)
q = synth.queries(8, 768)
for nq in (1, 8):
    lat = []
    for i in range(80):
        ts = time.perf_counter(); eng.search(q[:nq], 10); lat.append(time.perf_counter() - ts)
    print(f"host-pointer entry, 1M x 768, nq {nq}: p50 {np.median(lat[10:])*1e3:.4f} ms")
PY
cat $out/batch1_host_io.txt
