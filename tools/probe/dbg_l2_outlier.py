"""Probe (round 4): where an l2 pass loses its tightness when a few rows quantise badly and the index-wide limit is lifted
(I8_ERR_L2=150).  DEBUG_ENTRIES prints the entries each scan launch appends."""
import sys, numpy as np
sys.path.insert(0, '.')
from mlvectordb_amd.engine import HipScanEngine
rng = np.random.default_rng(3)
n, d = 300_000, 768
rows = rng.standard_normal((n, d), dtype=np.float32)
qs = rng.standard_normal((256, d), dtype=np.float32)
cases = {"clean": (), "odd rows in round 1 only": (10_001, 20_003, 50_007), "one odd row in round 2": (250_007,),
         "odd rows in round 2": (100_001, 200_003, 250_007), "odd row among the seed rows": (1_001,)}
for corpus, odd in cases.items():
    r = rows.copy()
    for i in odd:
        r[i, i % d] = 40.0
    eng = HipScanEngine(d, "l2", device=0)
    eng.append(r)
    eng.set_tuning(I8_ERR_L2=150, DEBUG_ENTRIES=1)
    print(f"--- {corpus}", flush=True)
    l, dd, c = eng.search(qs, 10)
    st = eng.last_stats()
    print({k: st[k] for k in ("strategy_used", "bound_dtype", "candidates_rescored", "fallback_queries", "scan_launches")}, flush=True)
    eng.close()
