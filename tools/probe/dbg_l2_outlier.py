"""Probe (round 4): where a pass loses its tightness when a few rows quantise badly (a 40-sigma component) -- DEBUG_ENTRIES
prints the entries each scan launch appends.  `python tools/probe/dbg_l2_outlier.py [l2|ip|cosine] [dim]`; the l2 result that
led to the exact seed is profiles/r04/outlier_row_probe_l2_300k.txt."""
import sys, numpy as np
sys.path.insert(0, '.')
from mlvectordb_amd.engine import HipScanEngine
rng = np.random.default_rng(3)
SPACE = sys.argv[1] if len(sys.argv) > 1 else "l2"
n, d = 300_000, int(sys.argv[2]) if len(sys.argv) > 2 else 768
rows = rng.standard_normal((n, d), dtype=np.float32)
qs = rng.standard_normal((256, d), dtype=np.float32)
cases = {"clean": (), "odd rows in round 1 only": (10_001, 20_003, 50_007), "one odd row in round 2": (250_007,),
         "odd rows in round 2": (100_001, 200_003, 250_007), "odd row among the seed rows": (1_001,)}
for corpus, odd in cases.items():
    r = rows.copy()
    for i in odd:
        r[i, i % d] = 40.0
    eng = HipScanEngine(d, SPACE, device=0)
    eng.append(r)
    eng.set_tuning(DEBUG_ENTRIES=1)
    print(f"--- {corpus}", flush=True)
    l, dd, c = eng.search(qs, 10)
    st = eng.last_stats()
    print({k: st[k] for k in ("strategy_used", "bound_dtype", "candidates_rescored", "fallback_queries", "scan_launches")}, flush=True)
    eng.close()
