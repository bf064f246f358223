#!/usr/bin/env python3
"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR
-o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2), as CSV on stdout -- the summary that gets committed under profiles/.
    python tools/rocpd_stats.py gpurun_out/prof_k100/p_results.db [--skip-names scatter_rows,row_norms,...] [--last N]
--last N: only the last N dispatches of every kernel (drops warm-up waves)."""
import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mlvdb::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--waves", type=int, default=0, help="divide the totals by this many waves (adds a per-wave column)")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = en - st
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    w = csv_row = None
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent" + (",us_per_wave" if args.waves else ""))
    for name, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        line = f'"{name}",{c},{t / 1e3:.1f},{t / c / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / tot:.2f}'
        if args.waves:
            line += f",{t / 1e3 / args.waves:.1f}"
        print(line)


if __name__ == "__main__":
    main()
