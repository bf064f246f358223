#!/bin/bash
# bench.py with both timed regions; timing events with / without the system-scope fence
O=gpurun_out/r4u; mkdir -p $O
for f in 0 1 0 1; do
  MLVDB_EVENT_FENCE=$f timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/bench_fence$f.json 2> $O/bench.err; echo "fence=$f rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r4u/bench_fence$f.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','other_wave_mode','p50_ms_per_wave')}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
done
