#!/bin/bash
out=gpurun_out/r3s; mkdir -p $out
for ev in 0 1 0 1; do
MLVDB_BENCH_NO_EVENTS=$ev python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/b_$ev.json 2> $out/b_$ev.err
python -c "
import json; d=json.loads(open('$out/b_$ev.json').read().strip().splitlines()[-1]); print('no_events=$ev', d['ms_per_step'], d['p50_ms_per_wave'], (d['roofline'] or {}).get('avg_launch_ms'))"
done
