#!/bin/bash
# rocprofv3 kernel stats of the bench command with the synchronised timed loop
O=gpurun_out/r4z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-160; fi
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python3 - "$t" <<'PY'
import csv, sys, statistics
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'filter_scan_asm_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
big=[x for x in d if x>300]
print("scan launches", len(d), "largest-round launches", len(big), "mean of all %.1f us"%statistics.mean(d))
PY
fi
grep -o '"ms_per_step": [0-9.]*' $O/bench_prof.json | head -3
