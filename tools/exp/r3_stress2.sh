#!/bin/bash
# exactness campaign on the small-batch steps (1-2 queries, int8 shadow): filter == exact scan, ids and fp32 distances
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python tools/filter_stress.py --cases 300 --seed 5151 --batches 1,2 --dims 256,512,768,1024 --max-rows 150000 > $O/stress_small.txt 2>&1; echo "rc=$?"
tail -3 $O/stress_small.txt; grep -c "BAD" $O/stress_small.txt
