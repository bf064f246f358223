#!/bin/bash
# exactness campaign on the final build, default mix of batches and dimensions
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 1000 python tools/filter_stress.py --cases 250 --seed 909 > $O/stress.txt 2>&1; echo "rc=$?"
tail -2 $O/stress.txt
