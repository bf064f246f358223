#!/bin/bash
# 10M x 768, batches of 1-8 queries (VERDICT r2 item 4 asked for <= 1.25 ms at batch 1)
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 500 python tools/small_batch_ab.py --rows 10000000 --batches 1,2,4,8 --modes exact,auto --iters 40 > $O/small_10m.txt 2>&1; echo "rc=$?"
cat $O/small_10m.txt
