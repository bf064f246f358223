#!/bin/bash
# per-kernel times of the batch-1 path with the prefix-exact seed
O=gpurun_out/r4e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python3 tools/small_batch_ab.py --rows 1000000 --batches 1 --modes auto --iters 60 > $O/run.txt 2>&1
echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cut -d, -f1-6 "$f" | cut -c1-200 > $O/kernel_stats.csv; cat $O/kernel_stats.csv; else echo "no stats file"; find $O/prof | head; tail -5 $O/run.txt; fi
