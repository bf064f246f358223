#!/bin/bash
set -o pipefail
out=gpurun_out/r3f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for sb in 0 1; do
  MLVDB_SMALL_BATCH=$sb MLVDB_SMALL_SEED=16 rocprofv3 --kernel-trace --stats -d $out/prof_sb$sb -o run -- python tools/small_batch_ab.py --rows 1000000 --batches 1 --modes auto --iters 60 > $out/run_sb$sb.txt 2> $out/run_sb$sb.err
  echo "sb=$sb rc=$?" | tee -a $out/log.txt
  f=$(find $out/prof_sb$sb -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_sb$sb.csv 2>/dev/null
done
head -25 $out/kernel_stats_sb0.csv; echo; head -25 $out/kernel_stats_sb1.csv
