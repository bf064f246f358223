#!/bin/bash
# 10M x 768: int8 narrow kernel for batches of 9-64 queries (MLVDB_NARROW_I8_MAX=64) vs the padded 256-query body (default 8)
O=gpurun_out/r4s; mkdir -p $O
for mx in 8 64; do
  echo "== MLVDB_NARROW_I8_MAX=$mx" >> $O/narrow.txt
  MLVDB_NARROW_I8_MAX=$mx timeout -k 10 500 python tools/small_batch_ab.py --rows 10000000 --batches 9,16,32,64 --modes exact,auto --iters 20 2>&1 | grep "nq" >> $O/narrow.txt
done
cat $O/narrow.txt
