#!/bin/bash
# how the caller waits: hipStreamSynchronize vs polling hipStreamQuery (batch 1-2 on 1M x 768)
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2 3; do
  for sp in "" "--spin"; do
    echo "== ${sp:-sync}" >> $O/spin.txt
    timeout -k 10 200 python tools/small_batch_ab.py --rows 1000000 --batches 1,2 --modes auto --iters 150 $sp 2>&1 | grep "nq" | tr '\n' ' ' >> $O/spin.txt; echo >> $O/spin.txt
  done
done
cat $O/spin.txt
