#!/bin/bash
# GPU-side span of a batch-1 call (first kernel's start -> last kernel's end, rocprofv3 kernel trace) by configuration, one box
O=gpurun_out/r4j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "0 0" "0 1" "2 0" "2 1" "0 0" "2 0"; do
  set -- $v
  tag=nq$1_seed$2_$RANDOM
  MLVDB_SMALL_NQ=$1 MLVDB_SMALL_SEED=$2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 tools/small_batch_ab.py --rows 1000000 --batches 1,2 --modes auto --iters 60 > $O/$tag.txt 2>&1
  echo "$tag rc=$?"; grep "nq" $O/$tag.txt | tr '\n' ' '; echo
done
