#!/bin/bash
# fused last-refine + rescoring + ranking for batches of <= 8 queries (MLVDB_SMALL_FINISH=0: the three kernels)
set -e
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_protocol.py -q -x > $O/pytest.log 2> $O/pytest.err || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for v in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $v
  echo "== MLVDB_SMALL_FINISH=$1 MLVDB_SMALL_SEED=$2" >> $O/ab.txt
  MLVDB_SMALL_FINISH=$1 MLVDB_SMALL_SEED=$2 timeout -k 10 200 python tools/small_batch_ab.py --rows 1000000 --batches 1,2,4,8 --modes auto --iters 80 2>&1 | grep "nq" >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python3 tools/small_batch_ab.py --rows 1000000 --batches 1 --modes auto --iters 60 > $O/run.txt 2>&1
echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; fi
