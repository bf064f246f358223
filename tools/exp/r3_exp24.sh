#!/bin/bash
# small batches: fused finish (MLVDB_SMALL_NQ = largest batch that takes it) x prefix-exact seed, interleaved repeats
set -e
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "narrow or small or batch or fuzz or tomb or golden" > $O/pytest.log 2> $O/pytest.err || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2 3; do
for v in "2 1" "0 0" "2 0" "0 1" "8 1"; do
  set -- $v
  echo "== MLVDB_SMALL_NQ=$1 MLVDB_SMALL_SEED=$2" >> $O/ab.txt
  MLVDB_SMALL_NQ=$1 MLVDB_SMALL_SEED=$2 timeout -k 10 200 python tools/small_batch_ab.py --rows 1000000 --batches 1,2,4,8 --modes auto --iters 150 2>&1 | grep "nq" | tr '\n' ' ' >> $O/ab.txt
  echo >> $O/ab.txt
done
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python3 tools/small_batch_ab.py --rows 1000000 --batches 1 --modes auto --iters 60 > $O/run.txt 2>&1
echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; fi
