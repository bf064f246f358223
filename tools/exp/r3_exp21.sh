#!/bin/bash
# prefix-exact seed for batches of <= 8 queries (MLVDB_SMALL_SEED=0: the dense int8 seeding pass + refine it replaces)
set -e
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "narrow or small or batch or fuzz or tomb" > $O/pytest.log 2> $O/pytest.err || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for v in "1 4" "0 4" "1 2" "0 4" "1 4"; do
  set -- $v
  echo "== MLVDB_SMALL_SEED=$1 MLVDB_PREFIX_WAVES=$2" >> $O/ab.txt
  MLVDB_SMALL_SEED=$1 MLVDB_PREFIX_WAVES=$2 timeout -k 10 200 python tools/small_batch_ab.py --rows 1000000 --batches 1,2,4,8 --modes auto --iters 80 2>&1 | grep "nq" >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python3 tools/small_batch_ab.py --rows 1000000 --batches 1 --modes auto --iters 60 > $O/run.txt 2>&1
echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; cut -d, -f1-4 $O/kernel_stats.csv | cut -c1-60,100-; fi
