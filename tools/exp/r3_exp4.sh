#!/bin/bash
set -o pipefail
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_protocol.py -q -x > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -3 $out/pytest.log
for cfg in "0 16" "1 8" "1 16" "1 32" "1 64"; do
  set -- $cfg
  echo "== MLVDB_SMALL_BATCH=$1 MLVDB_SMALL_SEED=$2" >> $out/small_batch_1m.txt
  MLVDB_SMALL_BATCH=$1 MLVDB_SMALL_SEED=$2 timeout -k 10 120 python tools/small_batch_ab.py --rows 1000000 --batches 1,4,8 --modes exact,auto >> $out/small_batch_1m.txt 2>> $out/small_batch_1m.err
done
cat $out/small_batch_1m.txt
