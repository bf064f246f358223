#!/bin/bash
set -o pipefail
out=gpurun_out/r3c; mkdir -p $out
MLVDB_HIP_LIBRARY=$PWD/mlvectordb_amd/csrc/libmlvdb_diag.so MLVDB_SCAN_DIAG=234 timeout -k 10 300 python tools/scan_ab.py --rows 10000000 --rounds 1 --waves 3 --envs "MLVDB_SCAN_DIAG=234" \
   > $out/phases.txt 2> $out/phases.err; echo "phases rc=$?" | tee -a $out/log.txt
grep "mlvdb" $out/phases.err | tail -9
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 8 --waves 5 \
   --envs "MLVDB_SCAN_VAR=0;MLVDB_SCAN_VAR=232;MLVDB_SCAN_VAR=0;MLVDB_SCAN_VAR=232" \
   > $out/scan_ab_eo.txt 2> $out/scan_ab_eo.err; echo "scan_ab rc=$?" | tee -a $out/log.txt
cat $out/scan_ab_eo.txt
