#!/bin/bash
set -o pipefail
out=gpurun_out/r3d; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "variants_agree and (VAR or XCD or default)" > $out/pytest_variants.log 2> $out/pytest_variants.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -3 $out/pytest_variants.log
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 8 --waves 5 \
   --envs "MLVDB_SCAN_VAR=0;MLVDB_SCAN_VAR=235;MLVDB_SCAN_VAR=236;MLVDB_SCAN_VAR=232" \
   > $out/scan_ab_fs.txt 2> $out/scan_ab_fs.err; echo "scan_ab rc=$?" | tee -a $out/log.txt
cat $out/scan_ab_fs.txt
