#!/bin/bash
out=gpurun_out/r3o; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x -k "variants_agree and (default or 237 or 236)" > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -3 $out/pytest.log
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 8 --waves 5 --envs "MLVDB_SCAN_VAR=237;MLVDB_SCAN_VAR=0;MLVDB_SCAN_VAR=237;MLVDB_SCAN_VAR=0" > $out/scan_ab_default_vs_r2.txt 2> $out/scan_ab.err; echo "scan_ab rc=$?"; cat $out/scan_ab_default_vs_r2.txt
timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4.json 2> $out/config4.err; cat $out/config4.json
