#!/bin/bash
# round 3, experiment 1: scan variants A/B (release build), phase stamps (DIAG build), bare --gpus 2 rehearsal
set -o pipefail
out=gpurun_out/r3b; mkdir -p $out
python bench.py --gpus 2 --single-device --backend gloo --rows-per-gpu 2000000 --steps 5 --warmup 2 --no-cpu-baseline \
   > $out/bench_bare_gpus2.json 2> $out/bench_bare_gpus2.err; echo "bare bench rc=$?" | tee -a $out/log.txt
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 4 --waves 5 \
   --envs "MLVDB_SCAN_VAR=0;MLVDB_SCAN_VAR=231;MLVDB_SCAN_VAR=232;MLVDB_SCAN_VAR=233;MLVDB_SCAN_XCD=1;MLVDB_SCAN_XCD=1,MLVDB_SCAN_VAR=233" \
   > $out/scan_ab_variants.txt 2> $out/scan_ab_variants.err; echo "scan_ab rc=$?" | tee -a $out/log.txt
cat $out/scan_ab_variants.txt
make -C mlvectordb_amd/csrc clean > /dev/null && make -C mlvectordb_amd/csrc DIAG=1 -j16 > $out/make_diag.log 2>&1; echo "make DIAG rc=$?" | tee -a $out/log.txt
MLVDB_SCAN_DIAG=234 timeout -k 10 300 python tools/scan_ab.py --rows 10000000 --rounds 1 --waves 3 --envs "MLVDB_SCAN_DIAG=234" \
   > $out/phases.txt 2> $out/phases.err; echo "phases rc=$?" | tee -a $out/log.txt
grep "mlvdb" $out/phases.err | tail -6
