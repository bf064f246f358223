#!/bin/bash
out=gpurun_out/r3q; mkdir -p $out
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 6 --waves 5 --envs "MLVDB_SEED_ROWS=5;MLVDB_SEED_ROWS=3;MLVDB_SEED_ROWS=2;MLVDB_SEED_ROWS=1;MLVDB_SEED_ROWS=4" > $out/scan_ab_seed_rows.txt 2> $out/scan_ab.err; echo "scan_ab rc=$?"; cat $out/scan_ab_seed_rows.txt
timeout -k 10 420 python tools/scan_ab.py --rows 10000000 --rounds 6 --waves 5 --envs "MLVDB_REFINE_PICKS=16;MLVDB_REFINE_PICKS=12;MLVDB_REFINE_PICKS=20;MLVDB_REFINE_PICKS=24" > $out/scan_ab_picks.txt 2> $out/scan_ab2.err; cat $out/scan_ab_picks.txt
