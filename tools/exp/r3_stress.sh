#!/bin/bash
out=gpurun_out/r3w; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "range or radius" > $out/pytest_range.log 2> $out/pytest_range.err; echo "range tests rc=$?"; tail -2 $out/pytest_range.log
timeout -k 10 900 python tools/filter_stress.py --cases 400 --seed 777 > $out/filter_stress_seed777_400cases.txt 2> $out/stress.err; echo "stress rc=$?"; tail -1 $out/filter_stress_seed777_400cases.txt
timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4.json 2> $out/config4.err; cat $out/config4.json
