#!/bin/bash
out=gpurun_out/r3k; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_protocol.py -q -x -k "range or radius" > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -2 $out/pytest.log
timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4.json 2> $out/config4.err; cat $out/config4.json
