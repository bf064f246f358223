#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -q -x --durations=5 > $O/pytest.log 2> $O/pytest.err; echo rc=$?
tail -12 $O/pytest.log
