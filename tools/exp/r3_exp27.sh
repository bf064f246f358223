#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide_rows" > $O/pytest.log 2> $O/pytest.err; echo rc=$?
tail -25 $O/pytest.log
