#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_protocol.py -q -x -k "row_mask or filtered or mask" > $O/pytest.log 2> $O/pytest.err; echo rc=$?
tail -25 $O/pytest.log
