#!/bin/bash
out=gpurun_out/r3t; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -q -x > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','p50_ms_per_wave','p50_ms_per_wave_host_io','protocol_qps')}); print(d['roofline']['frac'], d['roofline']['traffic']); print(d['config2_1Mx768_batch1'])"
