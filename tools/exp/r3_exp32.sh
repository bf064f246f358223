#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bench.py -q -x > $O/pytest.log 2> $O/pytest.err; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 900 python bench.py > $O/bench_default_flags.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4v/bench_default_flags.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','wave_mode','other_wave_mode','p50_ms_per_wave','p50_ms_per_wave_host_io','protocol_qps')})
print(d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['config2_1Mx768_batch1']['auto']['p50_ms'])
PY
