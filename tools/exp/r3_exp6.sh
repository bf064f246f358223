#!/bin/bash
set -o pipefail
out=gpurun_out/r3g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or exact_scan or overflow or fallback or range" > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -2 $out/pytest.log
timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4.json 2> $out/config4.err; echo "config4 rc=$?" | tee -a $out/log.txt; cat $out/config4.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o run -- python tools/config4.py --waves 10 > $out/config4_prof.json 2> $out/config4_prof.err; echo "config4 prof rc=$?" | tee -a $out/log.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/log.txt
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','p50_ms_per_wave','p50_ms_per_wave_host_io','protocol_qps')}); print(d['roofline']['frac'], d['roofline']['avg_launch_ms']); print(d.get('config2_1Mx768_batch1')); print(d.get('config4_10Mx768_l2_range'))"
