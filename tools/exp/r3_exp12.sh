#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out
bash tools/gpu_suite.sh r3m || exit 1
timeout -k 10 300 python tools/filter_stress.py --cases 60 --seed 31 > $out/stress.txt 2> $out/stress.err; echo "stress rc=$?"; tail -2 $out/stress.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/log.txt
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','p50_ms_per_wave','p50_ms_per_wave_host_io','protocol_qps','candidates_rescored_per_query')}); print(d['roofline']['frac'], d['roofline']['avg_launch_ms']); print(d.get('config2_1Mx768_batch1')['auto']); print(d.get('config4_10Mx768_l2_range'))"
