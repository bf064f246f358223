#!/bin/bash
out=gpurun_out/r4b; mkdir -p $out
bash tools/gpu_suite.sh r4b || { echo "SUITE FAILED"; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.txt
timeout -k 10 600 python bench.py > $out/bench_default_flags.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$out/bench_default_flags.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','n_gpus','p50_ms_per_wave_host_io','protocol_qps')}); print(d['roofline']['frac'], d['roofline']['traffic']); print(d['config2_1Mx768_batch1']['auto']['p50_ms'], d['config4_10Mx768_l2_range']['knn_ms_per_wave_host_io'], d['config4_10Mx768_l2_range']['range_ms_per_wave_host_io'])"
