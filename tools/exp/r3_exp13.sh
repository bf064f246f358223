#!/bin/bash
out=gpurun_out/r3n; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_plumbing_golden.py tests/test_gpu_protocol.py tests/test_pairwise.py -q -x -k "small_batches or fuzz or golden or replays or stream or narrow or distances" > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -3 $out/pytest.log
for lr in 1 0; do echo "== MLVDB_LAST_REFINE=$lr" >> $out/small.txt; MLVDB_LAST_REFINE=$lr timeout -k 10 120 python tools/small_batch_ab.py --rows 1000000 --batches 1,4,8 --modes exact,auto >> $out/small.txt 2>> $out/small.err; done
cat $out/small.txt
timeout -k 10 300 python bench.py --in-process --gpus 2 --devices 0,0 --rows-per-gpu 3000000 --steps 10 --warmup 3 > $out/bench_in_process.json 2> $out/bench_in_process.err; echo "in-process rc=$?"; python -c "
import json; d=json.loads(open('$out/bench_in_process.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('n_gpus','value','ms_per_step','ms_per_step_unpipelined_search_many','parity_gate')})"
