#!/bin/bash
# final evidence of round 3 on the frozen build (each step its own log under gpurun_out/r3z)
out=gpurun_out/r3z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/gpu_suite.sh r3z || { echo "SUITE FAILED"; exit 1; }
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench_n1_final.json 2> $out/bench_n1_final.err; echo "bench rc=$?" | tee -a $out/log.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_kt -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_n1_final_under_rocprof.json 2> $out/prof_kt.err; echo "kt rc=$?" | tee -a $out/log.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_fetch.json 2> $out/pmc_fetch.err; echo "pmc fetch rc=$?" | tee -a $out/log.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_write.json 2> $out/pmc_write.err; echo "pmc write rc=$?" | tee -a $out/log.txt
find $out -name "*.csv" | head -20 | tee -a $out/log.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --tombstones 0.1 --no-extras --no-cpu-baseline > $out/bench_n1_tombstones10.json 2> $out/bench_tomb.err; echo "tomb rc=$?" | tee -a $out/log.txt
timeout -k 10 500 python tools/filter_stress.py --cases 150 --seed 3003 > $out/filter_stress_seed3003_150cases.txt 2> $out/stress.err; echo "stress rc=$?" | tee -a $out/log.txt; tail -1 $out/filter_stress_seed3003_150cases.txt
