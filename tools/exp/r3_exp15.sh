#!/bin/bash
out=gpurun_out/r3p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_prof.json 2> $out/bench_prof.err; echo "rc=$?"
tail -c 400 $out/bench_prof.json
