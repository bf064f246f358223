#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o run -- python tools/config4.py --waves 6 > $out/config4_prof.json 2> $out/config4_prof.err; echo "rc=$?"
