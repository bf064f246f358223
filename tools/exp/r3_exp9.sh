#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/gpu_suite.sh r3j || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o run -- python tools/config4.py --waves 10 > $out/config4_prof.json 2> $out/config4_prof.err; echo "config4 prof rc=$?" | tee -a $out/log.txt
timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4.json 2> $out/config4.err; cat $out/config4.json
timeout -k 10 200 python tools/exp/proto_ab.py 4000000 > $out/proto.txt 2> $out/proto.err; cat $out/proto.txt
