#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
for pin in 0 1 2 3; do echo "== MLVDB_PINNED_IO=$pin" | tee -a $out/proto.txt; MLVDB_PINNED_IO=$pin timeout -k 10 200 python tools/exp/proto_ab.py 4000000 >> $out/proto.txt 2>> $out/proto.err; done
cat $out/proto.txt
