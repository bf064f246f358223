#!/bin/bash
set -o pipefail
out=gpurun_out/r3h; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "range" > $out/pytest.log 2> $out/pytest.err; echo "pytest rc=$?" | tee -a $out/log.txt; tail -2 $out/pytest.log
for flat in 0 1; do
MLVDB_RANGE_FLAT=$flat timeout -k 10 300 python tools/config4.py --waves 10 > $out/config4_flat$flat.json 2> $out/config4.err; echo "config4 flat=$flat rc=$?" | tee -a $out/log.txt; python -c "
import json; d=json.loads(open('$out/config4_flat$flat.json').read()); print({k: d[k] for k in ('knn_ms_per_wave_host_inclusive','range_ms_per_wave_host_inclusive','parity')})"
done
for pin in 0 1; do echo "== MLVDB_PINNED_IO=$pin" | tee -a $out/proto.txt; MLVDB_PINNED_IO=$pin timeout -k 10 200 python tools/exp/proto_ab.py 4000000 >> $out/proto.txt 2>> $out/proto.err; done
cat $out/proto.txt
