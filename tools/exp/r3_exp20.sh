#!/bin/bash
out=gpurun_out/r3v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/pass$i.json 2> $out/pass$i.err; echo "pass $i ($set) rc=$?" | tee -a $out/log.txt
done
python tools/sq_counters.py $out > $out/scan_i8_sq_counters_10m.json; tail -20 $out/scan_i8_sq_counters_10m.json
