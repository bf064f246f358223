#!/bin/bash
# The measurements a round commits under profiles/rNN/, in ONE gpurun call (from the repository root):
#   tools/measure_round.sh <tag> [bench|stats|pmc|sq ...]      default: bench stats pmc
#   bench  python bench.py (the driver's command)                        -> gpurun_out/<tag>/bench_n1.json
#   stats  rocprofv3 --kernel-trace --stats of the same command          -> gpurun_out/<tag>/kstats/run_results.db
#          (ROCm 7.2 writes a rocpd database: tools/rocpd_stats.py turns it into the per-kernel CSV that is committed)
#   pmc    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only (gpurun refuses other trace domains with
#          --pmc)                                                        -> pmc_fetch/, pmc_write/ -> tools/pmc_traffic.py
#   sq     five SQ / GRBM counter passes of the short bench              -> pass1..5/ -> tools/sq_counters.py
# rocprofv3 gets the program itself after `--` (python3 ...), never a shell or env wrapper.
set -o pipefail
tag=${1:-measure}; shift
what=${*:-bench stats pmc}
R=$PWD
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
short="--steps 4 --warmup 1 --no-cpu-baseline --no-extras --sustained-seconds 0"
for w in $what; do
  case $w in
    bench)
      timeout -k 10 600 python3 $R/bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?" | tee -a $out/log.txt ;;
    stats)
      timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/kstats -o run -- python3 $R/bench.py > $out/bench_n1_under_rocprof.json 2> $out/kstats.err
      echo "stats rc=$?" | tee -a $out/log.txt ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        d=$out/pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o run -- python3 $R/bench.py $short > $d.json 2> $d.err
        echo "pmc $c rc=$?" | tee -a $out/log.txt
      done
      python3 $R/tools/pmc_traffic.py $(ls $out/pmc_fetch/*counter_collection.csv | head -1) $(ls $out/pmc_write/*counter_collection.csv | head -1) > $out/pmc_traffic_i8.json
      tail -3 $out/pmc_traffic_i8.json ;;
    sq)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -o run -- python3 $R/bench.py $short > $out/pass$i.json 2> $out/pass$i.err
        echo "sq pass $i rc=$?" | tee -a $out/log.txt
      done
      python3 $R/tools/sq_counters.py $out > $out/scan_i8_sq_counters_10m.json; tail -12 $out/scan_i8_sq_counters_10m.json ;;
  esac
done
