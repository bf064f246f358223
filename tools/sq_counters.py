#!/usr/bin/env python3
"""rocprofv3 --pmc passes (csv) of one bench command -> SQ / GRBM counters of the LAST wave's scan launches, with the derived
figures DESIGN quotes: effective shader clock (GRBM_GUI_ACTIVE / 8 / duration), matrix-pipe busy share, cycles per MFMA and SIMD,
parked / issue-stalled shares of the wave cycles, LDS bank conflicts.  usage: sq_counters.py <dir with pass*/ subdirs> [launches per wave]"""
import csv, glob, json, sys
from collections import defaultdict

root = sys.argv[1]
per_wave = int(sys.argv[2]) if len(sys.argv) > 2 else 3
counters, durations = defaultdict(dict), {}
for path in sorted(glob.glob(f"{root}/pass*/*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "filter_scan_asm_kernel" in row["Kernel_Name"]:
                counters[row["Counter_Name"]][int(row["Dispatch_Id"])] = float(row["Counter_Value"])
for path in sorted(glob.glob(f"{root}/pass*/*kernel_trace.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "filter_scan_asm_kernel" in row["Kernel_Name"]:
                durations.setdefault(path, {})[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
out = {"source": "rocprofv3 --kernel-trace --pmc <4 counters> (one pass per set), python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras; "
                 "the scan launches of the last wave (tools/sq_counters.py)", "counters": {}}
for name, by_id in counters.items():
    ids = sorted(by_id)[-per_wave:]
    out["counters"][name] = [by_id[i] for i in ids]
dur = next(iter(durations.values()))
ids = sorted(dur)[-per_wave:]
out["launch_duration_us_in_a_profiled_pass"] = [dur[i] / 1e3 for i in ids]
c = out["counters"]
big = per_wave - 1  # the largest launch
d = {}
if "GRBM_GUI_ACTIVE" in c:
    d["effective_clock_GHz_largest_launch"] = round(c["GRBM_GUI_ACTIVE"][big] / 8 / (dur[ids[big]] * 1e-9) / 1e9, 3)
if "SQ_INSTS_MFMA" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    cycles = c["GRBM_GUI_ACTIVE"][big] / 8
    d["mfma_pipe_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"][big] / (cycles * 1024), 3)
    d["cycles_per_mfma_and_simd"] = round(cycles * 1024 / c["SQ_INSTS_MFMA"][big], 2)
if "SQ_WAVE_CYCLES" in c:
    for k, n in (("SQ_WAIT_ANY", "parked_share_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "issue_stalled_share_of_wave_cycles")):
        if k in c:
            d[n] = round(c[k][big] / c["SQ_WAVE_CYCLES"][big], 3)
if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
    d["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"][big] / max(1.0, c["SQ_LDS_IDX_ACTIVE"][big]), 4)
out["derived_largest_launch"] = d
print(json.dumps(out, indent=1))
