#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs of the same bench command, csv output) ->
HBM bytes per launch of the filter scan kernel, with the gfx950 FETCH_SIZE x2 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Takes the three largest-grid launches' pattern of one wave:
the counters of the LAST complete wave (three consecutive launches of filter_scan_asm_kernel) are reported."""
import csv, hashlib, json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


sys.path.insert(0, str(ROOT))
from bench import kernel_source_sha16  # noqa: E402  (one definition: ties the measurement to the kernel sources it was taken on)


def launches(path, counter):
    out = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if "filter_scan_asm_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                out.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), row["Kernel_Name"]))
    out.sort()
    return out

fetch = launches(sys.argv[1], "FETCH_SIZE")
write = launches(sys.argv[2], "WRITE_SIZE")
per_wave = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f3, w3 = [v for _, v, _ in fetch[-per_wave:]], [v for _, v, _ in write[-per_wave:]]
fetch_bytes = 2.0 * 1024.0 * sum(f3)   # counter unit KB; x2: gfx950 reports half of a wide coalesced stream
write_bytes = 1024.0 * sum(w3)
print(json.dumps({
    "kernel_source_sha16": kernel_source_sha16(),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras, MI355X (tools/pmc_traffic.py)",
    "kernel": fetch[-1][2] + ": the launches of one 256-query wave over 10M x 768",
    "FETCH_SIZE_KB_per_launch": f3, "WRITE_SIZE_KB_per_launch": w3,
    "fetch_correction": "x2: on gfx950 FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read",
    "fetch_bytes_per_wave_corrected": fetch_bytes, "write_bytes_per_wave": write_bytes,
    "traffic_bytes_per_launch_avg": (fetch_bytes + write_bytes) / per_wave,
}, indent=1))
