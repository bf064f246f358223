#!/usr/bin/env python3
"""Build-time check of the register trick the default int8 scan bodies rest on.

The ArchVGPR-accumulator bodies (tools/gen_scan_asm.py, flag `va`) name v64..v191 explicitly and list them as clobbers;
hipcc warns that v128.. are "reserved registers" there.  It works because the kernel descriptor hipcc emits for
these kernels asks for 240 registers per lane (192 ArchVGPRs + 48 AccVGPRs, two waves per SIMD) with nothing spilled.  A
compiler that decided otherwise would still build -- and run wrong, or slowly through scratch.  This script reads the
descriptors back from the object file and fails the build unless every default body says exactly that:

    vgpr_count == 240, vgpr_spill_count == 0, private_segment_fixed_size == 0 (no scratch)

Usage: check_scan_regs.py <kernels_filter.o>   (called by `make` after the object is built, and by __graft_entry__.build()).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
# filter_scan_asm_kernel<SPACE, R, NW, NT, QD, ...>: the QD slot carries the body code; these are the ArchVGPR bodies the
# default library dispatches to (kernels_filter.hip, launch_scan_space)
VA_CODES = {211, 237} | set(range(214, 223)) | {228, 229, 231, 232, 233, 235, 236} | set(range(241, 250))
WANT = {".vgpr_count": 240, ".vgpr_spill_count": 0, ".private_segment_fixed_size": 0}
# AB variants that use all 64 AccVGPRs (ring of 6 k-steps, or B fragments read 8 ahead): 192 + 64 registers
WIDE_CODES = {214, 215, 228, 229}


def kernel_records(obj: Path):
    """(kernel name, {metadata field: int}) of every gfx950 kernel in the object's offload bundle (decompressed if need be)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = Path(tmp) / "fat.bin", Path(tmp) / "k.co"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True)
        if not fat.exists() or fat.stat().st_size == 0:
            raise SystemExit(f"{obj}: no .hip_fatbin section")
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        # amdhsa.kernels is a YAML list; one record per "  - .key:" item
        for rec in re.split(r"\n\s+- (?=\.[a-z_]+:)", notes):
            m = re.search(r"\.name:\s+(\S+)", rec)
            if m:
                fields = {k: int(v) for k, v in re.findall(r"(\.[a-z_]+):\s+(\d+)\s*$", rec, flags=re.M)}
                yield m.group(1), fields


def main(argv):
    if len(argv) != 2:
        raise SystemExit(__doc__)
    obj = Path(argv[1])
    checked, bad = 0, []
    for name, f in kernel_records(obj):
        m = re.match(r"_ZN5mlvdb22filter_scan_asm_kernelILi(\d)ELi(\d)ELi(\d)ELb[01]ELi(\d+)E", name)
        if not m or int(m.group(4)) not in VA_CODES:
            continue
        checked += 1
        for key, want in WANT.items():
            if key == ".vgpr_count" and int(m.group(4)) in WIDE_CODES:
                want = 256
            if f.get(key) != want:
                bad.append(f"{name}: {key} = {f.get(key)} (want {want})")
    if checked < 3:
        bad.append(f"only {checked} ArchVGPR scan bodies found in {obj} (want one per space at least)")
    if bad:
        print("check_scan_regs: the default scan bodies do not have the register budget they were written for:", file=sys.stderr)
        for b in bad:
            print("  " + b, file=sys.stderr)
        return 1
    print(f"check_scan_regs: {checked} ArchVGPR scan bodies: 240 VGPRs, no spills, no scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
