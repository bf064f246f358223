"""The scan bodies (mlvectordb_amd/csrc/scan_asm_*.inc: build products of `make`, not tracked) are what
tools/gen_scan_asm.py emits into any directory: every listed file is written, the dispatch header covers every body, each
body is one asm statement; a built tree must hold exactly the generator's current output (CPU only: text generation)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "mlvectordb_amd" / "csrc"


def test_scan_bodies_are_what_the_generator_emits(tmp_path):
    gen = ROOT / "tools" / "gen_scan_asm.py"
    # --diag: the default library's bodies + the AB variants + the timing diagnostics (make / make AB=1 / make DIAG=1)
    names = subprocess.run([sys.executable, str(gen), "--list", "--diag"], check=True, capture_output=True, text=True).stdout.split()
    default = subprocess.run([sys.executable, str(gen), "--list"], check=True, capture_output=True, text=True).stdout.split()
    assert set(default) < set(names) and len(default) <= 24, "the default library carries only its own bodies"
    subprocess.run([sys.executable, str(gen), "--outdir", str(tmp_path), "--diag"], check=True, capture_output=True)
    assert names and sorted(p.name for p in tmp_path.iterdir()) == sorted(names)
    dispatch = (tmp_path / "scan_asm_dispatch.inc").read_text()
    for name in names:
        fresh = (tmp_path / name).read_text()
        if (CSRC / name).exists():  # a built tree: what the library was compiled from is the generator's current output
            assert (CSRC / name).read_text() == fresh, f"{name}: regenerate with `make -C mlvectordb_amd/csrc`"
        if name.startswith("scan_asm_") and name not in ("scan_asm_dispatch.inc", "scan_asm_consts.inc"):
            assert f'#include "{name}"' in dispatch, f"{name} is generated but never dispatched"
            assert fresh.count("asm volatile(") == 1  # one statement: nothing in flight crosses compiler-managed code
    body = (tmp_path / "scan_asm_cosine_i8.inc").read_text()
    assert "v_mfma_i32_16x16x64_i8" in body and "v_mfma_f32_16x16x32_bf16" not in body


def test_bench_hash_of_the_default_bodies_needs_no_built_tree():
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen", ROOT / "tools" / "gen_scan_asm.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    body = gen.default_i8_body("cosine")
    assert body.count("asm volatile(") == 1 and "v_mfma_i32_16x16x64_i8" in body
    if (CSRC / "scan_asm_cosine_i8_va.inc").exists():
        assert (CSRC / "scan_asm_cosine_i8_va.inc").read_text() == body


def test_query_tile_and_l2_bodies_have_the_structure_they_claim():
    """Round 4 bodies, by their text: the 4- / 8-tile bodies issue a quarter / half of the 16-tile body's MFMAs (nothing for the
    empty query tiles); the l2 body takes its first k-step's C operand from the offset registers, prefetches them a tile ahead
    through the row pairs' descriptor and keeps its thresholds in registers (one compare per query tile, like cosine's)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen2", ROOT / "tools" / "gen_scan_asm.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    mf = {}
    for nqt in (16, 8, 4):
        gen.DBG.clear()
        body = gen.generate("cosine", 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True, nqt=nqt)
        mf[nqt] = body.count("v_mfma_i32_16x16x64_i8 v[")
        assert body.count("asm volatile(") == 1 and f".Lhit{nqt - 1}c200" in body and f".Lhit{nqt}c200" not in body
    assert mf[16] == 4 * 128 and mf[8] * 2 == mf[16] and mf[4] * 4 == mf[16]  # 4 copies of the 4-k-step body x 2 * NQT * 2 x 2
    l2 = gen.default_i8_body("l2")
    assert "v_mfma_i32_16x16x64_i8 v[64:67], %[x0], %[t0], %[eo0]" in l2      # first k-step: C = the lane's offsets
    assert l2.count("buffer_load_dwordx4 %[eo0]") == 3                          # prologue + the two last-body copies (.Llast, .Lsingle)
    assert "%[tq15]" in l2 and "%[kec]" in l2 and "%[krc]" in l2 and "%[sqc]" in l2 and "%[k1]" not in l2
    assert l2.count("ds_read_b32 %[tq") == 16                                  # thresholds: once per launch, not per row tile
