#!/usr/bin/env python3
"""Generates the gfx950 assembly of the filter scan (bf16 MFMA bounds of all rows x 256 queries).

Output: mlvectordb_amd/csrc/scan_asm_{space}_nw{NW}_r{R}.inc -- ONE `asm volatile(...)` statement,
the whole body of filter_scan_asm_kernel (kernels_filter.hip): prologue, the persistent loop over
the workgroup's row tiles, the k-loop of a tile, the admission test and the (rare) append path.

Why assembly: hipcc's schedule of the same loop drains the X prefetch every two k-steps
(vmcnt(0) + register copies at the back edge), reads each B fragment right before its MFMAs, and
spills around the epilogue; values that are the targets of loads still in flight cannot be handed
through compiler-managed code at all.  Here everything that is in flight stays inside one
statement, and the waits are computed by simulating the two in-order queues (vmcnt: buffer/global
operations; lgkmcnt: LDS operations).

Per wave: MT row panels (16 rows each) x 16 query tiles = 16*MT accumulators of 16x16 in AGPRs.
MT = 2 (32 rows, a[0:127]): two waves per SIMD -- hipcc splits the 256 registers 128 VGPR / 128 AGPR
as soon as a kernel touches an AGPR, so the accumulators take the AGPRs and everything else lives in
<= 128 VGPRs.  MT = 4 (64 rows, a[0:255]): one wave per SIMD, 256 + 256 registers; every B fragment
read from LDS feeds 4 MFMAs instead of 2 and the query image is staged once per 4 waves -- less
energy per MFMA, which is what counts on a power-bound pipe (tools/probe).
  * X (bf16 shadow panels, HBM): buffer_load_dwordx4 into a ring of R k-steps that are the MFMA A
    operands; a slot is refilled right after its last MFMA with the k-step R ahead -- the last R
    k-steps of a tile fetch the first R of the workgroup's next tile through a second descriptor,
    so the stream never stops, not even during the admission test.
  * Q (bf16 image, L2): 64-column chunks, double buffered in LDS, one s_barrier per chunk.  Default
    (dma): early in chunk c every wave sends its share of chunk c+1 global -> LDS directly
    (buffer_load ... lds; the LDS address is M0 + 16*lane) and waits for it before the barrier.
    Variant: through registers -- during chunk c every thread writes its share of chunk c+1
    (fetched one chunk earlier into two register sets) and fetches its share of chunk c+2.
    vmcnt completes in order, so waiting for a Q transfer also waits for every X refill issued
    before it: the transfers sit BEFORE the refills of their k-step, which leaves each refill two
    to three k-steps.
  * the 32 B fragments of a chunk are one software-pipelined stream: ds_read_b128 runs QD
    fragments ahead of the two MFMAs that consume a fragment.
  * admission test per query tile: 8 bounds per lane (same arithmetic as scan_epilogue), their
    maximum against the threshold; only if some lane passes, an out-of-line routine appends
    (bound, row, query) entries to the WAVE's private buffer, staged in LDS until the kernel ends:
    the fill count lives in an SGPR, slots come from v_mbcnt -- no atomic, no wait, and no global
    store in the loop (see gen_slow).  The kernel's C++ tail then moves the entries into the
    per-query lists.  The routine's memory
    operations are younger than every prefetch, so the counted waits elsewhere stay sufficient.
"""
import argparse
from pathlib import Path

MT = 2
CHUNK_BYTES = 0x8000      # 256 queries x 64 columns x 2 B
WG_CAP = 16384            # kWgCap: append entries per workgroup (split evenly over its waves)
SPACES = {"l2": 0, "cosine": 1, "ip": 2}
I8_SPACE = None
I8 = False     # generate(): int8 shadow -- v_mfma_i32_16x16x64_i8, k-steps of 64 columns, integer accumulators
STAG = False   # generate(): the later-dispatched half of the waves runs half a tile behind (see generate)
VA = False     # generate(): accumulators in ArchVGPRs v[VA_BASE : VA_BASE + 64*MT), ring and B fragments in AccVGPRs (see generate)
VA_BASE = 64   # v0..v63 stay with the compiler (the statement's "v" operands)
BURST = 0      # generate(): ring refills issued in bursts of BURST consecutive k-steps of a panel (contiguous KiBs) instead of one per k-step
Q3D = False    # generate(): four Q buffers, chunk c+3 staged during chunk c, its landing awaited TWO chunks later (see generate)
Q4 = False     # generate(): four Q chunk buffers in LDS, chunk c+2 staged during chunk c, ONE barrier per two chunks
FUSE = False   # generate(): the admission test is folded into the tile's last k-step (cosine, VA; see gen_pretest)
QA = False     # generate(): four Q buffers, chunk c+2 staged during chunk c, one barrier per chunk, and the B-fragment read
#                stream runs ACROSS the chunk (and tile) boundary: the first QD fragments of chunk c+1 are read during the
#                last QD fragments of chunk c, before the barrier (see generate)
EO = False     # generate(): hit stubs leave at once when none of the 8 exact bounds passes (see gen_hit_stubs)
FS = False     # generate(): append routine with a straight-line common case (see gen_slow_fast)
L2E = False    # generate(): l2 with the admission test folded into the last k-step, made sharp by per-row integer offsets that
#                enter the accumulators through the first k-step's C operand (see generate)
L2C = False    # generate(): L2E with ONE query scale SQ and ONE error coefficient KE for the whole pass (the prep builds the images
#                that way): the pre-test is cosine's -- one fma against a threshold held in a register (see generate)
NQT = 16       # generate(): query tiles (of 16 queries) the body computes: 16 = a full 256-query pass; 8 / 4 (int8 VA bodies, round 4) for
#                passes of <= 128 / <= 64 queries -- the MFMAs, B-fragment reads, Q staging and admission tests of the empty tiles are
#                not issued at all, which leaves a pure stream of the shadow (see generate)
DBG = set()   # timing diagnostics only (wrong results): 'nolds' drops the B-fragment reads, 'nox' the X refills


class Sched:
    """Instruction list + in-order queue simulation for counted waits."""

    def __init__(self):
        self.lines = []
        self.vm = []
        self.lg = []
        self.vm_done = 0
        self.lg_done = 0
        self.recording = True
        self.label = 0
        self.copy = ""   # which copy of the tile's last body this is (FUSE: its hit stubs return into it)

    def emit(self, text):
        if "nobar" in DBG and text == "s_barrier":
            return
        if self.recording:
            self.lines.append(text)

    def vmem(self, text, tag):
        if not ("nox" in DBG and tag[0] == "x") and not ("noq" in DBG and tag[0] in ("qa", "qb")):
            self.emit(text)
        self.vm.append(tag)

    def lds(self, text, tag):
        if not ("nolds" in DBG and tag[0] == "rd" and tag[-1] >= 4) and not ("noq" in DBG and tag[0] == "wr"):
            self.emit(text)
        self.lg.append(tag)

    def _last(self, q, tag):
        for i in range(len(q) - 1, -1, -1):
            if q[i] == tag:
                return i
        if not self.recording:
            return -1      # warm-up pass: nothing older exists yet
        raise KeyError(tag)

    def need_vm(self, *tags):
        idx = max(self._last(self.vm, t) for t in tags)
        if idx < self.vm_done:
            return
        n = len(self.vm) - 1 - idx
        assert n <= 63, n
        self.emit(f"s_waitcnt vmcnt({n})")
        self.vm_done = idx + 1

    def need_lg(self, *tags):
        idx = max(self._last(self.lg, t) for t in tags)
        if idx < self.lg_done:
            return
        n = min(len(self.lg) - 1 - idx, 15)
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.lg_done = max(self.lg_done, len(self.lg) - n)

    def drain_lg(self):
        if self.lg_done < len(self.lg):
            self.emit("s_waitcnt lgkmcnt(0)")
            self.lg_done = len(self.lg)


def acc(m, n):
    b = (m * 16 + n) * 4
    if VA:
        return f"v[{VA_BASE + b}:{VA_BASE + b + 3}]"
    return f"a[{b}:{b + 3}]"


def acc_reg(m, n, i):
    """Register i (0..3: rows 4g+i of panel m) of the accumulator of (panel m, query tile n)."""
    return f"v{VA_BASE + (m * 16 + n) * 4 + i}" if VA else f"a{(m * 16 + n) * 4 + i}"


def ring(b, m):
    return f"%[x{b * MT + m}]"


# explicit scalar registers (listed as clobbers): descriptors need sub-register arithmetic
PRIO_STEPS = {8: (0, 2), 12: (1, 2), 16: (0, 1), 20: (1, 1), 24: (0, 0), 28: (1, 0)}   # fragment -> (wave half, priority)
XCUR, XNEXT, RNS, RET = "s[80:83]", "s[84:87]", "s[88:91]", "s[92:93]"


def gen_pretest(s, n, part):
    """Admission pre-test of query tile n, folded into the tile's last k-step (int8 cosine, ArchVGPR accumulators).

    Exact test per row j and query: float(I_j) r_j + p_j >= T (gen_admission).  With r_j >= 0 and
    R = max_j r_j, P = max_j p_j over the lane's 8 rows (NaN = tombstoned rows drop out of v_max_f32),
    float(max(0, max_j I_j)) R + P >= float(I_j) r_j + p_j for every j (rounding is monotone), so a lane whose
    left-hand side stays below T holds no admissible row: 4 v_max3_i32 + cvt + fma + compare per query tile instead
    of 8 reads + 8 cvt + 8 fma + 5 max + compare, issued between the MFMAs of the following query tiles.  A lane
    that passes sends the wave to .Lhit<n>, which computes the 8 exact bounds and calls the append routine; on
    N(0,1) rows the pre-test lets ~2 % of the (wave, query tile) pairs through, the exact test 0.3 %.
    part 0 / 1: the halves issued after the first / second MFMA of the query tile two steps later."""
    a = s.emit
    regs = [acc_reg(m, n, i) for m in range(MT) for i in range(4)]
    if I8_SPACE != "cosine" and not L2C:
        return gen_pretest_l2ip(s, n, part, regs)
    t0, t1 = (("%[e0]", "%[e1]") if n & 1 == 0 else ("%[e2]", "%[e3]"))
    if part == 0:
        a(f"v_max3_i32 {t0}, {regs[0]}, {regs[1]}, {regs[2]}")
        a(f"v_max3_i32 {t1}, {regs[3]}, {regs[4]}, {regs[5]}")
        a(f"v_max3_i32 {t0}, {t0}, {regs[6]}, {regs[7]}")
        if L2C:   # (A_j = I_j + e_j times ONE positive factor S SQ: no clamp at 0 needed, and none wanted)
            a(f"v_max_i32 {t0}, {t0}, {t1}")
        else:
            a(f"v_max3_i32 {t0}, {t0}, {t1}, 0")
    else:
        a(f"v_cvt_f32_i32 {t0}, {t0}")
        a(f"v_fma_f32 {t0}, {t0}, %[e10], %[e12]")
        a(f"v_cmp_ge_f32 vcc, {t0}, %[tq{n}]")
        if "nohit" not in DBG:
            a(f"s_cbranch_vccnz .Lhit{n}c{s.copy}_%=")
        a(f".Lback{n}c{s.copy}_%=:")


# l2 / ip folded pre-test: the e registers that only the serial admission test used (the append routine keeps e5..e9, e11)
# plus four of the tq pool (the thresholds come from LDS per query tile here)
L2IP_SMAX, L2IP_NMAX, L2IP_PMAX, L2IP_T0, L2IP_T1 = "%[e10]", "%[e12]", "%[e4]", "%[e0]", "%[e1]"
L2IP_TQ = 4


def l2ip_consts(n):
    """(thr, ke, sq) registers of query tile n's per-query constants: two sets, by parity."""
    return ("%[e2]", "%[e3]", "%[tq0]") if n & 1 == 0 else ("%[tq1]", "%[tq2]", "%[tq3]")


def l2ip_fetch(s, n):
    thr, ke, sq = l2ip_consts(n)
    s.lds(f"ds_read_b32 {thr}, %[thra]" + (f" offset:{n * 64}" if n else ""), ("thr", n))
    s.lds(f"ds_read_b32 {ke}, %[thra] offset:{2048 + n * 64}", ("ke", n))
    if I8_SPACE == "l2":
        s.lds(f"ds_read_b32 {sq}, %[thra] offset:{1024 + n * 64}", ("sq", n))


def gen_pretest_l2ip(s, n, part, regs):
    """Folded admission pre-test of query tile n, l2 / ip (int8, ArchVGPR accumulators).

    Exact test per row j (gen_admission / the hit stub, in this order of operations): u = float(I_j); u = u s_j;
    u = fma(ke, N_j, u); l2: u = fma(sq, u, p_j); u >= thr, with s_j the row's scale, N_j = |x_j|, p_j = -(1 - slack) N_j^2,
    and ke, sq, thr >= 0-scaled constants of the lane's query.  The pre-test runs the SAME operations on a virtual row that
    dominates the lane's 8 rows -- I* = max(0, max_j I_j), S = max_j s_j, N = max_j N_j, P = max_j p_j: every operation is
    monotone in each of its inputs (ke, sq >= 0; rounding is monotone), so its result is >= every row's u, and a lane whose
    virtual row stays below thr holds no admissible row.  It is as sharp as the rows of a lane group are alike: the shadow
    builder gives them one scale (shadow8_rows_kernel), the norms are what they are (Gaussian rows: ~12 % of the (wave,
    query tile) pairs reach the stub in the last round, where the exact test passes 0.7 %)."""
    a = s.emit
    thr, ke, sq = l2ip_consts(n)
    if part == 0:
        a(f"v_max3_i32 {L2IP_T0}, {regs[0]}, {regs[1]}, {regs[2]}")
        a(f"v_max3_i32 {L2IP_T1}, {regs[3]}, {regs[4]}, {regs[5]}")
        a(f"v_max3_i32 {L2IP_T0}, {L2IP_T0}, {regs[6]}, {regs[7]}")
        if L2E:   # A_j = I_j + e_j, one scale S for the lane's rows: max_j (A_j S) = (max_j A_j) S whatever the sign
            a(f"v_max_i32 {L2IP_T0}, {L2IP_T0}, {L2IP_T1}")
        else:
            a(f"v_max3_i32 {L2IP_T0}, {L2IP_T0}, {L2IP_T1}, 0")
        if n + 1 < NQT:
            l2ip_fetch(s, n + 1)
    else:
        a(f"v_cvt_f32_i32 {L2IP_T0}, {L2IP_T0}")
        a(f"v_mul_f32 {L2IP_T0}, {L2IP_T0}, {L2IP_SMAX}")
        s.need_lg(("ke", n), ("thr", n), *([("sq", n)] if I8_SPACE == "l2" else []))
        a(f"v_fma_f32 {L2IP_T0}, {ke}, {L2IP_NMAX}, {L2IP_T0}")
        if I8_SPACE == "l2":
            a(f"v_fma_f32 {L2IP_T0}, {sq}, {L2IP_T0}, {L2IP_PMAX}")
        a(f"v_cmp_ge_f32 vcc, {L2IP_T0}, {thr}")
        if "nohit" not in DBG:
            a(f"s_cbranch_vccnz .Lhit{n}c{s.copy}_%=")
        a(f".Lback{n}c{s.copy}_%=:")


def gen_rowmax_l2ip(s, part):
    """Start of the last k-step, l2 / ip: p_j = k1 N_j^2 (l2), then S = max s_j, N = max N_j, P = max p_j over the lane's
    rows (NaN = tombstoned rows drop out of v_max_f32), and the constants of query tile 0."""
    a = s.emit
    NR = 4 * MT
    if part == 0:
        if I8_SPACE == "l2":
            for j in range(NR):
                a(f"v_mul_f32 %[p{j}], %[r{j}], %[r{j}]")
            for j in range(NR):
                a(f"v_mul_f32 %[p{j}], %[k1], %[p{j}]")        # k1 = -(1 - slack)
    else:
        # (L2E: the rows' scales were loaded into the u registers, free until the first stub runs; only their maximum is kept)
        trees = [(L2IP_SMAX, "u" if L2E else "s"), (L2IP_NMAX, "r")] + ([(L2IP_PMAX, "p")] if I8_SPACE == "l2" else [])
        for dst, src in trees:
            a(f"v_max3_f32 {dst}, %[{src}0], %[{src}1], %[{src}2]")
            for j in range(3, NR - 1, 2):
                a(f"v_max3_f32 {dst}, {dst}, %[{src}{j}], %[{src}{j + 1}]")
            if (NR - 3) % 2:
                a(f"v_max_f32 {dst}, {dst}, %[{src}{NR - 1}]")
        l2ip_fetch(s, 0)


def gen_rowmax_l2c(s, part):
    """Start of the last k-step, l2c.  Loaded: u_j = the x-slots of the lane's 8 row pairs -- the group's scale S in the first
    panel's rows (u0..u3), the group's largest relative row error Bg in the second panel's (u4..u7), never NaN -- and r_j =
    |x_j| (NaN = dead row).  e10 = S SQ (the factor of the integer A = I + e); KEg = KEq + KEr Bg (the lane's error
    coefficient: the query part + the row part); r_j becomes c_j = KEg N_j + P0 with P0 = max_j -(1 - slack) N_j^2; e12 =
    max_j c_j.  Pre-test: float(max_j A_j) e10 + e12 >= thr; per row (stubs): float(A_j) e10 + c_j >= thr.  NaN rows drop out of
    every v_max; their c_j stays NaN and fails every compare."""
    a = s.emit
    NR = 4 * MT
    assert NR == 8

    def tree(dst, src):
        a(f"v_max3_f32 {dst}, %[{src}0], %[{src}1], %[{src}2]")
        a(f"v_max3_f32 {dst}, {dst}, %[{src}3], %[{src}4]")
        a(f"v_max3_f32 {dst}, {dst}, %[{src}5], %[{src}6]")
        a(f"v_max_f32 {dst}, {dst}, %[{src}7]")

    if part == 0:
        a("v_mul_f32 %[e10], %[sqc], %[u0]")                 # S SQ
        a("v_mov_b32 %[e4], %[kec]")                         # (one SGPR per VALU instruction: the constant-bus limit)
        a("v_fma_f32 %[e4], %[krc], %[u4], %[e4]")           # KEg = KEr Bg + KEq
        for j in range(NR):
            a(f"v_mul_f32 %[u{j}], %[r{j}], %[r{j}]")
        for j in range(NR):
            a(f"v_mul_f32 %[u{j}], 0xbf7fffe0, %[u{j}]")    # -(1 - 2^-19) = -(1.0f - kSlack): the constant of scan_epilogue and of filter_l2_offsets_kernel
    else:
        tree("%[e12]", "u")                                 # P0
        for j in range(NR):
            a(f"v_fma_f32 %[r{j}], %[e4], %[r{j}], %[e12]")
        tree("%[e12]", "r")


def gen_rowmax(s, part):
    """Start of the last k-step: p_j *= K, then R = max r_j and P = max p_j over this lane's rows (e10, e12)."""
    if L2C:
        return gen_rowmax_l2c(s, part)
    if I8_SPACE != "cosine":
        return gen_rowmax_l2ip(s, part)
    a = s.emit
    NR = 4 * MT
    if part == 0:
        for j in range(NR):
            a(f"v_mul_f32 %[p{j}], %[k1], %[p{j}]")
    else:
        for dst, src in (("%[e10]", "r"), ("%[e12]", "p")):
            a(f"v_max3_f32 {dst}, %[{src}0], %[{src}1], %[{src}2]")
            for j in range(3, NR - 1, 2):
                a(f"v_max3_f32 {dst}, {dst}, %[{src}{j}], %[{src}{j + 1}]")
            if (NR - 3) % 2:
                a(f"v_max_f32 {dst}, {dst}, %[{src}{NR - 1}]")


def dma_pieces(NW):
    """This wave's LDS-DMA transfers per chunk: (set name, index, k-step half, byte offset inside the chunk / the LDS buffer).
    A chunk is 2 * NQT fragments of 1 KiB at n * 2048 + h * 1024 (the image keeps the 16-tile layout whatever NQT is).
    NQT = 16: wave w moves tiles w and w + NW, both halves (4 transfers); NQT = 8: tile w, both halves (2); NQT = 4: ONE
    fragment -- tile w & 3, half w >> 2: the wrapper puts that into the wave's base offsets (qvoff / wave2k), offset 0 here.
    Every wave issues the same number of transfers: the counted vmcnt waits assume identical issue sequences."""
    if NQT == 16:
        KQ = 1024 // (NW * 64)
        return [(sn, i, half, i * NW * 2048 + half * 1024) for sn, half in (("qb", 0), ("qa", 1)) for i in range(KQ)]
    assert NW == 8
    if NQT == 8:
        return [("qb", 0, 0, 0), ("qa", 0, 1, 1024)]
    assert NQT == 4
    return [("qb", 0, 0, 0)]


def gen_chunk(s, R, QD, KQ, NW, step0, zero_first, last, nt, prio=False, dma=False, final=False, sync=True, ch=0):
    """One 64-column chunk = 2 k-steps = 32 fragments x MT MFMAs.  final: the tile's last chunk, whose second k-step
    carries the admission pre-tests (FUSE)."""
    if QA:   # the read base moved on when the previous chunk started reading this one's fragments (below)
        s.emit("s_add_u32 %[sldw], %[sldw], 0x8000")
        s.emit("s_and_b32 %[sldw], %[sldw], 0x1ffff")
    elif Q4 or Q3D:   # buffers 0..3 in rotation: this chunk reads the next one, its DMAs fill the one after the next (Q3D: one further)
        s.emit("v_add_u32 %[ldr], 0x8000, %[ldr]")
        s.emit("v_and_b32 %[ldr], 0x1ffff, %[ldr]")
        s.emit("s_add_u32 %[sldw], %[sldw], 0x8000")
        s.emit("s_and_b32 %[sldw], %[sldw], 0x1ffff")
    else:
        s.emit("v_xor_b32 %[ldr], 0x8000, %[ldr]")
        if dma:
            s.emit("s_xor_b32 %[sldw], %[sldw], 0x8000")
        else:
            s.emit("v_xor_b32 %[ldw], 0x8000, %[ldw]")

    NF = 2 * NQT   # fragments per chunk: NQT query tiles x 2 k-steps (k-step major)

    def read(f):
        h, n = f // NQT, f % NQT
        s.lds(f"ds_read_b128 %[t{f % QD}], %[ldr] offset:{n * 2048 + h * 1024}", ("rd", f))

    def refill(h):
        last_step = step0 + h
        if BURST:
            # burst mode: nothing until the last k-step of a group of BURST is done, then the whole group of the next
            # body at once -- BURST consecutive KiBs of each panel, issued back to back (DRAM page locality)
            if last_step % BURST != BURST - 1:
                return
            steps = list(range(last_step - BURST + 1, last_step + 1))
        else:
            steps = [last_step]
        pol = " nt" if nt else ""   # streamed once: non-temporal
        for idx, step in enumerate(steps):
            b = step % R
            for m in range(MT):
                if last and step * 1024 < 4096:
                    if m >= 2:
                        s.emit(f"s_mul_i32 %[st0], %[pb], {m}")
                    so = "0" if m == 0 else ("%[pb]" if m == 1 else "%[st0]")
                    if STAG:
                        so = "%[xrot]" if m == 0 else "%[pbrot]"
                    off = f" offset:{step * 1024}" if step else ""
                    s.vmem(f"buffer_load_dwordx4 {ring(b, m)}, %[lane16], {XNEXT}, {so} offen{off}{pol}", ("x", b, m))
                elif last:
                    if m == 0:
                        s.emit(f"s_movk_i32 %[st0], 0x{step * 1024:x}")
                    else:
                        if m >= 2:
                            s.emit(f"s_mul_i32 %[st0], %[pb], {m}")
                        s.emit(f"s_add_u32 %[st0], {'%[pb]' if m == 1 else '%[st0]'}, 0x{step * 1024:x}")
                    s.vmem(f"buffer_load_dwordx4 {ring(b, m)}, %[lane16], {XNEXT}, %[st0] offen{pol}", ("x", b, m))
                else:
                    off = f" offset:{idx * 1024}" if idx else ""
                    s.vmem(f"buffer_load_dwordx4 {ring(b, m)}, %[lane16], {XCUR}, %[xso{m}] offen{off}{pol}", ("x", b, m))
        if not last:
            for m in range(MT):
                s.emit(f"s_add_u32 %[xso{m}], %[xso{m}], 0x{0x400 * len(steps):x}")
            if STAG:   # a rotated k origin passes the end of the panel in mid-tile
                s.emit("s_cmp_eq_u32 %[xso0], %[pb]")
                s.emit("s_cselect_b32 %[xso0], 0, %[xso0]")
                s.emit("s_cmp_eq_u32 %[xso1], %[pb2]")
                s.emit("s_cselect_b32 %[xso1], %[pb], %[xso1]")

    # Staging plan, fragment index -> action.  Set qb (first halves) is written and re-fetched just
    # before the ring refill that follows fragment 15, set qa (second halves) just before the one
    # that follows fragment 31 (see the module docstring).
    plan = {}
    if dma:
        # LDS-DMA staging: chunk c+1 goes global -> LDS directly (buffer_load ... lds: LDS address = M0 +
        # 16*lane), early in chunk c; no staging registers, no ds_write (tools/probe: +5 % on the bare loop).
        f = 2
        for setname, i, half, const in dma_pieces(NW):
            plan[f] = ("d", setname, i, half, const)
            f += 1
    else:
        for setname, half, f in (("qb", 0, 16 - 2 * KQ), ("qa", 1, 32 - 2 * KQ)):
            for i in range(KQ):
                plan[f] = ("w", setname, i, half)
                f += 1
            for i in range(KQ):
                plan[f] = ("l", setname, i, half)
                f += 1

    if prio:
        s.emit("s_setprio 3")
    if not QA:   # (QA: fragments 0..QD-1 were read at the end of the previous chunk / by the prologue)
        for f0 in range(QD):
            read(f0)
    prio_steps = {k * NF // 32: v for k, v in PRIO_STEPS.items()}
    for f in range(NF):
        h, n = f // NQT, f % NQT
        b = (step0 + h) % R
        if prio and f in prio_steps:
            # Experiment (off by default).  The two waves of a SIMD share its MFMA pipe and issue is
            # arbitrated by priority, then age: left alone, the older wave (0..NW/2-1) runs its 64 MFMAs
            # of the chunk first and parks at the barrier for a third of its life (s_memtime stamps:
            # 35 % vs 5 % for the younger half).  Priority that falls with progress (3, 2, 1, 0 per
            # quarter chunk), the younger wave's steps half a quarter later, makes the two leapfrog
            # every 4 fragments and evens the parking out at 7 % -- and the scan gets 3 % SLOWER: the
            # kernel is power-bound (tools/probe), cycles saved come back as a lower clock.
            who, level = prio_steps[f]
            s.emit(f"s_cmp_eq_u32 %[wtype], {who}")
            s.emit(f"s_cbranch_scc0 .Lp{s.label}_%=")
            s.emit(f"s_setprio {level}")
            s.emit(f".Lp{s.label}_%=:")
            s.label += 1
        if n == 0:
            s.need_vm(*[("x", b, m) for m in range(MT)])
            if final and h == 1:   # issued before this body's ring refills: landed with them (vmcnt completes in order)
                s.need_vm(*[("rn", j) for j in range(4 * MT)])
        s.need_lg(("rd", f))
        for m in range(MT):
            c = (f"%[eo{m}]" if L2E else "0") if (zero_first and h == 0) else acc(m, n)
            op = "v_mfma_i32_16x16x64_i8" if I8 else "v_mfma_f32_16x16x32_bf16"
            if "nomfma" not in DBG:
                s.emit(f"{op} {acc(m, n)}, {ring(b, m)}, %[t{f % QD}], {c}")
            if final and h == 1:
                # the accumulators of query tile n - 2 are complete (their last MFMAs were issued four MFMAs ago)
                if n < 2:
                    if m == 1:
                        gen_rowmax(s, n)
                else:
                    gen_pretest(s, n - 2, m)
        if f + QD < NF:
            read(f + QD)
        elif QA:
            # the next chunk's first fragments: its buffer was published by the PREVIOUS barrier (staged two chunks
            # ahead), so the read stream never stops at a chunk boundary -- after the barrier the MFMAs go on at once
            # instead of both waves of the SIMD waiting out an LDS round trip with the pipe idle
            if f + QD == NF:
                s.emit("v_add_u32 %[ldr], 0x8000, %[ldr]")
                s.emit("v_and_b32 %[ldr], 0x1ffff, %[ldr]")
            read(f + QD - NF)
        if f in plan:
            kind, setname, i, half = plan[f][:4]
            reg = f"%[{setname}{i}]"
            const = plan[f][4] if len(plan[f]) > 4 else i * NW * 2048 + half * 1024
            if kind == "d":
                s.emit(f"s_add_u32 m0, %[sldw], 0x{const:x}")
                s.emit(f"s_add_u32 %[st0], %[qcur], 0x{const:x}")
                s.vmem(f"buffer_load_dwordx4 %[qvoff], %[qsrd], %[st0] offen lds", (setname, i, ch) if Q3D else (setname, i))
            elif kind == "w":
                s.need_vm((setname, i))
                s.lds(f"ds_write_b128 %[ldw], {reg} offset:{const}", ("wr", setname, i))
            else:
                s.emit(f"s_add_u32 %[st0], %[qcur], 0x{const:x}")
                s.vmem(f"buffer_load_dwordx4 {reg}, %[qvoff], %[qsrd], %[st0] offen", (setname, i))
        if n == NQT - 1:
            refill(h)
    if final:   # the last two query tiles: nothing left to hide behind
        gen_pretest(s, NQT - 2, 0)
        gen_pretest(s, NQT - 2, 1)
        s.emit("s_nop 7")   # XDL write (the last query tile's MFMAs, 8 instructions back) -> VALU read: 16 wait states with this
        #                     (an 8-pass MFMA needs 11; once per tile, so the margin is free)
        gen_pretest(s, NQT - 1, 0)
        gen_pretest(s, NQT - 1, 1)
    # advance the Q cursor (chunk c+2 -> c+3, wrapping) and publish the chunk just staged
    s.emit("s_add_u32 %[qcur], %[qcur], 0x8000")
    s.emit("s_cmp_eq_u32 %[qcur], %[qbytes]")
    s.emit("s_cselect_b32 %[qcur], 0, %[qcur]")
    if not sync:
        # Q4: no barrier after the first chunk of a pair.  The chunk read next was published by the previous barrier
        # (every wave waited for its share of it there), and the buffer this wave's DMAs are filling was last read two
        # chunks ago, before that same barrier.
        return
    if dma and Q3D:
        # the chunk read NEXT was staged two chunks ago: waiting for those transfers (vmcnt completes in order) only
        # forces the X refills issued before them, i.e. more than two chunks ago -- the ring of 6 k-steps gets its
        # full three chunks of latency cover instead of the one a wait for this chunk's own transfers leaves
        nch = R // 2
        s.need_vm(*[(sn, i, (ch + 1) % nch) for sn in ("qb", "qa") for i in range(KQ)])
    elif dma:   # this wave's share of the chunk(s) staged since the last barrier has landed in LDS
        s.need_vm(*[(sn, i) for sn, i, _, _ in dma_pieces(NW)])
    if not QA:   # (QA: the reads in flight are of the next chunk's buffer, which nobody writes for two more chunks)
        s.drain_lg()
    if "stamp" in DBG:   # cycles parked at the barrier, summed in an SGPR (timing diagnostic)
        s.emit("s_memtime s[78:79]")
        s.emit("s_waitcnt lgkmcnt(0)")
        s.emit("s_sub_u32 %[sacc0], %[sacc0], s78")
    s.emit("s_barrier")
    if "stamp" in DBG:
        s.emit("s_memtime s[78:79]")
        s.emit("s_waitcnt lgkmcnt(0)")
        s.emit("s_add_u32 %[sacc0], %[sacc0], s78")


def stage_only_chunk(KQ, NW):
    """A chunk period of a wave that has no tile to work on (stagger: the first half tile of the late waves, the last
    half tile of the early ones): its share of the next Q chunk (LDS-DMA), the buffer toggles and the barrier."""
    o = ["v_xor_b32 %[ldr], 0x8000, %[ldr]", "s_xor_b32 %[sldw], %[sldw], 0x8000"]
    for half in (0, 1):
        for i in range(KQ):
            const = i * NW * 2048 + half * 1024
            o += [f"s_add_u32 m0, %[sldw], 0x{const:x}", f"s_add_u32 %[st0], %[qcur], 0x{const:x}",
                  "buffer_load_dwordx4 %[qvoff], %[qsrd], %[st0] offen lds"]
    o += ["s_add_u32 %[qcur], %[qcur], 0x8000", "s_cmp_eq_u32 %[qcur], %[qbytes]", "s_cselect_b32 %[qcur], 0, %[qcur]",
          "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    return o


def gen_eo_loads(s, next_tile=True):
    """L2E: the per-row integer offsets of this wave's 32 rows of the NEXT tile (the last tile of a workgroup re-reads its own:
    harmless) -> the two 4-register tuples eo0 / eo1, which are the C operands of that tile's first k-step.  They live in the
    same allocation as the row pairs, 8 x capacity bytes further on; the SGPR `eo` holds that distance less 4 x (the wave's
    first row), so that one descriptor (the row pairs') serves both."""
    s.emit("v_lshrrev_b32 %[e0], 1, %[rnvoff]")       # 16 g: this lane's 4 rows x 4 bytes inside a panel's 64
    if next_tile:
        s.emit("s_lshr_b32 %[st0], %[rnstride], 1")    # the next tile's rows: 4 bytes per row where the pairs have 8
        s.emit("s_cmp_gt_u32 %[tl], 1")
        s.emit("s_cselect_b32 %[st0], %[st0], 0")
        s.emit("s_add_u32 %[st0], %[st0], %[eo]")
    for m in range(MT):
        so = "%[st0]" if next_tile else "%[eo]"
        s.vmem(f"buffer_load_dwordx4 %[eo{m}], %[e0], {RNS}, {so} offen" + (f" offset:{64 * m}" if m else ""), ("eo", m))


def gen_body(s, R, QD, KQ, NW, first, last, nt, prio=False, dma=False):
    if last and L2E and not first:
        # issued before the row-pair loads below: in-order completion makes the wait for those a wait for these too.  (The
        # offsets in the registers now were last read in this tile's FIRST k-step, which is not in this body.)
        gen_eo_loads(s)
    if last:
        # |x| of this lane's 4*MT rows (rows 4g..4g+3 of every panel) for the admission test
        for j in range(4 * MT):
            off = (j >> 2) * 64 + (j & 3) * 4
            if I8:
                # per-row pairs {a, b}: cosine a = sx/(|x|+1e-30) (b unused); l2 / ip a = sx, b = |x| (NaN: tombstoned)
                if I8_SPACE == "cosine":   # {sx/(|x|+1e-30), the row's own error}
                    s.vmem(f"buffer_load_dword %[r{j}], %[rnvoff], {RNS}, 0 offen" + (f" offset:{2 * off}" if off else ""), ("rn", j))
                    s.vmem(f"buffer_load_dword %[p{j}], %[rnvoff], {RNS}, 0 offen offset:{2 * off + 4}", ("rn", j))
                else:
                    sreg = f"u{j}" if L2E else f"s{j}"
                    s.vmem(f"buffer_load_dword %[{sreg}], %[rnvoff], {RNS}, 0 offen" + (f" offset:{2 * off}" if off else ""), ("rn", j))
                    s.vmem(f"buffer_load_dword %[r{j}], %[rnvoff], {RNS}, 0 offen offset:{2 * off + 4}", ("rn", j))
                continue
            s.vmem(f"buffer_load_dword %[r{j}], %[rnvoff], {RNS}, 0 offen" + (f" offset:{off}" if off else ""),
                   ("rn", j))
    for ch in range(R // 2):
        if last and L2E and first and ch == 1:   # a one-body tile: its own offsets were read in chunk 0's first k-step
            gen_eo_loads(s)
        gen_chunk(s, R, QD, KQ, NW, 2 * ch, first and ch == 0, last, nt, prio, dma, FUSE and last and ch == R // 2 - 1,
                  sync=not Q4 or ch % 2 == 1, ch=ch)
    if last:
        s.need_vm(*([("rn", j) for j in range(4 * MT)] + ([("eo", m) for m in range(MT)] if L2E else [])))


def body_lines(R, QD, KQ, NW, first, last, nt, prio=False, label0=0, dma=False):
    s = Sched()
    s.recording = False
    for _ in range(2):   # history: every predecessor issues this pattern of memory operations
        gen_body(s, R, QD, KQ, NW, False, False, nt, prio, dma)
    s.recording = True
    s.label = label0
    s.copy = str(label0)
    gen_body(s, R, QD, KQ, NW, first, last, nt, prio, dma)
    return s.lines


def gen_admission(space):
    """After the k-loop of a tile: bounds, quick reject per query tile, calls into .Lslow."""
    s = Sched()
    a = s.emit
    if not VA:
        a("s_nop 15")   # XDL write -> v_accvgpr_read of the accumulators
        a("s_nop 7")
    # (VA: the accumulators are ArchVGPRs, read by the VALU directly; query tile n's last MFMA was issued 2*(16-n)
    # MFMAs and 22*n vector instructions before its first read here, far beyond the XDL-write -> VALU-read distance)
    # ke = the query's error term from LDS (filter_prep_kernel).  Per-row constants (scan_epilogue):
    # cosine p0 = 1/(|x|+1e-30), u = a*p0 + ke; ip p0 = |x|, u = a + ke*p0; l2 p0 = |x|, p1 = -|x|^2 (1-slack),
    # u = sq*(a + ke*p0) + p1
    NR = 4 * MT
    if FUSE:
        return s.lines   # the pre-tests ran inside the last k-step (gen_pretest); .Lback<n> live there
    if I8 and space == "cosine":
        # int8 shadow, cosine: r_j = sx/(|x|+1e-30) of the row (NaN: tombstoned), p_j = the row's own rounding error,
        # accumulators = exact integer dot products I; the test is float(I)*r_j + p_j*K >= T[q] with T = (thr - ke8)/sq8
        # rounded down and K = (1 + max eq8)/min sq8 (filter_prep8_fin_kernel); the append path stores the left-hand side,
        # the kernel's tail (the in-kernel scatter) turns it into the bound u = w*sq8 + ke8
        if "noadm" in DBG:   # timing diagnostic: no admission test at all (labels only: the hit stubs refer to them)
            for n in range(NQT):
                a(f".Lback{n}_%=:")
            return s.lines
        for j in range(NR):
            a(f"v_mul_f32 %[p{j}], %[k1], %[p{j}]")
        s.lds(f"ds_read_b32 %[e0], %[thra]", ("thr", 0))
        for n in range(NQT):
            if n + 1 < NQT:
                s.lds(f"ds_read_b32 %[e{(n + 1) & 1}], %[thra] offset:{(n + 1) * 64}", ("thr", n + 1))
            for j in range(NR):
                m, i = j >> 2, j & 3
                if VA:
                    a(f"v_cvt_f32_i32 %[u{j}], {acc_reg(m, n, i)}")
                elif "noread" not in DBG:   # timing diagnostic: the test's arithmetic on stale registers
                    a(f"v_accvgpr_read_b32 %[u{j}], a{(m * 16 + n) * 4 + i}")
            for j in range(NR):
                if not VA:
                    a(f"v_cvt_f32_i32 %[u{j}], %[u{j}]")
            for j in range(NR):
                a(f"v_fma_f32 %[u{j}], %[u{j}], %[r{j}], %[p{j}]")
            a("v_max3_f32 %[e4], %[u0], %[u1], %[u2]")
            a("v_max3_f32 %[e5], %[u3], %[u4], %[u5]")
            for j in range(6, NR, 4):
                a(f"v_max3_f32 %[e4], %[u{j}], %[u{j + 1}], %[e4]")
                if j + 3 < NR:
                    a(f"v_max3_f32 %[e5], %[u{j + 2}], %[u{j + 3}], %[e5]")
            a("v_max_f32 %[e4], %[e4], %[e5]")
            s.need_lg(("thr", n))
            a(f"v_cmp_ge_f32 vcc, %[e4], %[e{n & 1}]")
            if "nohit" not in DBG:
                a(f"s_cbranch_vccnz .Lhit{n}_%=")
            a(f".Lback{n}_%=:")
        return s.lines
    for j in range(NR):
        if space == "cosine":
            a(f"v_add_f32 %[r{j}], 0x0da24260, %[r{j}]")   # + 1e-30f
            a(f"v_rcp_f32 %[r{j}], %[r{j}]")
        elif space == "l2":
            a(f"v_mul_f32 %[p{j}], %[r{j}], %[r{j}]")
            a(f"v_mul_f32 %[p{j}], %[k1], %[p{j}]")        # k1 = -(1 - slack)
    thr = lambda n: f"%[e{n & 1}]"
    sq = lambda n: f"%[e{2 + (n & 1)}]"
    ke = lambda n: f"%[e{10 if n & 1 else 12}]"

    def fetch(n):
        s.lds(f"ds_read_b32 {thr(n)}, %[thra] offset:{n * 64}", ("thr", n))
        s.lds(f"ds_read_b32 {ke(n)}, %[thra] offset:{2048 + n * 64}", ("ke", n))
        if space == "l2":
            s.lds(f"ds_read_b32 {sq(n)}, %[thra] offset:{1024 + n * 64}", ("sq", n))

    if "noadm" in DBG:   # timing diagnostic: no admission test at all (labels only: the hit stubs refer to them)
        for n in range(NQT):
            a(f".Lback{n}_%=:")
        return s.lines
    fetch(0)
    for n in range(NQT):
        if n + 1 < NQT:
            fetch(n + 1)
        for j in range(NR):
            m, i = j >> 2, j & 3
            if VA:
                a(f"v_cvt_f32_i32 %[u{j}], {acc_reg(m, n, i)}")
            else:
                a(f"v_accvgpr_read_b32 %[u{j}], a{(m * 16 + n) * 4 + i}")
        if I8:
            # int8 shadow, l2 / ip: w = float(I) * sx_j takes the place of the bf16 dot product; the per-query constants in
            # LDS are rescaled by the query's scale (filter_scan_asm_kernel): ip  w + ke' |x| >= thr/sq8 (the append path
            # stores that, the in-kernel scatter multiplies by sq8), l2  sq' (w + ke' |x|) + p1 >= thr
            for j in range(NR):
                if not VA:
                    a(f"v_cvt_f32_i32 %[u{j}], %[u{j}]")
            for j in range(NR):
                a(f"v_mul_f32 %[u{j}], %[u{j}], %[s{j}]")
        s.need_lg(("ke", n), *([("sq", n)] if space == "l2" else []))
        for j in range(NR):
            if space == "cosine":
                a(f"v_fma_f32 %[u{j}], %[u{j}], %[r{j}], {ke(n)}")
            elif space == "ip":
                a(f"v_fma_f32 %[u{j}], {ke(n)}, %[r{j}], %[u{j}]")
            else:
                a(f"v_fma_f32 %[u{j}], {ke(n)}, %[r{j}], %[u{j}]")
                a(f"v_fma_f32 %[u{j}], {sq(n)}, %[u{j}], %[p{j}]")
        a("v_max3_f32 %[e4], %[u0], %[u1], %[u2]")
        a("v_max3_f32 %[e5], %[u3], %[u4], %[u5]")
        for j in range(6, NR, 4):   # two dependency chains
            a(f"v_max3_f32 %[e4], %[u{j}], %[u{j + 1}], %[e4]")
            if j + 3 < NR:
                a(f"v_max3_f32 %[e5], %[u{j + 2}], %[u{j + 3}], %[e5]")
        a("v_max_f32 %[e4], %[e4], %[e5]")
        s.need_lg(("thr", n))
        a(f"v_cmp_ge_f32 vcc, %[e4], {thr(n)}")
        if "nohit" not in DBG:
            a(f"s_cbranch_vccnz .Lhit{n}_%=")
        a(f".Lback{n}_%=:")
    return s.lines


def gen_hit_stubs(copy=""):
    out = []
    for n in range(NQT):
        out.append(f".Lhit{n}{copy}_%=:")
        if FUSE and (I8_SPACE == "cosine" or L2C):   # the pre-test let a lane through: the 8 exact bounds of this query tile (gen_admission's arithmetic)
            for j in range(4 * MT):
                out.append(f"v_cvt_f32_i32 %[u{j}], {acc_reg(j >> 2, n, j & 3)}")
            for j in range(4 * MT):   # (l2c: float(A_j) S SQ + c_j, the pre-test's own arithmetic per row: gen_rowmax_l2c)
                out.append(f"v_fma_f32 %[u{j}], %[u{j}], " + ("%[e10], %[r" + str(j) + "]" if L2C else f"%[r{j}], %[p{j}]"))
            if EO and MT == 2:
                # Most calls are false alarms of the pre-test (it tests a row that dominates the lane's 8): one max tree
                # and one compare send those straight back, instead of through the append routine's 8 compares and 8
                # skipped row blocks (a taken branch each).  All 8 waves meet at the next barrier, so a tile is as slow
                # as the wave with the most stub calls: the call's length is what counts.
                out += ["v_max3_f32 %[e4], %[u0], %[u1], %[u2]", "v_max3_f32 %[e5], %[u3], %[u4], %[u5]",
                        "v_max3_f32 %[e4], %[u6], %[u7], %[e4]", "v_max_f32 %[e4], %[e4], %[e5]",
                        f"v_cmp_ge_f32 vcc, %[e4], %[tq{n}]", f"s_cbranch_vccz .Lback{n}{copy}_%="]
        elif FUSE:   # l2 / ip: the same, with the constants the pre-test holds in registers (gen_pretest_l2ip)
            thr, ke, sq = l2ip_consts(n)
            for j in range(4 * MT):
                out.append(f"v_cvt_f32_i32 %[u{j}], {acc_reg(j >> 2, n, j & 3)}")
            # (L2E: the pre-test's own arithmetic per row -- A_j = I_j + e_j, the lane's scale S, the lane's P0 -- is an upper
            # bound of the row's score and is what gets appended: no offset has to be recovered here)
            for j in range(4 * MT):
                out.append(f"v_mul_f32 %[u{j}], %[u{j}], " + (L2IP_SMAX if L2E else f"%[s{j}]"))
            for j in range(4 * MT):
                out.append(f"v_fma_f32 %[u{j}], {ke}, %[r{j}], %[u{j}]")
            if I8_SPACE == "l2":
                for j in range(4 * MT):
                    out.append(f"v_fma_f32 %[u{j}], {sq}, %[u{j}], " + (L2IP_PMAX if L2E else f"%[p{j}]"))
            if EO and L2E and MT == 2:   # false alarms of the pre-test leave at once (as the cosine stubs do)
                out += ["v_max3_f32 %[e7], %[u0], %[u1], %[u2]", "v_max3_f32 %[e5], %[u3], %[u4], %[u5]",
                        "v_max3_f32 %[e7], %[u6], %[u7], %[e7]", "v_max_f32 %[e7], %[e7], %[e5]",
                        f"v_cmp_ge_f32 vcc, %[e7], {thr}", f"s_cbranch_vccz .Lback{n}{copy}_%="]
        thr_src = (f"%[tq{n}]" if (I8_SPACE == "cosine" or L2C) else l2ip_consts(n)[0]) if FUSE else f"%[e{n & 1}]"
        out += [f"v_mov_b32 %[e6], {thr_src}",          # the threshold of this query tile
                f"s_movk_i32 %[sn64], 0x{n * 16:x}",      # first query of this tile
                f"s_getpc_b64 {RET}",
                "s_add_u32 s92, s92, 12",                 # return to the instruction after the branch below
                "s_addc_u32 s93, s93, 0",
                "s_branch .Lslow_%=",
                f"s_branch .Lback{n}{copy}_%="]
    return out


def lds_stage_cap(NW, mt=2, qbufs=None):
    """Entries of a wave's staging area in LDS (12 B each, SoA): what is left of the 160 KiB per CU."""
    qbufs = qbufs or (4 if (Q4 or Q3D or QA) else 2)
    wgs_per_cu = (16 // mt) // NW      # mt = 2: two waves per SIMD, mt = 4: one
    per_wg = (160 * 1024) // wgs_per_cu - (qbufs * CHUNK_BYTES + 3072)   # Q buffers + thr[256], qscale[256], ke[256]
    return min(WG_CAP // NW, (per_wg // NW) // 12 // 8 * 8)


def gen_slow(NW):
    """u0.. = bounds of this lane's 4*MT rows for query sn64 + c16, e6 = threshold.

    Wave-private append: the wave keeps its entry count in an SGPR; per row j the passing lanes form
    an SGPR mask and take the slots count + (passing lanes below), via v_mbcnt -- no atomics.
    The first LCW entries of a launch are staged in LDS (u[], row[], q[]) and copied to the wave's
    global buffer when the kernel ends: a global store inside the loop would sit in the vmcnt queue
    behind the prefetched loads for ~1-2 us and turn every counted wait into a drain of the X
    prefetch (measured: 17 % of the scan).  Entries beyond the staging area go straight to global
    memory (same slot numbering), entries beyond the global buffer flag their query as overflowed."""
    capw = WG_CAP // NW
    lcw = lds_stage_cap(NW, MT)
    o = [".Lslow_%=:",
         "v_add_u32 %[e9], %[sn64], %[c16v]",                       # e9 = query
         "v_add_u32 %[e11], %[trow], %[crow]"]                      # e11 = this lane's first row
    for j0 in range(0, 4 * MT, 8):                                  # 8 rows (mask registers) at a time
        for j in range(j0, j0 + 8):
            o.append(f"v_cmp_ge_f32_e64 s[{60 + 2 * (j - j0)}:{61 + 2 * (j - j0)}], %[u{j}], %[e6]")
        for j in range(j0, j0 + 8):
            lo, hi = 60 + 2 * (j - j0), 61 + 2 * (j - j0)
            o += [f"s_bcnt1_i32_b64 %[st0], s[{lo}:{hi}]",
                  f"s_cbranch_scc0 .Lskip{j}_%=",
                  f"s_mov_b64 exec, s[{lo}:{hi}]",
                  f"v_mbcnt_lo_u32_b32 %[e8], s{lo}, 0",
                  f"v_mbcnt_hi_u32_b32 %[e8], s{hi}, %[e8]",
                  "v_add_u32 %[e8], %[wcnt], %[e8]",                    # e8 = this entry's slot
                  f"v_add_u32 %[e5], {16 * (j >> 2) + (j & 3)}, %[e11]",  # e5 = row
                  f"v_cmp_gt_u32 vcc, 0x{lcw:x}, %[e8]",
                  "s_and_b64 exec, exec, vcc",                          # slots inside the LDS staging area
                  "v_lshl_add_u32 %[e7], %[e8], 2, %[stg]",
                  f"ds_write_b32 %[e7], %[u{j}]",
                  f"ds_write_b32 %[e7], %[e5] offset:{lcw * 4}",
                  f"ds_write_b32 %[e7], %[e9] offset:{lcw * 8}",
                  f"s_andn2_b64 exec, s[{lo}:{hi}], vcc",               # the rest
                  f"s_cbranch_execz .Lnext{j}_%=",
                  f"v_cmp_gt_u32 vcc, 0x{capw:x}, %[e8]",
                  f"s_mov_b64 s[76:77], exec",
                  "s_and_b64 exec, exec, vcc",                          # slots inside the global buffer
                  "v_lshlrev_b32 %[e7], 2, %[e8]",
                  f"global_store_dword %[e7], %[u{j}], %[wgbu]",
                  "global_store_dword %[e7], %[e5], %[wgbr]",
                  "global_store_dword %[e7], %[e9], %[wgbq]",
                  "s_andn2_b64 exec, s[76:77], vcc",                    # slots past the buffer
                  "v_lshlrev_b32 %[e7], 2, %[e9]",
                  "v_mov_b32 %[e5], 1",
                  "global_store_dword %[e7], %[e5], %[ovfb]",           # overflow[q] = 1: the query is re-run exactly
                  "s_mov_b64 exec, -1",
                  "s_waitcnt vmcnt(0)",   # stores may complete before older loads: no counted vmcnt wait may see them
                  f".Lnext{j}_%=:",
                  "s_add_u32 %[wcnt], %[wcnt], %[st0]",
                  f".Lskip{j}_%=:",
                  "s_mov_b64 exec, -1"]
    o += ["s_mov_b64 exec, -1", f"s_setpc_b64 {RET}"]
    return o


def gen_slow_fast(NW):
    """gen_slow with the common case as a straight line (FS).  A call appends, typically, ONE entry: one row of one lane.
    gen_slow walks 8 row blocks and skips the 7 empty ones with a taken branch each, and inside the one block that has a hit
    it takes another (no entry beyond the LDS staging area); all 8 waves of the workgroup meet at the next chunk barrier, so
    a tile is as slow as the wave with the most calls (phase stamps, profiles/r03: 11.2 us per tile in the second scan round,
    8 calls per wave and tile, against 9.3 us in the third with 1.3).  Here an empty row costs two scalar instructions and a
    branch that is NOT taken; the row blocks sit out of line, their own rare part (slots beyond the staging area) too."""
    capw = WG_CAP // NW
    lcw = lds_stage_cap(NW, MT)
    NR = 4 * MT
    assert NR == 8
    o = [".Lslow_%=:",
         "v_add_u32 %[e9], %[sn64], %[c16v]",                       # e9 = query
         "v_add_u32 %[e11], %[trow], %[crow]"]                      # e11 = this lane's first row
    for j in range(NR):
        o.append(f"v_cmp_ge_f32_e64 s[{60 + 2 * j}:{61 + 2 * j}], %[u{j}], %[e6]")
    for j in range(NR):
        lo, hi = 60 + 2 * j, 61 + 2 * j
        o += [f"s_cmp_lg_u64 s[{lo}:{hi}], 0",
              f"s_cbranch_scc1 .Lrow{j}_%=",
              f".Lrowret{j}_%=:"]
    o.append(f"s_setpc_b64 {RET}")
    for j in range(NR):
        lo, hi = 60 + 2 * j, 61 + 2 * j
        o += [f".Lrow{j}_%=:",
              f"s_bcnt1_i32_b64 %[st0], s[{lo}:{hi}]",
              f"s_mov_b64 exec, s[{lo}:{hi}]",
              f"v_mbcnt_lo_u32_b32 %[e8], s{lo}, 0",
              f"v_mbcnt_hi_u32_b32 %[e8], s{hi}, %[e8]",
              "v_add_u32 %[e8], %[wcnt], %[e8]",                    # e8 = this entry's slot
              f"v_add_u32 %[e5], {16 * (j >> 2) + (j & 3)}, %[e11]",  # e5 = row
              f"v_cmp_gt_u32 vcc, 0x{lcw:x}, %[e8]",
              "s_and_b64 exec, exec, vcc",                          # slots inside the LDS staging area
              "v_lshl_add_u32 %[e7], %[e8], 2, %[stg]",
              f"ds_write_b32 %[e7], %[u{j}]",
              f"ds_write_b32 %[e7], %[e5] offset:{lcw * 4}",
              f"ds_write_b32 %[e7], %[e9] offset:{lcw * 8}",
              f"s_andn2_b64 exec, s[{lo}:{hi}], vcc",               # the rest: none, unless the staging area is full
              f"s_cbranch_execnz .Lovf{j}_%=",
              f".Lovfret{j}_%=:",
              "s_add_u32 %[wcnt], %[wcnt], %[st0]",
              "s_mov_b64 exec, -1",
              f"s_branch .Lrowret{j}_%="]
    for j in range(NR):
        o += [f".Lovf{j}_%=:",
              f"v_cmp_gt_u32 vcc, 0x{capw:x}, %[e8]",
              "s_mov_b64 s[76:77], exec",
              "s_and_b64 exec, exec, vcc",                          # slots inside the global buffer
              "v_lshlrev_b32 %[e7], 2, %[e8]",
              f"global_store_dword %[e7], %[u{j}], %[wgbu]",
              "global_store_dword %[e7], %[e5], %[wgbr]",
              "global_store_dword %[e7], %[e9], %[wgbq]",
              "s_andn2_b64 exec, s[76:77], vcc",                    # slots past the buffer
              "v_lshlrev_b32 %[e7], 2, %[e9]",
              "v_mov_b32 %[e5], 1",
              "global_store_dword %[e7], %[e5], %[ovfb]",           # overflow[q] = 1: the query is re-run exactly
              "s_mov_b64 exec, -1",
              "s_waitcnt vmcnt(0)",   # stores may complete before older loads: no counted vmcnt wait may see them
              f"s_branch .Lovfret{j}_%="]
    return o


def gen_flush(NW):
    """Kernel end.  The entries a wave staged in LDS are moved into the per-query candidate lists by the C++ tail of
    filter_scan_asm_kernel (the workgroup's own scatter: LDS histogram, one device atomic per query it has entries
    for) -- no separate scatter launch, no round trip of the entries through global memory.  The assembly only has to
    make sure everything it issued has landed; the wave's entry count leaves through the wcnt operand."""
    return ["s_waitcnt vmcnt(0) lgkmcnt(0)"]   # ring / Q sets still in flight that nobody consumes; staged entries landed


def generate(space, R, QD, NW, nt=False, prio=False, mt=2, dma=False, stag=False, i8=False, va=False, q4=False, place=None, burst=0, q3d=False,
             qa=False, eo=False, fs=False, nqt=16, l2e=False, l2c=False):
    """stag: both waves of a SIMD reach the admission test (VALU only) together and leave the MFMA pipe idle for it.
    With the stagger the later-dispatched half of a workgroup's waves (wtype 1) runs half a tile behind: it sits out
    the first nkc/2 chunk periods (staging only), starts every row tile at column ld/2 (k origin rotated by xrot,
    wrapping at the end of the panel; the shared Q chunk stream is the same for everybody) and therefore reaches its
    admission test while its SIMD partner is in mid-tile; the early half sits out nkc/2 periods at the end.
    hc = 0 (and xrot = 0) turns it off at run time."""
    global MT, STAG, I8, I8_SPACE, VA, NQT
    # nqt (int8 VA bodies): passes of <= 64 / <= 128 queries compute 4 / 8 of the 16 query tiles.  The empty tiles' MFMAs, B
    # reads, Q transfers and admission tests are simply not generated; chunks, barriers, the ring and the image layout stay
    # (a chunk period is then ~3,300 cycles of HBM stream against ~250 of MFMAs: the body is a pure stream of the shadow).
    NQT = nqt
    assert nqt == 16 or (nqt in (4, 8) and i8 and va and dma and NW == 8 and not (stag or q4 or q3d or qa or burst))
    MT = mt
    STAG = stag
    I8 = i8
    I8_SPACE = space if i8 else None
    # va (int8 bodies): the 64*MT accumulator registers are ArchVGPRs, named explicitly (v[VA_BASE:...], clobbered), and
    # the MFMA operands -- the X ring and the B fragments, which only loads and MFMAs ever touch -- are AccVGPRs
    # (loads may target them: MUBUF / DS acc bit).  An MFMA's C and D must be of one register class (the assembler
    # rejects v-dst with a-srcC), so "results straight into VGPRs" means the whole accumulator lives there; the
    # admission test then reads it with no v_accvgpr_read and no XDL drain.  With two waves per SIMD the kernel
    # descriptor becomes 192 ArchVGPRs + 48 AccVGPRs (accum_offset 192) instead of hipcc's 128 / 128 split.
    VA = va
    assert not va or (i8 and mt == 2 and dma)
    global FUSE, Q4
    # (l2 keeps the serial test after the k-loop: its row term -|x|^2 differs too much between the 8 rows of a lane for the
    # dominating-row pre-test -- measured on 10M x 768: scan 2.002 ms per wave folded against 1.895 for round 1's body,
    # profiles/r02/scan_ab_l2_ip_folded_pretest_tried.txt; "fuse_l2" in DBG regenerates that variant)
    # l2e (round 4): l2 folded too.  The exact test of row j is  sq (I_j S + ke N_j) + p_j >= thr,  p_j = -(1 - slack) N_j^2; the
    # dominating-row pre-test with P0 = max_j p_j was loose by the spread of the norms inside a lane (one sigma of the score).
    # With e_j = ceil((p_j - P0) / (SQ S)) + 1 (SQ = the pass's largest sq; <= 0; filter_l2_offsets_kernel, once per pass)
    # added to the integer dot product -- the first k-step's MFMAs take the lane's e_j as their C operand instead of 0 --
    #   u'_j = sq ((I_j + e_j) S + ke N_j) + P0  >=  sq (I_j S + ke N_j) + p_j      for every query of the pass (sq <= SQ),
    # an upper bound of the row's score that differs from the exact test's by ~two quanta sq S; it is what the pre-test
    # dominates (one scale S per lane: the shadow builder groups l2 rows like ip's) and what the stubs append.
    # l2c: l2e where the prep quantises every query of the pass with one step, so that sq = SQ for all of them, and one
    # error coefficient KE = max_q 2 |q| ke_q stands for every query's: u'_j = float(I_j + e_j) (S SQ) + (KE N_j + P0).  The
    # per-query constants shrink to the threshold, kept in registers for the launch like cosine's; S SQ and KE N_j + P0 are
    # formed once per row tile (gen_rowmax_l2c).  SQ, KE: scalars of the pass (filter_l2_offsets_kernel).
    global L2E, L2C
    L2C = bool(l2c)
    L2E = bool(l2e) or L2C
    assert not L2E or (space == "l2" and va and R == 4)
    FUSE = va and "noadm" not in DBG and (space in ("cosine", "ip") or "fuse_l2" in DBG or L2E)
    # q4: four 32 KiB Q buffers (128 KiB), chunk c + 2 is staged while chunk c is consumed, and the workgroup meets at
    # ONE barrier per two chunks (after the odd ones) instead of one per chunk: half the parking, half the refills of
    # the software pipeline.  Needs the DMA staging and a ring of 4 k-steps (one loop body = one pair of chunks).
    Q4 = q4
    assert not q4 or (dma and R == 4 and not stag)
    # q3d: four Q buffers, chunk c + 3 is staged while chunk c is consumed, one barrier per chunk, and the wait before
    # that barrier is for the transfers issued TWO chunks ago (the chunk read next).  Why: vmcnt completes in order, so
    # waiting for this chunk's own transfers (the two-buffer scheme) also waits for every X refill issued before them --
    # whatever the ring depth, an X load then has about one chunk period to arrive (a ring of 6 brought 0.9 %).
    global Q3D
    Q3D = q3d
    assert not q3d or (dma and R == 6 and not stag and not q4)
    # qa: see the QA flag.  Why: with two Q buffers a chunk's first B fragments can only be read after the barrier that
    # publishes it, so after every barrier both waves of a SIMD wait for an LDS round trip (~200 cycles with all eight waves
    # asking at once) before their first MFMA -- ~7 % of a chunk period with the matrix pipe idle.  Staged two chunks ahead,
    # the chunk read next is already public one barrier earlier.
    global QA, EO, FS
    QA = qa
    EO = eo
    FS = fs
    assert not qa or (dma and R == 4 and not stag and not q4 and not q3d and va)
    global BURST
    BURST = burst
    assert not burst or (R % burst == 0 and not stag)
    assert R in (2, 4, 6) and 2 <= QD <= 8 and mt in (2, 4)
    assert not stag or (dma and mt == 2 and R * 1024 <= 4096)
    KQ = 1024 // (NW * 64)
    out = []
    a = out.append
    if place is not None:
        # code placement (MI355X_MICROARCH.md, "Two waves per SIMD" item 8: a hand-written stream can lose 13 % when its
        # 8-byte instructions sit at addresses = 4 mod 8): pin the phase of the statement, optionally shifted by 4 bytes
        a(".p2align 6")
        for _ in range(place):
            a("s_nop 0")
    # ---- descriptors and per-workgroup state
    a("s_mov_b32 s80, %[xlo]")
    a("s_mov_b32 s81, %[xhi]")
    a("s_mov_b32 s82, %[wbytes]")
    a("s_mov_b32 s83, 0x00020000")
    a("s_mov_b32 s86, s82")
    a("s_mov_b32 s87, s83")
    a("s_mov_b32 s88, %[rnlo]")
    a("s_mov_b32 s89, %[rnhi]")
    if L2E:   # the offsets are read through the same descriptor, 8 x capacity bytes further on: no range to check against
        a("s_mov_b32 s90, -1")
        a("s_mov_b32 %[eo], %[eo0in]")
    else:
        a(f"s_movk_i32 s90, 0x{16 * MT * (8 if i8 else 4):x}")   # this wave's rows x 4 B (int8 shadow: pairs)
    a("s_mov_b32 s91, s83")
    a("s_mov_b32 %[tl], %[ntiles]")
    a("s_mov_b32 %[trow], %[row0]")
    a("s_mov_b32 %[wcnt], 0")
    if "stamp" in DBG:
        a("s_mov_b32 %[sacc0], 0")
        a("s_memtime s[78:79]")
        a("s_waitcnt lgkmcnt(0)")
        a("s_mov_b32 %[sacc1], s78")
    if qa:
        a("v_mov_b32 %[ldr], %[lane16]")   # buffer 0: the prologue below reads its first fragments itself
    else:
        a("v_add_u32 %[ldr], 0x18000, %[lane16]" if (q4 or q3d) else "v_add_u32 %[ldr], 0x8000, %[lane16]")   # the first chunk moves it to buffer 0
    if not dma:
        a("v_mov_b32 %[ldw], %[qvoff]")        # ... and this one to buffer 1
    # ---- prologue: Q chunk 0 -> LDS buffer 0, chunk 1 -> the sets (register staging only), k-steps 0..R-1 -> the ring
    pieces = ([(const, setname, i) for setname, i, _, const in dma_pieces(NW)] if dma else
              [(i * NW * 2048 + half * 1024, setname, i) for half, setname in ((0, "qb"), (1, "qa")) for i in range(KQ)])
    if dma:
        a("s_mov_b32 %[sldw], %[wave2k]")          # buffer 0; the first chunk toggles it to buffer 1
        for const, setname, i in pieces:
            a(f"s_add_u32 m0, %[sldw], 0x{const:x}")
            a(f"s_movk_i32 %[st0], 0x{const:x}")
            a("buffer_load_dwordx4 %[qvoff], %[qsrd], %[st0] offen lds")
        if q4 or q3d or qa:   # chunk 1 -> buffer 1 as well; the first chunk then moves the write base to buffer 2
            a("s_add_u32 %[sldw], %[sldw], 0x8000")
            for const, setname, i in pieces:
                a(f"s_add_u32 m0, %[sldw], 0x{const:x}")
                a(f"s_add_u32 %[st0], %[qc1], 0x{const:x}")
                a("buffer_load_dwordx4 %[qvoff], %[qsrd], %[st0] offen lds")
        if q3d:  # ... and chunk 2 -> buffer 2; the first chunk moves the write base to buffer 3.  The Q cursor (the chunk
            #     staged next: 3, 4, ... mod the image's chunks) lives in sacc1 between tiles
            a("s_add_u32 %[sldw], %[sldw], 0x8000")
            for const, setname, i in pieces:
                a(f"s_add_u32 m0, %[sldw], 0x{const:x}")
                a(f"s_add_u32 %[st0], %[qcur0], 0x{const:x}")
                a("buffer_load_dwordx4 %[qvoff], %[qsrd], %[st0] offen lds")
            a("s_add_u32 %[qcur], %[qcur0], 0x8000")
            a("s_cmp_eq_u32 %[qcur], %[qbytes]")
            a("s_cselect_b32 %[qcur], 0, %[qcur]")
    else:
        for const, setname, i in pieces:
            a(f"s_movk_i32 %[st0], 0x{const:x}")
            a(f"buffer_load_dwordx4 %[{setname}{i}], %[qvoff], %[qsrd], %[st0] offen")
    for b in range(R):
        for m in range(MT):
            if m >= 2:
                a(f"s_mul_i32 %[st0], %[pb], {m}")
            if b * 1024 < 4096:
                so = "0" if m == 0 else ("%[pb]" if m == 1 else "%[st0]")
                if stag:
                    so = "%[xrot]" if m == 0 else "%[pbrot]"
                off = f" offset:{b * 1024}" if b else ""
                a(f"buffer_load_dwordx4 {ring(b, m)}, %[lane16], {XCUR}, {so} offen{off}")
            else:
                a(f"s_movk_i32 %[st0], 0x{b * 1024:x}" if m == 0 else
                  f"s_add_u32 %[st0], {'%[pb]' if m == 1 else '%[st0]'}, 0x{b * 1024:x}")
                a(f"buffer_load_dwordx4 {ring(b, m)}, %[lane16], {XCUR}, %[st0] offen")
    if L2E:   # the first tile's offsets (every later tile's are fetched a tile ahead: gen_eo_loads)
        a("v_lshrrev_b32 %[e0], 1, %[rnvoff]")
        for m in range(MT):
            a(f"buffer_load_dwordx4 %[eo{m}], %[e0], {RNS}, %[eo] offen" + (f" offset:{64 * m}" if m else ""))
    a("s_waitcnt vmcnt(0)")
    if not dma:
        for const, setname, i in pieces:
            a(f"ds_write_b128 %[ldw], %[{setname}{i}] offset:{const}")
        for const, setname, i in pieces:
            a(f"s_add_u32 %[st0], %[qc1], 0x{const:x}")
            a(f"buffer_load_dwordx4 %[{setname}{i}], %[qvoff], %[qsrd], %[st0] offen")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")   # counted waits below assume the steady-state issue pattern
    a("s_barrier")
    if FUSE and (space == "cosine" or L2C):   # the thresholds of this lane's query column in the 16 query tiles: constant for the whole launch
        for n in range(NQT):
            a(f"ds_read_b32 %[tq{n}], %[thra]" + (f" offset:{n * 64}" if n else ""))
        a("s_waitcnt lgkmcnt(0)")
    if qa:   # the first tile's first fragments (every later chunk's are read at the end of the chunk before it)
        for f0 in range(QD):
            a(f"ds_read_b128 %[t{f0}], %[ldr] offset:{f0 * 2048}")
    if stag:
        a("s_mov_b32 %[qcur], %[qc1]")   # the Q cursor follows the shared chunk stream from here on, never reset
        a("s_cmp_eq_u32 %[wtype], 1")
        a("s_cbranch_scc0 .Lnopre_%=")
        a("s_cmp_eq_u32 %[hc], 0")
        a("s_cbranch_scc1 .Lnopre_%=")
        a("s_mov_b32 %[cnt], %[hc]")
        a(".Lpre_%=:")
        out += stage_only_chunk(KQ, NW)
        a("s_sub_u32 %[cnt], %[cnt], 1")
        a("s_cmp_lg_u32 %[cnt], 0")
        a("s_cbranch_scc1 .Lpre_%=")
        a(".Lnopre_%=:")
    # ---- persistent loop over this workgroup's tiles
    a(".Ltile_%=:")
    a("s_cmp_gt_u32 %[tl], 1")           # next tile's descriptor (the last tile re-reads itself: harmless)
    a("s_cselect_b32 %[st0], %[xslo], 0")
    a("s_cselect_b32 %[cnt], %[xshi], 0")
    a("s_add_u32 s84, s80, %[st0]")
    a("s_addc_u32 s85, s81, %[cnt]")
    if not stag and not q3d:   # (q3d: the cursor simply keeps running: a tile is a whole number of image periods)
        a("s_mov_b32 %[qcur], %[qc1]" if dma and not (q4 or qa) else "s_mov_b32 %[qcur], %[qcur0]")   # first chunk staged inside this tile's loop
    a(f"s_add_u32 %[xso0], %[xrot], 0x{R * 1024:x}" if stag else f"s_movk_i32 %[xso0], 0x{R * 1024:x}")
    a("s_add_u32 %[xso1], %[pb], %[xso0]")
    for m in range(2, MT):
        a(f"s_mul_i32 %[xso{m}], %[pb], {m}")
        a(f"s_add_u32 %[xso{m}], %[xso{m}], %[xso0]")
    a("s_cmp_eq_u32 %[nb], 1")
    a("s_cbranch_scc1 .Lsingle_%=")
    out += body_lines(R, QD, KQ, NW, True, False, nt, prio, 0, dma)
    a("s_sub_u32 %[cnt], %[nb], 2")
    a(".Lloop_%=:")
    a("s_cmp_eq_u32 %[cnt], 0")
    a("s_cbranch_scc1 .Llast_%=")
    out += body_lines(R, QD, KQ, NW, False, False, nt, prio, 100, dma)
    a("s_sub_u32 %[cnt], %[cnt], 1")
    a("s_branch .Lloop_%=")
    a(".Llast_%=:")
    out += body_lines(R, QD, KQ, NW, False, True, nt, prio, 200, dma)
    a("s_branch .Ladmit_%=")
    a(".Lsingle_%=:")
    out += body_lines(R, QD, KQ, NW, True, True, nt, prio, 300, dma)
    a(".Ladmit_%=:")
    if prio:
        a("s_setprio 0")
    out += gen_admission(space)
    a("s_mov_b32 s80, s84")
    a("s_mov_b32 s81, s85")
    a("s_add_u32 s88, s88, %[rnstride]")
    a("s_addc_u32 s89, s89, 0")
    if L2E:   # the pairs' base moved on by 8 bytes per row, the offsets' by 4: the distance shrinks by the difference
        a("s_lshr_b32 %[st0], %[rnstride], 1")
        a("s_sub_u32 %[eo], %[eo], %[st0]")
    a("s_add_u32 %[trow], %[trow], %[rowstride]")
    a("s_sub_u32 %[tl], %[tl], 1")
    a("s_cmp_lg_u32 %[tl], 0")
    a("s_cbranch_scc1 .Ltile_%=")
    if stag:
        a("s_cmp_eq_u32 %[wtype], 0")
        a("s_cbranch_scc0 .Lnopost_%=")
        a("s_cmp_eq_u32 %[hc], 0")
        a("s_cbranch_scc1 .Lnopost_%=")
        a("s_mov_b32 %[cnt], %[hc]")
        a(".Lpost_%=:")
        out += stage_only_chunk(KQ, NW)
        a("s_sub_u32 %[cnt], %[cnt], 1")
        a("s_cmp_lg_u32 %[cnt], 0")
        a("s_cbranch_scc1 .Lpost_%=")
        a(".Lnopost_%=:")
    out += gen_flush(NW)
    if "stamp" in DBG:   # q[cap-2] = cycles parked at barriers, q[cap-1] = cycles of the whole kernel body
        a("s_memtime s[78:79]")
        a("s_waitcnt lgkmcnt(0)")
        a("s_sub_u32 %[sacc1], s78, %[sacc1]")
        a(f"v_mov_b32 %[e1], 0x{(WG_CAP // NW - 2) * 4:x}")
        a("v_mov_b32 %[e2], %[sacc0]")
        a("v_mov_b32 %[e3], %[sacc1]")
        a("global_store_dword %[e1], %[e2], %[wgbq]")
        a("global_store_dword %[e1], %[e3], %[wgbq] offset:4")
        a("s_waitcnt vmcnt(0)")
    a("s_branch .Ldone_%=")
    if FUSE:   # one set per copy of the tile's last body (.Llast / .Lsingle): a stub returns into its copy
        out += gen_hit_stubs("c200") + gen_hit_stubs("c300")
    else:
        out += gen_hit_stubs()
    out += gen_slow_fast(NW) if (FS and MT == 2) else gen_slow(NW)
    a(".Ldone_%=:")

    ops_out, ops_in = [], []
    opc = "a" if va else "v"   # register class of the MFMA A / B operands
    for b in range(R):
        for m in range(MT):
            ops_out.append(f'[x{b * MT + m}] "=&{opc}"(xring[{b * MT + m}])')
    if not dma:
        for i in range(KQ):
            ops_out.append(f'[qa{i}] "=&v"(qsa[{i}])')
        for i in range(KQ):
            ops_out.append(f'[qb{i}] "=&v"(qsb[{i}])')
    for i in range(QD):
        ops_out.append(f'[t{i}] "=&{opc}"(qt[{i}])')
    for j in range(4 * MT):
        ops_out.append(f'[r{j}] "=&v"(vr[{j}])')
    if (space == "l2" and not L2C) or (i8 and space == "cosine"):
        for j in range(4 * MT):
            ops_out.append(f'[p{j}] "=&v"(vp[{j}])')
    if i8 and space != "cosine" and not L2E:
        for j in range(4 * MT):
            ops_out.append(f'[s{j}] "=&v"(vs[{j}])')
    if L2E:
        for m in range(MT):
            ops_out.append(f'[eo{m}] "=&v"(veo[{m}])')
        ops_out.append('[eo] "=&s"(s_eo)')
    for j in range(4 * MT):
        ops_out.append(f'[u{j}] "=&v"(vu[{j}])')
    for j in range(13):
        ops_out.append(f'[e{j}] "=&v"(ve[{j}])')
    if FUSE:
        for n in range(NQT if (space == "cosine" or L2C) else L2IP_TQ):
            ops_out.append(f'[tq{n}] "=&v"(vt[{n}])')
    ops_out += ['[ldr] "=&v"(ldr)'] + (['[sldw] "=&s"(s_sldw)'] if dma else ['[ldw] "=&v"(ldw)'])
    for name in [f"xso{m}" for m in range(MT)] + ["qcur", "cnt", "st0", "tl", "trow", "sn64", "wcnt", "sacc0", "sacc1"]:
        ops_out.append(f'[{name}] "=&s"(s_{name})')
    ops_in += ['[qsrd] "s"(qsrd)', '[lane16] "v"(lane16)', '[qvoff] "v"(qvoff)', '[rnvoff] "v"(rnvoff)',
               '[thra] "v"(thra)', '[c16v] "v"(c16v)', '[crow] "v"(crow)', '[stg] "v"(stg)',
               '[xlo] "s"(xlo)', '[xhi] "s"(xhi)', '[wbytes] "s"(wbytes)', '[xslo] "s"(xslo)', '[xshi] "s"(xshi)',
               '[rnlo] "s"(rnlo)', '[rnhi] "s"(rnhi)', '[rnstride] "s"(rnstride)',
               '[row0] "s"(row0)', '[rowstride] "s"(rowstride)', '[ntiles] "s"(ntiles)',
               '[pb] "s"(pb)', '[qbytes] "s"(qbytes)', '[nb] "s"(nb)', '[qcur0] "s"(qcur0)', '[qc1] "s"(qc1)',
               '[wtype] "s"(wtype)', '[wgbu] "s"(wgbu)', '[wgbr] "s"(wgbr)', '[wgbq] "s"(wgbq)', '[wgcp] "s"(wgcp)',
               '[ovfb] "s"(ovfb)']
    if (space == "l2" and not L2C) or (i8 and space == "cosine"):
        ops_in.append('[k1] "s"(k1)')
    if L2C:
        ops_in += ['[sqc] "s"(sqc)', '[kec] "s"(kec)', '[krc] "s"(krc)']
    if dma:
        ops_in.append('[wave2k] "s"(wave2k)')
    if L2E:
        ops_in.append('[eo0in] "s"(eo0)')
    if stag:
        ops_in += ['[xrot] "s"(xrot)', '[pbrot] "s"(pbrot)', '[pb2] "s"(pb2)', '[hc] "s"(hc)']
    clobbers = ['"memory"', '"scc"', '"vcc"'] + (['"m0"'] if dma else []) + [f'"s{i}"' for i in range(60, 80)] + [f'"s{i}"' for i in range(80, 94)] + (
        [f'"v{i}"' for i in range(VA_BASE, VA_BASE + 64 * MT)] if va else [f'"a{i}"' for i in range(64 * MT)])

    text = ["// GENERATED by tools/gen_scan_asm.py -- do not edit.",
            f"// filter scan body: space {space}, NW={NW} waves x {16 * MT} rows, ring R={R} k-steps, B fragments read {QD} ahead"
            f"{', X loads non-temporal' if nt else ''}{', progress-based wave priority' if prio else ''}{', Q staged by LDS-DMA' if dma else ''}"
            f"{', late waves staggered by half a tile' if stag else ''}{', int8 shadow (v_mfma_i32_16x16x64_i8)' if i8 else ''}"
            f"{', accumulators in ArchVGPRs' if va else ''}.",
            "asm volatile("]
    for ln in out:
        text.append(f'    "{ln}\\n\\t"')
    text.append("    : " + ",\n      ".join(ops_out))
    text.append("    : " + ",\n      ".join(ops_in))
    text.append("    : " + ", ".join(clobbers) + ");")
    return "\n".join(text) + "\n"


# ---------------------------------------------------------------------------------------------------------------------
# What gets generated.  Every entry: (file name, dispatch condition of filter_scan_asm_kernel's template arguments, thunk
# that returns the text, kind) with kind "default" (the library as shipped), "ab" (`make AB=1`: tuning variants for
# tools/scan_ab.py and the `ab`-marked tests) or "diag" (`make DIAG=1`: timing diagnostics, wrong results by design).
def cond(space, nw, r, nt, qd, prio, mt, dma, stag):
    return (f"SPACE == {SPACES[space]} && NW == {nw} && R == {r} && NT == {'true' if nt else 'false'} && QD == {qd}"
            f" && PRIO == {'true' if prio else 'false'} && MT == {mt} && DMA == {'true' if dma else 'false'}"
            f" && STAG == {'true' if stag else 'false'}")


def inc_name(space, nw, r, nt, qd, prio, mt, dma, stag=False):
    return (f"scan_asm_{space}_nw{nw}_r{r}{'_nt' if nt else ''}{'_qd%d' % qd if qd != 4 else ''}"
            f"{'_pr' if prio else ''}{'_mt4' if mt == 4 else ''}{'_dma' if dma else ''}{'_stag' if stag else ''}.inc")


def default_i8_body(space):
    """The int8 body the library runs by default for `space` (ArchVGPR accumulators, wave priorities): scan_asm_<space>_i8_va.inc;
    l2: the 16-tile l2c body, scan_asm_l2_i8_va_c_nqt16.inc."""
    DBG.clear()
    if space == "l2":
        return generate("l2", 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True, nqt=16, l2c=True)
    # round 3: straight-line append routine (all spaces) and early-out hit stubs (cosine): gen_slow_fast / gen_hit_stubs
    return generate(space, 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True)


def with_dbg(knobs, *args, **kw):
    DBG.clear()
    DBG.update(knobs)
    try:
        return generate(*args, **kw)
    finally:
        DBG.clear()


def entries():
    E = []
    # ---- default library
    for sp in ("cosine", "ip"):   # the int8 bodies with ArchVGPR accumulators (wave priorities on): QD slot 211.  (l2: 243-245 below)
        E.append((f"scan_asm_{sp}_i8_va.inc", cond(sp, 8, 4, True, 211, True, 2, True, False), (lambda sp=sp: default_i8_body(sp)), "default"))
    # 241 / 242 (round 4): the default body computing 8 / 4 of the 16 query tiles: passes of <= 128 / <= 64 queries
    for sp in ("cosine", "ip"):
        for code, nqt in ((241, 8), (242, 4)):
            E.append((f"scan_asm_{sp}_i8_va_nqt{nqt}.inc", cond(sp, 8, 4, True, code, True, 2, True, False),
                      (lambda sp=sp, nqt=nqt: with_dbg((), sp, 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True, nqt=nqt)), "default"))
    # 243 / 244 / 245 (round 4): l2 -- admission test folded into the last k-step by per-row integer offsets through the first
    # k-step's C operand, one query scale per pass, per-row-group errors (cosine's one-constant test); 16 / 8 / 4 query tiles.
    # (The l2 pairs carry the group's scale / error in their x-slots: no other int8 body reads them -- the round-3 l2 bodies,
    # serial test or AccVGPR accumulators, are gone; their A/Bs are in profiles/r04/scan_ab_l2_*.txt.)
    for code, nqt in ((243, 16), (244, 8), (245, 4)):
        E.append((f"scan_asm_l2_i8_va_c_nqt{nqt}.inc", cond("l2", 8, 4, True, code, True, 2, True, False),
                  (lambda nqt=nqt: with_dbg((), "l2", 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True, nqt=nqt, l2c=True)), "default"))
    # 237: round 2's default body (append routine with eight skipped row blocks, stubs without the early out): the A/B reference
    E.append(("scan_asm_cosine_i8_va_r2.inc", cond("cosine", 8, 4, True, 237, True, 2, True, False),
              lambda: with_dbg((), "cosine", 4, 4, 8, True, True, 2, True, False, True, True), "default"))
    for sp in SPACES:   # bf16 bodies of an index that keeps a bf16 shadow: 8 waves, LDS-DMA staging, ring of 4 (2: odd chunk counts)
        for r in (4, 2):
            c = (sp, 8, r, True, 4, False, 2, True, False)
            E.append((inc_name(*c), cond(*c), (lambda c=c: with_dbg((), c[0], c[2], c[4], c[1], c[3], c[5], c[6], c[7], c[8])), "default"))
    # ---- make AB=1: the other geometries of the bf16 body
    ab_cfgs = [(sp, nw, r, True, 4, False, 2, False, False) for sp in SPACES for nw in (4, 8) for r in (2, 4)] + [
        ("cosine", 4, 4, False, 4, False, 2, False, False), ("cosine", 8, 4, False, 4, False, 2, False, False),
        ("cosine", 8, 4, True, 4, True, 2, False, False)] + [
        (sp, 4, r, True, 4, False, 4, False, False) for sp in SPACES for r in (2, 4)] + [
        (sp, 8, 4, True, 4, False, 2, True, True) for sp in SPACES]
    for c in ab_cfgs:
        E.append((inc_name(*c), cond(*c), (lambda c=c: with_dbg((), c[0], c[2], c[4], c[1], c[3], c[5], c[6], c[7], c[8])), "ab"))
    # int8 bodies with AccVGPR accumulators and the serial admission phase (round 1; QD slot 208), with / without wave priorities
    for sp in ("cosine", "ip"):
        for pr in (False, True):
            E.append((f"scan_asm_{sp}_i8{'_pr' if pr else ''}.inc", cond(sp, 8, 4, True, 208, pr, 2, True, False),
                      (lambda sp=sp, pr=pr: with_dbg((), sp, 4, 4, 8, True, pr, 2, True, False, True)), "ab"))
    G = lambda *a, **k: (lambda: with_dbg((), *a, **k))
    cos = "cosine"
    # tuning variants of the folded cosine body (DESIGN / profiles/r02, r03)
    E += [
        ("scan_asm_cosine_i8_va_r6.inc", cond(cos, 8, 6, True, 214, True, 2, True, False), G(cos, 6, 4, 8, True, True, 2, True, False, True, True), "ab"),
        ("scan_asm_cosine_i8_va_qd8.inc", cond(cos, 8, 4, True, 215, True, 2, True, False), G(cos, 4, 8, 8, True, True, 2, True, False, True, True), "ab"),
        ("scan_asm_cosine_i8_va_nopr.inc", cond(cos, 8, 4, True, 216, False, 2, True, False), G(cos, 4, 4, 8, True, False, 2, True, False, True, True), "ab"),
        ("scan_asm_cosine_i8_va_nw4.inc", cond(cos, 4, 4, True, 217, False, 2, True, False), G(cos, 4, 4, 4, True, False, 2, True, False, True, True), "ab"),
        ("scan_asm_cosine_i8_va_nw4_pr.inc", cond(cos, 4, 4, True, 218, True, 2, True, False), G(cos, 4, 4, 4, True, True, 2, True, False, True, True), "ab"),
        ("scan_asm_cosine_i8_va_q3d.inc", cond(cos, 8, 6, True, 229, True, 2, True, False), G(cos, 6, 4, 8, True, True, 2, True, False, True, True, False, None, 0, True), "ab"),
        ("scan_asm_cosine_i8_va_r6b3.inc", cond(cos, 8, 6, True, 228, True, 2, True, False), G(cos, 6, 4, 8, True, True, 2, True, False, True, True, False, None, 3), "ab"),
        ("scan_asm_cosine_i8_va_stag.inc", cond(cos, 8, 4, True, 222, True, 2, True, True), G(cos, 4, 4, 8, True, True, 2, True, True, True, True), "ab"),
        ("scan_asm_cosine_i8_va_p0.inc", cond(cos, 8, 4, True, 220, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, False, 0), "ab"),
        ("scan_asm_cosine_i8_va_p4.inc", cond(cos, 8, 4, True, 221, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, False, 1), "ab"),
        ("scan_asm_cosine_i8_mt4.inc", cond(cos, 4, 4, True, 230, False, 4, True, False), G(cos, 4, 4, 4, True, False, 4, True, False, True), "ab"),
        ("scan_asm_cosine_i8_va_q4.inc", cond(cos, 8, 4, True, 219, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, True), "ab"),
        ("scan_asm_cosine_i8_va_qa.inc", cond(cos, 8, 4, True, 231, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, qa=True), "ab"),
        ("scan_asm_cosine_i8_va_eo.inc", cond(cos, 8, 4, True, 232, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, eo=True), "ab"),
        ("scan_asm_cosine_i8_va_qa_eo.inc", cond(cos, 8, 4, True, 233, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, qa=True, eo=True), "ab"),
        ("scan_asm_cosine_i8_va_fs.inc", cond(cos, 8, 4, True, 235, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, fs=True), "ab"),
        ("scan_asm_cosine_i8_va_eo_fs.inc", cond(cos, 8, 4, True, 236, True, 2, True, False), G(cos, 4, 4, 8, True, True, 2, True, False, True, True, eo=True, fs=True), "ab"),
    ]
    # ---- make DIAG=1 (implies AB): timing diagnostics.  bf16 body (NW=8, R=4, nt, register staging): QD slot = the knob
    D = lambda knobs, *a, **k: (lambda: with_dbg(knobs, *a, **k))
    for code, knobs in ((101, {"nolds"}), (102, {"nox"}), (103, {"nolds", "nox"}), (104, {"nolds", "nox", "nobar"}),
                        (107, {"nohit"}), (108, {"stamp"}), (109, {"noadm"})):
        E.append((f"scan_asm_diag{code}.inc", cond(cos, 8, 4, True, code, False, 2, False, False), D(knobs, cos, 4, 4, 8, True, False), "diag"))
    # int8 body, AccVGPR accumulators: 209 without its admission test, 210 the test's arithmetic without the accumulator reads
    E.append(("scan_asm_diag209.inc", cond(cos, 8, 4, True, 209, True, 2, True, False), D({"noadm"}, cos, 4, 4, 8, True, True, 2, True, False, True), "diag"))
    E.append(("scan_asm_diag210.inc", cond(cos, 8, 4, True, 210, True, 2, True, False), D({"noread", "nohit"}, cos, 4, 4, 8, True, True, 2, True, False, True), "diag"))
    # folded body: 212 no hit ever taken, 213 no admission test, 223..227 without its MFMAs / X loads / B reads / Q staging
    for code, knobs in ((212, {"nohit"}), (213, {"noadm"}), (223, {"nomfma", "nohit"}), (224, {"nox", "nohit"}), (225, {"nolds", "nohit"}),
                        (226, {"nox", "nolds", "nohit"}), (227, {"noq", "nohit"})):
        E.append((f"scan_asm_diag{code}.inc", cond(cos, 8, 4, True, code, True, 2, True, False), D(knobs, cos, 4, 4, 8, True, True, 2, True, False, True, True), "diag"))
    # 234: the default body; the C++ wrapper stamps its phases around it (correct results).  Stamps INSIDE the statement were
    # tried: two more live SGPR outputs do not fit (the "s" inputs then come out as VGPRs and the assembler refuses them), two
    # more VGPR outputs make hipcc's register allocator hang (> 40 minutes, killed)
    E.append(("scan_asm_diag234.inc", cond(cos, 8, 4, True, 234, True, 2, True, False), lambda: default_i8_body(cos), "diag"))
    return E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default=str(Path(__file__).resolve().parents[1] / "mlvectordb_amd" / "csrc"))
    ap.add_argument("--list", action="store_true", help="print the generated file names and exit")
    ap.add_argument("--ab", action="store_true", help="also the tuning variants (make AB=1)")
    ap.add_argument("--diag", action="store_true", help="also the timing diagnostics (make DIAG=1; implies --ab)")
    args = ap.parse_args()
    kinds = {"default"} | ({"ab"} if args.ab or args.diag else set()) | ({"diag"} if args.diag else set())
    E = entries()
    names = [e[0] for e in E if e[3] in kinds] + ["scan_asm_dispatch.inc", "scan_asm_consts.inc"]
    if args.list:
        print(" ".join(names))
        return
    out = Path(args.outdir)
    for name, _, thunk, kind in E:
        if kind in kinds:
            (out / name).write_text(thunk())
    # the dispatch names every body; the AB / DIAG ones sit behind the build's macros, so their files need not exist otherwise
    disp = ["// GENERATED by tools/gen_scan_asm.py -- do not edit.  Body of filter_scan_asm_kernel<SPACE, R, NW, NT, QD, PRIO, MT, DMA, STAG>."]
    first = True
    for kind, guard in (("default", None), ("ab", "MLVDB_AB"), ("diag", "MLVDB_SCAN_DIAGNOSTICS")):
        if guard:
            disp.append(f"#ifdef {guard}")
        for name, c, _, k in E:
            if k != kind:
                continue
            disp.append(("if" if first else "} else if") + f" constexpr ({c}) {{")
            disp.append(f'#include "{name}"')
            first = False
        if guard:
            disp.append("#endif")
    disp.append("} else {")
    disp.append('    static_assert(SPACE < 0, "configuration not generated (or not in this build: make AB=1 / DIAG=1): see entries() in tools/gen_scan_asm.py");')
    disp.append("}")
    (out / "scan_asm_dispatch.inc").write_text("\n".join(disp) + "\n")
    (out / "scan_asm_consts.inc").write_text(
        "// GENERATED by tools/gen_scan_asm.py -- do not edit.\n"
        f"constexpr int kAsmWgCap = {WG_CAP};\n"
        f"constexpr int kAsmStageCapNw4 = {lds_stage_cap(4, 2, 2)};  // entries per wave staged in LDS\n"
        f"constexpr int kAsmStageCapNw8 = {lds_stage_cap(8, 2, 2)};\n"
        f"constexpr int kAsmStageCapNw4Mt4 = {lds_stage_cap(4, 4, 2)};  // one wave per SIMD, 64 rows per wave\n"
        f"constexpr int kAsmStageCapNw8Q4 = {lds_stage_cap(8, 2, 4)};  // four Q buffers (QD slot 219)\n")
    print("wrote", len(names), "files to", args.outdir)


if __name__ == "__main__":
    main()
