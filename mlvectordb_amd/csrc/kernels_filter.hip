// Filter path for large query batches: a bf16-MFMA bound filter over the fp32 corpus followed
// by exact fp64 rescoring of the few surviving rows.
//
// Why: with 256 queries per pass the scan needs 2*N*d*256 flop per 4*N*d corpus bytes; in
// fp32 (VALU or f32-input MFMA, 157 TF) that is compute-bound at ~15 % of the HBM roof, on
// bf16 MFMA it fits under the HBM stream.  bf16 scores are not exact, so they are used only
// as BOUNDS:  for every (query, row) the kernel computes u >= s, an upper bound of the true
// score s (higher = nearer), from the bf16 dot product a and a rigorous rounding bound
//       |a - <q^, x>| <= E1q * |x|,  E1q = eq + (1 + 2^-8) * rmax + ld*2^-22
// with eq = |q^ - q^_b| the MEASURED rounding error of this query's bf16 image (query_prep_kernel)
// and rmax = max over rows of |x - x_b| / |x| (row_norms_kernel), both rounded up:
//   <q^,x> - <q^_b,x_b> = <q^ - q^_b, x> + <q^_b, x - x_b>,  |.| <= eq |x| + (1 + eq) rmax |x|
// (Cauchy-Schwarz, eq <= 2^-8), plus the fp32 accumulation slack of the MFMA chain.  (The worst
// case of two bf16 roundings is 2^-7; measured errors are ~0.4 * 2^-8 each, and a 2.3x tighter
// bound means ~4x fewer candidates per query.)
// A row is discarded only if u < thr[q], where thr[q] is a proven LOWER bound of the k-th
// best true score (k rows with lower bound l = u - 2*eps >= thr exist, or k exactly scored
// seed rows do).  Everything else is appended to the query's candidate list and rescored in
// fp64 by the same arithmetic as the exact scan, so the answer is the exact one.  A list
// that overflows (adversarial near-ties) flags its query for the exact scan instead.
//
// Score units per space (q^ = q/(|q|+1e-30)):
//   cosine  s = <q^,x>/(|x|+1e-30)       = 1 - dist
//   ip      s = <q^,x>                   = (1 - dist)/|q|
//   l2      s = 2|q|<q^,x> - |x|^2       = |q|^2 - dist
#include <cstdlib>
#include <type_traits>

#include "bound_common.h"
#include "internal.h"
#include "scan_common.h"

namespace mlvdb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: wave-uniform descriptor + scalar byte offset, one VGPR of lane offset.
// Keeps the streaming loads' address arithmetic on the scalar unit (no 64-bit VGPR adds).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// the same, non-temporal (aux bit 1 = nt on gfx950): the corpus stream is read once per pass
__device__ __forceinline__ float4 buf_load_f4_nt(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
__device__ __forceinline__ uint4 buf_load_u4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int kChunkVec = kFilterQueries * kFilterChunkK * 2 / 16;  // uint4 per Q chunk (2048)
constexpr float kSlack = 1.9073486328125e-06f;                      // 2^-19
constexpr int kSeedTileRows = 128;                                  // tile of the 2-panel geometry
static_assert(kFilterTile % 192 == 0 && kFilterTile % 128 == 0, "kFilterTile");

bool filter_supported(int32_t ld) { return ld >= kFilterChunkK && (ld % kFilterChunkK) == 0; }
size_t filter_qimg_bytes(int32_t ld) { return (size_t)kFilterQueries * ld * 2; }


// ------------------------------------------------------------------ query image
// Qimg[kc][n][ks][lane][j] = bf16(q^[16n + (lane&15)][64kc + 16(2ks + (j>>2)) + 4(lane>>4) + (j&3)])
// i.e. exactly the B-operand fragments of v_mfma_f32_16x16x32_bf16 for the k-order in which
// the scan kernel receives its A fragments from the panel layout (layout.h).
__global__ __launch_bounds__(256) void filter_prep_kernel(const FilterArgs a) {
    const int nkc = a.ld / kFilterChunkK;
    __bf16* img = reinterpret_cast<__bf16*>(a.qimg);
    const int64_t total = (int64_t)nkc * kChunkVec * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = idx & 7;
        const int lane = (idx >> 3) & 63;
        const int ks = (idx >> 9) & 1;
        const int n = (idx >> 10) & 15;
        const int kc = (int)(idx >> 14);
        const int q = 16 * n + (lane & 15);
        // k order of the A fragments: fp32 panels give lane g the columns {4g..4g+3} of two 16-col groups,
        // the bf16 shadow gives it 8 consecutive columns
        const int col = a.Xb ? 64 * kc + 32 * ks + 8 * (lane >> 4) + j
                             : 64 * kc + 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
        float v = 0.f;
        if (q < a.nq) {
            const double aux = a.qaux[q];
            const double inv = a.space == kSpaceCosine ? aux : 1.0 / (aux + 1e-30);
            v = a.Qpad[(int64_t)q * a.ld + col] * (float)inv;
        }
        img[idx] = (__bf16)v;
    }
    if (blockIdx.x == 0) {
        const int q = threadIdx.x;  // 256 threads = kFilterQueries
        const double nrm = q < a.nq ? (a.space == kSpaceCosine ? 0.0 : a.qaux[q]) : 0.0;
        a.qscale[q] = a.space == kSpaceL2 ? (float)(2.0 * nrm) : 1.0f;
        // per-query error term of the bound (see the header): E1q, plus the slack of the epilogue arithmetic
        const double e1q = (q < a.nq ? (double)a.qerr[q] : 0.0) + 1.00390625 * (double)*a.row_err + (double)a.ld * 2.384185791015625e-07;
        float ke = (float)(e1q * 1.000001) + (a.space == kSpaceCosine ? 2.0f : 1.0f) * kSlack;
        a.ke[q] = __uint_as_float(__float_as_uint(ke) + 1u);
        a.thr[q] = q < a.nq ? -3.0e38f : 3.4e38f;  // padded queries never admit anything
        a.cnt[q] = 0;
        a.overflow[q] = 0;
        if (q == 0 && a.sqmin) {  // what filter_prep8_kernel's atomicMin / atomicMax start from (it runs after this kernel)
            a.sqmin[0] = 0x7f7f7f7fu;  // 3.39e38: above every query scale
            a.sqmin[1] = 0u;           // largest query error so far
        }
    }
}

// thr[q] from the exact distance of the k-th nearest seed row
__global__ __launch_bounds__(256) void filter_seed_thr_kernel(const FilterArgs a, const double* seed_d64, int32_t k) {
    const int q = threadIdx.x;
    if (q >= a.nq || a.overflow[q]) return;
    const double dk = seed_d64[(int64_t)q * k + (k - 1)];
    if (!(dk < 1.0e300)) return;  // fewer than k seeds (inf) or NaN
    const double aux = a.qaux[q];
    double s;
    if (a.space == kSpaceCosine) {
        s = 1.0 - dk;
    } else if (a.space == kSpaceIp) {
        if (!(aux > 0.0)) return;
        s = (1.0 - dk) / aux;
    } else {
        s = aux * aux - dk;
    }
    const double mag = a.space == kSpaceL2 ? aux * aux + __builtin_fabs(dk) : __builtin_fabs(s) + 1.0;
    a.thr[q] = float_below(s - 1e-9 * mag);
}

// thr[q] for a range query: every row with dist <= radius has s >= s(radius)
__global__ __launch_bounds__(256) void filter_range_thr_kernel(const FilterArgs a, float radius) {
    const int q = threadIdx.x;
    if (q >= a.nq || a.overflow[q]) return;  // (a query the pass took off the filter keeps its +inf threshold: filter_l2_offsets_kernel)
    const double aux = a.qaux[q];
    const double r = (double)radius;
    double s;
    if (a.space == kSpaceCosine) {
        s = 1.0 - r;
    } else if (a.space == kSpaceIp) {
        if (!(aux > 0.0)) return;  // |q| = 0: every distance is 1; keep thr = -max (admit all)
        s = (1.0 - r) / aux;
    } else {
        s = aux * aux - r;
    }
    const double mag = a.space == kSpaceL2 ? aux * aux + __builtin_fabs(r) : __builtin_fabs(s) + 1.0;
    a.thr[q] = float_below(s - 1e-9 * mag);
}

// ------------------------------------------------------------------ the scan
// Tile finished: bounds, admission test, rare appends.  acc[m][n] = bf16 dot products of this
// lane's rows (16 m + 4 g + i, i = register component) with query 16 n + c16; rnv[m] = |x| of the rows;
// row0 = first of this lane's rows; dump = this lane's column of a [4*kMT][64] LDS scratch per wave.
template <int SPACE, int kMT, bool DENSE, int NQT = 16, bool I8 = false>
__device__ __forceinline__ void scan_epilogue(const FilterArgs& a, const f32x4 (&acc)[kMT][NQT], const float4 (&rnv)[kMT],
                                              const float4 (&rbv)[kMT], const int32_t row0, const int32_t base_row,
                                              const float* thr_l, const float* sq_l, const float* ke_l, float* dump,
                                              const int c16, const int qbase = 0) {
    // qbase: first query of the 16*NQT this call covers (thr_l / sq_l / ke_l are already offset by it; the lists are not)
    // ke = the query's error term (filter_prep_kernel).  Per-row constants:
    //   cosine  u = a*p0 + ke             (p0 = 1/(|x|+1e-30))
    //   ip      u = a + ke*p0             (p0 = |x|)
    //   l2      u = sq*(a + ke*p0) + p1   (p1 = -|x|^2 (1-slack))
    float p0[kMT][4], p1[kMT][4];
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
        const float nr[4] = {rnv[m].x, rnv[m].y, rnv[m].z, rnv[m].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (I8) {  // int8 shadow: acc = float(integer dot).  cosine: rnv = sx/(|x|+1e-30), rbv = the row's error x K;
                       // l2 / ip: rnv = the row's scale sx, rbv = |x| (p1 of l2 is derived from it below)
                const float nb[4] = {rbv[m].x, rbv[m].y, rbv[m].z, rbv[m].w};
                p0[m][i] = nr[i];
                p1[m][i] = nb[i];
            } else if (SPACE == kSpaceCosine) {
                p0[m][i] = 1.0f / (nr[i] + 1e-30f);
                p1[m][i] = 0.f;
            } else {
                p0[m][i] = nr[i];
                p1[m][i] = SPACE == kSpaceL2 ? -(nr[i] * nr[i]) * (1.0f - kSlack) : 0.f;
            }
        }
    }
    auto bound = [&](int m, int i, int n, float sq, float ke) __attribute__((always_inline)) {
        const float av = acc[m][n][i];
        if (I8 && SPACE == kSpaceCosine) return __builtin_fmaf(__builtin_fmaf(av, p0[m][i], p1[m][i]), sq, ke);  // sq8 (w + b K) + ke8
        if (I8) {
            // l2 / ip on the int8 shadow, the assembly body's arithmetic (filter_scan_asm_kernel's preamble): w = float(I) sx;
            // ip  u = sq8 (w + ke' |x|),  l2  u = sq' (w + ke' |x|) - |x|^2 (1 - slack)  with ke' = ke / sq8 rounded up and
            // sq = sq8 (ip) or 2 |q| sq8 (l2), both prepared per query by the kernel
            const float w = av * p0[m][i];
            const float t = __builtin_fmaf(ke, p1[m][i], w);
            if (SPACE == kSpaceIp) return sq * t;
            return __builtin_fmaf(sq, t, -(p1[m][i] * p1[m][i]) * (1.0f - kSlack));
        }
        if (SPACE == kSpaceCosine) return __builtin_fmaf(av, p0[m][i], ke);
        if (SPACE == kSpaceIp) return __builtin_fmaf(ke, p0[m][i], av);
        return __builtin_fmaf(sq, __builtin_fmaf(ke, p0[m][i], av), p1[m][i]);
    };
    if (DENSE) {
        // seeding pass: every (query,row) bound of these tiles goes straight into the candidate
        // lists, slot = row - first row of the pass (the caller sets cnt and runs the update kernel)
#pragma unroll
        for (int n = 0; n < NQT; ++n) {
            const float sq = (SPACE == kSpaceL2 || I8) ? sq_l[16 * n + c16] : 1.0f;
            const float ke = ke_l[16 * n + c16];
            CandEntry* dst = a.cand + (int64_t)(qbase + 16 * n + c16) * a.cand_cap + (row0 - base_row);
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    CandEntry e;
                    e.u = bound(m, i, n, sq, ke);
                    e.row = row0 + 16 * m + i;
                    dst[16 * m + i] = e;
                }
        }
        return;
    }
    // pass 1: quick reject per query tile on the maximum bound (NaN bounds of tombstoned rows are
    // ignored by v_max); the per-row mask and one slot reservation (atomic) per lane only where it
    // is needed.  The atomics' results are not touched before pass 2: all of them are in flight.
    uint32_t packed[NQT];  // bits 0..11 hit mask, bits 12.. first reserved slot (clamped)
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
        __builtin_amdgcn_sched_barrier(0);  // keep only one query tile's scores live at a time
        const float thr = thr_l[16 * n + c16];
        const float sq = (SPACE == kSpaceL2 || I8) ? sq_l[16 * n + c16] : 1.0f;
        const float ke = ke_l[16 * n + c16];
        float mx = -3.4e38f;
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = __builtin_fmaxf(mx, bound(m, i, n, sq, ke));
        uint32_t mask = 0, slot = 0;
        if (__ballot(mx >= thr)) {
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) mask |= bound(m, i, n, sq, ke) >= thr ? 1u << (4 * m + i) : 0u;
            if (mask) slot = atomicAdd(&a.cnt[qbase + 16 * n + c16], (uint32_t)__popc(mask));
        }
        packed[n] = mask | (min(slot, (uint32_t)a.cand_cap) << 12);
    }
    // pass 2 (rare): write the admitted (bound, row) pairs into the reserved slots
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
        if (__ballot((packed[n] & 0xfffu) != 0)) {
            const float sq = (SPACE == kSpaceL2 || I8) ? sq_l[16 * n + c16] : 1.0f;
            const float ke = ke_l[16 * n + c16];
#pragma unroll
            for (int m = 0; m < kMT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) dump[(4 * m + i) * 64] = bound(m, i, n, sq, ke);
            const int q = qbase + 16 * n + c16;
            uint32_t mask = packed[n] & 0xfffu;
            uint32_t slot = packed[n] >> 12;
            while (mask) {
                const int j = __builtin_ctz(mask);
                mask &= mask - 1;
                if (slot < (uint32_t)a.cand_cap) {
                    CandEntry e;
                    e.u = dump[j * 64];
                    e.row = row0 + 16 * (j >> 2) + (j & 3);
                    a.cand[(int64_t)q * a.cand_cap + slot] = e;
                } else {
                    a.overflow[q] = 1u;
                }
                ++slot;
            }
        }
    }
}

// The compiler-scheduled scan (hipcc builtins).  Serves corpora without bf16 shadow (XB = false: fp32
// rows converted in registers), the dense seeding pass (DENSE) and the A/B reference of the assembly
// body below, which has the same geometry and data flow per tile:
// one workgroup = 4 waves = one 128-row tile against all 256 queries, two workgroups per CU; wave w owns
// panels 2w, 2w+1, i.e. a 32 x 256 block of scores in 128 accumulator registers.
//   X   HBM -> registers: each wave load is one 1 KiB burst of one panel (layout.h) and is already
//       the MFMA A fragment; nobody else needs those rows, so no LDS.  Two register buffers of one
//       32-column k-step each rotate statically (the step loop is unrolled twice: a run-time rotation
//       makes hipcc copy registers that are targets of in-flight loads, which drains the prefetch).
//   Q   the bf16 query image (L2 resident) goes registers -> LDS in 64-column chunks, double
//       buffered, one barrier per chunk, and is read back as ready-made B fragments
//       (ds_read_b128, lane-linear, conflict free).  Each B fragment feeds 2 MFMAs.
// All loads are ordinary loads on purpose: hipcc then tracks them with counted s_waitcnt.
template <int SPACE, bool XB, bool DENSE>
__global__ __launch_bounds__(256, 2) void filter_scan_kernel(const FilterArgs a, const int64_t tile_begin,
                                                             const int64_t tile_end) {
    constexpr int R = 2, QD = 2, kMT = 2, NW = 4;  // ring depth, B read-ahead, panels per wave, waves (measured best for hipcc's schedule)
    constexpr int U = R;                          // steps per unrolled body
    constexpr int kThreads = NW * 64;
    constexpr int kQPer = 1024 / kThreads;                 // uint4 of a Q half-chunk moved per thread
    constexpr int kFilterTileRows = NW * 16 * kMT;  // rows per workgroup tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* qlds = reinterpret_cast<uint4*>(smem);                                  // [2][kChunkVec]
    float* thr_l = reinterpret_cast<float*>(smem + 2 * kChunkVec * sizeof(uint4));  // [256]
    float* sq_l = thr_l + kFilterQueries;                                           // [256]
    float* ke_l = sq_l + kFilterQueries;                                            // [256]
    float* hit_l = ke_l + kFilterQueries;                                           // [NW waves][4*kMT][64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const int ld = a.ld;
    const int nkc = ld / kFilterChunkK;
    if (threadIdx.x < kFilterQueries) {
        thr_l[threadIdx.x] = a.thr[threadIdx.x];
        sq_l[threadIdx.x] = a.qscale[threadIdx.x];
        ke_l[threadIdx.x] = a.ke[threadIdx.x];
    }

    const int64_t ntiles = tile_end - tile_begin;
    const int64_t my_tiles = ntiles > blockIdx.x ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t total_chunks = my_tiles * nkc;
    const int64_t total_steps = total_chunks * 2;
    if (total_steps == 0) return;
    const int steps_per_tile = 2 * nkc;  // a multiple of U by construction (launcher picks R)
    const int64_t panel_stride = (int64_t)kPanelRows * ld;  // floats per panel

    f32x4 acc[kMT][16];
    constexpr int kLoads = XB ? 1 : 2;  // 16-byte loads per panel and 32-column step
    float4 xr[R][kMT][kLoads];  // X k-steps in flight (fp32, or bf16 bits); indexed only by unrolled constants
    uint4 qst[kQPer];    // this thread's share of the next Q half-chunk
    float4 rnv[kMT];     // |x| of this lane's kMT x 4 rows of the current tile

    // X prefetch cursor: a 64-bit wave-uniform base (this wave's 3 panels of the tile being
    // prefetched) plus a 32-bit byte offset of the half-chunk inside the panels.  Everything is
    // advanced incrementally on the scalar unit; no multiplies on the per-step path.
    auto tile_of = [&](int64_t i) __attribute__((always_inline)) { return tile_begin + blockIdx.x + i * gridDim.x; };
    // the only per-lane part of every X address: lane-linear in the bf16 shadow, row r / column quarter g in the fp32 panels
    const uint32_t lane_off16 = XB ? lane * 16 : lane_group_offset(lane) * 4;
    const uint32_t panel_bytes = (uint32_t)(panel_stride * (XB ? 2 : 4));
    const uint32_t wave_bytes = kMT * panel_bytes;         // this wave's panels of one tile
    const uint32_t row_bytes_in_panel = (uint32_t)ld * (XB ? 32 : 64);  // bytes of one panel (groups of 1 KiB)
    const uint64_t tile_stride_bytes = (uint64_t)gridDim.x * (NW * wave_bytes);
    const char* pre_base = reinterpret_cast<const char*>(XB ? a.Xb : (const void*)a.X) +
                           (uint64_t)(tile_begin + blockIdx.x) * (NW * wave_bytes) +
                           (uint64_t)wave * wave_bytes;
    uint32_t pre_soff = 0;
    int64_t pre_tiles_left = my_tiles - 1;
    auto load_x = [&](float4(&xb)[kMT][kLoads]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(pre_base, wave_bytes);
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int kb = 0; kb < kLoads; ++kb)
                xb[m][kb] = buf_load_f4_nt(r, lane_off16 + kb * 1024, pre_soff + m * panel_bytes);
        // advance, saturating at the last half-chunk (the tail re-loads it: harmless, and it keeps
        // every load unconditional -- a conditional load's phi makes hipcc wait for it at once)
        pre_soff += kLoads * 1024;
        if (pre_soff == row_bytes_in_panel) {
            if (pre_tiles_left > 0) {
                --pre_tiles_left;
                pre_soff = 0;
                pre_base += tile_stride_bytes;
            } else {
                pre_soff -= kLoads * 1024;
            }
        }
    };
    const __amdgpu_buffer_rsrc_t q_rsrc = make_rsrc(a.qimg, (uint32_t)(nkc * kChunkVec * sizeof(uint4)));
    const uint32_t tid_off16 = (((threadIdx.x >> 6) * 2) * 64 + (threadIdx.x & 63)) * 16;
    // half `h` of a chunk = its 16 B-fragment pieces with k-step == h; thread t moves uint4 t + kThreads*i
    auto load_q = [&](int kc, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kQPer; ++i)
            qst[i] = buf_load_u4(q_rsrc, tid_off16, (uint32_t)(kc * kChunkVec + ((i * NW) * 2 + h) * 64) * 16);
    };
    auto store_q = [&](int buf, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kQPer; ++i) {
            const int v = threadIdx.x + i * kThreads;
            qlds[buf * kChunkVec + ((v >> 6) * 2 + h) * 64 + (v & 63)] = qst[i];
        }
    };

    // ---- prologue: first half of Q(0) in LDS, second half in registers, X(0..R-1) in flight
    load_q(0, 0);
    store_q(0, 0);
    load_q(0, 1);
    int q_next_kc = nkc > 1 ? 1 : 0;  // chunk whose halves the next two load_q calls fetch
#pragma unroll
    for (int b = 0; b < R; ++b) load_x(xr[b]);

    auto epilogue = [&](const int64_t ti) __attribute__((always_inline)) {
        scan_epilogue<SPACE, kMT, DENSE>(a, acc, rnv, rnv, (int32_t)(tile_of(ti) * kFilterTileRows) + wave * (16 * kMT) + g * 4,
                                         (int32_t)(tile_begin * kFilterTileRows), thr_l, sq_l, ke_l,
                                         hit_l + wave * (4 * kMT * 64) + lane, c16);
    };

    for (int64_t ti = 0; ti < my_tiles; ++ti) {
        const __amdgpu_buffer_rsrc_t rn_rsrc =
            make_rsrc(a.rn + tile_of(ti) * kFilterTileRows + wave * (16 * kMT), 16 * kMT * 4);
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int n = 0; n < 16; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < steps_per_tile; st += U) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                // one step = one 32-column half-chunk; b, h are compile-time after unrolling
                const int b = j % R;
                const int h = j & 1;
                const int64_t s = ti * steps_per_tile + st + j;
                const int64_t c = s >> 1;
                // 1. this k-step's A fragments: bf16 bits as loaded, or fp32 -> bf16 in registers
                bf16x8 xa[kMT];
#pragma unroll
                for (int m = 0; m < kMT; ++m) {
                    if (XB) {
                        xa[m] = __builtin_bit_cast(bf16x8, xr[b][m][0]);
                    } else {
                        const float4 lo = xr[b][m][0], hi = xr[b][m][kLoads - 1];
                        const f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        xa[m] = __builtin_convertvector(v, bf16x8);
                    }
                }
                // 2. Q staging, one half-chunk per step: write the half fetched one step ago, then
                //    (new chunk) barrier, then fetch the next half.  Step 2c completes chunk c.
                if (h == 0) {
                    store_q((int)(c & 1), 1);
                    __syncthreads();  // everyone is done with chunk c-1; chunk c is complete and visible
                    load_q(q_next_kc, 0);
                } else {
                    store_q((int)((c + 1) & 1), 0);
                    load_q(q_next_kc, 1);
                    q_next_kc = q_next_kc + 1 == nkc ? 0 : q_next_kc + 1;
                    if (j == U - 1) {
                        // row norms for the epilogue; reloaded every U steps so the load is unconditional
#pragma unroll
                        for (int m = 0; m < kMT; ++m) rnv[m] = buf_load_f4(rn_rsrc, g * 16, m * 64);
                    }
                }
                // 3. refill the register buffer just consumed with the half-chunk R steps ahead
                load_x(xr[b]);
                // 4. 64 MFMAs: 16 query tiles x 4 row tiles, k-step h of the chunk
                const uint4* qb_base = qlds + (c & 1) * kChunkVec + h * 64 + lane;
                uint4 qraw[QD];  // B fragments read QD-1 query tiles ahead of their MFMAs
#pragma unroll
                for (int i = 0; i < QD - 1; ++i) qraw[i] = qb_base[i * 128];
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    if (n + QD - 1 < 16) qraw[(n + QD - 1) % QD] = qb_base[(n + QD - 1) * 128];
                    const bf16x8 qb = __builtin_bit_cast(bf16x8, qraw[n % QD]);
#pragma unroll
                    for (int m = 0; m < kMT; ++m) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[m], qb, acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
        epilogue(ti);
    }
}

// Narrow scan for batches of <= 16*NQT queries: the whole query image of those queries stays in LDS
// for the life of the (persistent) workgroup, so the tile loop has no staging and no barrier; the
// kernel is a pure stream of the bf16 shadow (1 KiB burst per wave load = one MFMA A fragment) and is
// HBM-bound at HALF the bytes of the exact fp32 scan.  Same bounds, admission test, candidate lists
// and rescoring as the 256-query kernels, so the ids are the exact ones here too.
// One workgroup = NW waves = one 32*NW-row tile; wave w owns panels 2w, 2w+1 (NW = 8; 4 for the seeding pass).
template <int SPACE, int NQT, bool DENSE, int R, int NW, bool I8>
__global__ __launch_bounds__(NW * 64) void filter_scan_narrow_kernel(const FilterArgs a, const int64_t tile_begin,
                                                                     const int64_t tile_end) {
    constexpr int kMT = 2;  // panels per wave; R = k-steps in flight per wave, NW = waves
    // I8: the int8 shadow of a cosine index (k-steps of 64 int8 columns, integer accumulators, rp8 row constants)
    constexpr int kFilterTileRows = NW * 16 * kMT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ld = a.ld;
    const int nsteps = I8 ? a.ld8 / 64 : ld / 32;  // a multiple of R (the launcher picks R = 4 or 2); int8: the shadow's own (zero-padded) width
    uint4* qlds = reinterpret_cast<uint4*>(smem);  // [nsteps][NQT][64]
    float* thr_l = reinterpret_cast<float*>(smem + (size_t)nsteps * NQT * 1024);  // [256]
    float* sq_l = thr_l + kFilterQueries;
    float* ke_l = sq_l + kFilterQueries;
    float* hit_l = ke_l + kFilterQueries;  // [NW waves][4*kMT][64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const int64_t ntiles = tile_end - tile_begin;
    const int64_t my_tiles = ntiles > blockIdx.x ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) return;
    // blockIdx.y (dense seeding pass of a batch of more than 16*NQT queries): which group of 16*NQT queries this workgroup serves
    const int qbase = blockIdx.y * (16 * NQT);
    if (threadIdx.x < kFilterQueries) {
        thr_l[threadIdx.x] = a.thr[threadIdx.x];
        float sqv = a.qscale[threadIdx.x], kev = a.ke[threadIdx.x];
        if (I8) {
            const float sq8 = a.sq8[threadIdx.x];
            if (SPACE == kSpaceCosine) {
                sqv = sq8;
                kev = a.ke8[threadIdx.x];
            } else {  // (scan_epilogue: u = sq (w + ke' |x|) [+ p1])
                kev = float_above((double)kev / (double)sq8);
                sqv = SPACE == kSpaceL2 ? sqv * sq8 : sq8;
            }
        }
        sq_l[threadIdx.x] = sqv;
        ke_l[threadIdx.x] = kev;
    }
    // query image (filter_prep_kernel's [kc][n][ks][lane] order) -> LDS [2kc+ks][n < NQT][lane]
    {
        const uint4* qimg = reinterpret_cast<const uint4*>(I8 ? a.qimg8 : a.qimg);
        for (int v = threadIdx.x; v < nsteps * NQT * 64; v += NW * 64) {
            const int l = v & 63, n = (v >> 6) % NQT, st = (v >> 6) / NQT;
            qlds[v] = qimg[(((st >> 1) * 16 + (qbase >> 4) + n) * 2 + (st & 1)) * 64 + l];
        }
    }
    __syncthreads();

    typename std::conditional<I8, i32x4, f32x4>::type acc[kMT][NQT];
    float4 xr[R][kMT];
    float4 rnv[kMT], rbv[kMT];
    const float kq = I8 ? (SPACE == kSpaceCosine ? a.ke8[kFilterQueries] : 1.0f) : 0.f;  // cosine: the factor of the rows' own errors
    const uint32_t panel_bytes = I8 ? (uint32_t)a.ld8 * 16 : (uint32_t)ld * 32;  // 16 rows of int8 / bf16
    const uint32_t wave_bytes = kMT * panel_bytes;
    const uint64_t tile_stride_bytes = (uint64_t)gridDim.x * (NW * wave_bytes);
    const char* pre_base = reinterpret_cast<const char*>(I8 ? a.X8 : a.Xb) + (uint64_t)(tile_begin + blockIdx.x) * (NW * wave_bytes) +
                           (uint64_t)wave * wave_bytes;
    uint32_t pre_soff = 0;
    int64_t pre_tiles_left = my_tiles - 1;
    auto load_x = [&](float4(&xb)[kMT]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(pre_base, wave_bytes);
#pragma unroll
        for (int m = 0; m < kMT; ++m) xb[m] = buf_load_f4_nt(r, lane * 16, pre_soff + m * panel_bytes);
        pre_soff += 1024;  // saturates at the last k-step of the last tile (every load stays unconditional)
        if (pre_soff == panel_bytes) {
            if (pre_tiles_left > 0) {
                --pre_tiles_left;
                pre_soff = 0;
                pre_base += tile_stride_bytes;
            } else {
                pre_soff -= 1024;
            }
        }
    };
#pragma unroll
    for (int b = 0; b < R; ++b) load_x(xr[b]);

    for (int64_t ti = 0; ti < my_tiles; ++ti) {
        const int64_t tile = tile_begin + blockIdx.x + ti * gridDim.x;
        const __amdgpu_buffer_rsrc_t rn_rsrc = I8 ? make_rsrc(a.rp8 + 2 * (tile * kFilterTileRows + wave * (16 * kMT)), 16 * kMT * 8)
                                                   : make_rsrc(a.rn + tile * kFilterTileRows + wave * (16 * kMT), 16 * kMT * 4);
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            if constexpr (I8) {  // pairs per row: keep the first components (sx/(|x|+1e-30))
                const float4 lo = buf_load_f4(rn_rsrc, g * 32, m * 128), hi = buf_load_f4(rn_rsrc, g * 32 + 16, m * 128);
                rnv[m] = make_float4(lo.x, lo.z, hi.x, hi.z);
                rbv[m] = make_float4(lo.y * kq, lo.w * kq, hi.y * kq, hi.w * kq);  // the rows' own errors x K
                if (SPACE == kSpaceL2 && m == 1) rnv[1] = rnv[0];  // l2 pairs: the second panel's x-slot holds the group's error, the scale is the first's
            } else {
                rnv[m] = buf_load_f4(rn_rsrc, g * 16, m * 64);
            }
        }
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int n = 0; n < NQT; ++n) acc[m][n] = {0, 0, 0, 0};
        for (int st = 0; st < nsteps; st += R) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                float4 xa[kMT];
#pragma unroll
                for (int m = 0; m < kMT; ++m) xa[m] = xr[j][m];
                load_x(xr[j]);
                const uint4* qb = qlds + (size_t)(st + j) * (NQT * 64) + lane;
#pragma unroll
                for (int n = 0; n < NQT; ++n) {
                    const uint4 qf = qb[n * 64];
#pragma unroll
                    for (int m = 0; m < kMT; ++m) {
                        if constexpr (I8)
                            acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, xa[m]), __builtin_bit_cast(i32x4, qf),
                                                                              acc[m][n], 0, 0, 0);
                        else
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[m]),
                                                                                __builtin_bit_cast(bf16x8, qf), acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
        f32x4 accf[kMT][NQT];
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int n = 0; n < NQT; ++n) {
                if constexpr (I8) accf[m][n] = __builtin_convertvector(acc[m][n], f32x4);  // |I| < 2^24 for d <= 1040: exact; beyond: 1 ulp, inside the slack
                else accf[m][n] = acc[m][n];
            }
        scan_epilogue<SPACE, kMT, DENSE, NQT, I8>(a, accf, rnv, rbv, (int32_t)(tile * kFilterTileRows) + wave * (16 * kMT) + g * 4,
                                              (int32_t)(tile_begin * kFilterTileRows), thr_l + qbase, sq_l + qbase, ke_l + qbase,
                                              hit_l + wave * (4 * kMT * 64) + lane, c16, qbase);
    }
}

// ------------------------------------------------------------------ the scan, hand-written for gfx950
// Same geometry, data flow and bounds as filter_scan_kernel<.., kMT = 2, XB = true> above, but the
// whole body -- prologue, persistent tile loop, k-loop, admission test, append path -- is the
// generated assembly of tools/gen_scan_asm.py (scan_asm_<space>_nw<NW>_r<R>.inc; the schedule and
// the reasons are documented there).  This wrapper only computes addresses.
// Requires the bf16 shadow.  Tile = NW*32 rows; the 2*ld/64 k-steps of a tile must be a multiple of R.
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

#include "scan_asm_consts.inc"
static_assert(kAsmWgCap == kWgCap, "tools/gen_scan_asm.py and internal.h disagree");

// The QD template slot doubles as a variant code: 2..8 = B-fragment read-ahead of a bf16 body; 1xx = timing diagnostics of
// the bf16 body; 208..219 = int8 bodies (208 AccVGPR accumulators; 211 ArchVGPR accumulators, cosine: admission folded into the
// last k-step; 214 / 215 / 216 tuning variants of 211: ring of 6, read-ahead 8, no wave priorities; 209 / 210 / 212 / 213
// timing diagnostics).
constexpr bool scan_code_i8(int qd) { return qd >= 208 && qd <= 249; }
// 241 / 242 (round 4): the default int8 body computing only the first 8 / 4 query tiles (passes of <= 128 / <= 64 queries)
constexpr int scan_code_nqt(int qd) { return (qd == 241 || qd == 247 || qd == 244) ? 8 : ((qd == 242 || qd == 248 || qd == 245) ? 4 : 16); }
// 246 / 247 / 248 (round 4): l2 with the folded admission test, per-row integer offsets through the first k-step's C operand
constexpr bool scan_code_l2e(int qd) { return qd >= 243 && qd <= 248; }  // 243-245: l2c (+ one query scale, one error coefficient per pass)
constexpr bool scan_code_l2c(int qd) { return qd >= 243 && qd <= 245; }

// l2c: what the pass's common error coefficient KE = max_q KE_q costs query q, taken back.  The body's bound of row j is
//   u'_j = (per-query bound with q's own KE_q) + (KE - KE_q) N_j   and   N_j >= Nmin  (the smallest row norm the index ever held),
// so testing  u'_j >= thr_q + delta_q  with delta_q = (KE - KE_q) Nmin (rounded down) admits every row the per-query bound
// admits, and  u'_j - delta_q  (rounded up) is still an upper bound of the row's score: the thresholds the body sees are
// raised by delta_q (the wrapper's preamble), the appended bounds lowered by it (its tail).  KE_q: the offsets kernel's formula.
__device__ __forceinline__ float l2c_delta(const FilterArgs& a, int q) {
    const double keq = (double)float_above((double)a.qscale[q] * ((double)a.ke8[q] * 1.000002 + 1.0e-6));
    const double nmin = (double)a.row_err8[1];
    const double d = ((double)a.l2c_out[1] - keq) * nmin * 0.999999;
    if (!(d > 0.0) || !(nmin < 3.0e38)) return 0.f;
    return float_below(d);
}
constexpr int scan_code_qd(int qd) { return qd == 215 ? 8 : (qd > 8 ? 4 : qd); }
constexpr bool scan_code_q4(int qd) { return qd == 219 || qd == 229 || qd == 231 || qd == 233; }  // four Q chunk buffers
constexpr int scan_code_qbufs(int qd) { return scan_code_q4(qd) ? 4 : 2; }
constexpr int scan_code_stage_cap(int qd, int nw, int mt) {
    return scan_code_q4(qd) ? kAsmStageCapNw8Q4 : (mt == 4 ? kAsmStageCapNw4Mt4 : (nw == 8 ? kAsmStageCapNw8 : kAsmStageCapNw4));
}

template <int SPACE, int R, int NW, bool NT, int QD, bool PRIO, int MT, bool DMA, bool STAG>
__global__ __launch_bounds__(NW * 64, MT == 4 ? 1 : 2) void filter_scan_asm_kernel(const FilterArgs a, const int64_t tile_begin,
                                                                                  const int64_t tile_end, const int xcd_mode) {
    constexpr int kThreads = NW * 64;
    constexpr int kQPer = 1024 / kThreads;  // uint4 of a Q half-chunk moved per thread
    constexpr int kWaveRows = 16 * MT;  // MT = 2: two waves per SIMD; MT = 4: one, 64 rows each
    constexpr int kTileRowsV = NW * kWaveRows;
    constexpr int kQBufs = scan_code_qbufs(QD);
    constexpr int kStageCap = scan_code_stage_cap(QD, NW, MT);  // entries a wave stages in LDS
    constexpr bool I8 = scan_code_i8(QD);  // int8 shadow: k-steps of 64 int8 columns, same bytes per step
    // LDS: [2][32 KiB] Q chunks at offset 0, thr[256], qscale[256], ke[256], [NW waves] staging {u[], row[], q[]}
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + kQBufs * kChunkVec * sizeof(uint4));
    float* sq_l = thr_l + kFilterQueries;
    float* ke_l = sq_l + kFilterQueries;
    if (threadIdx.x < NW) a.wgcnt[blockIdx.x * NW + threadIdx.x] = 0;  // workgroups without tiles return below
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1, MLVDB_SCAN_DIAG=234: phase stamps (100 MHz) per wave, in the unused upper half of wgbuf
    // (the pointer is recomputed at every stamp: kept live across the assembly it costs the SGPRs the statement's "s"
    // operands need -- they then come out as VGPRs and the assembler refuses them)
    auto phase = [&]() __attribute__((always_inline)) {
        return reinterpret_cast<unsigned long long*>(a.wgbuf + (size_t)256 * kWgCap) +
               ((size_t)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 8;
    };
    constexpr bool stamping = QD == 234;  // wave-uniform: every lane stores the same word
    if (stamping) phase()[0] = __builtin_amdgcn_s_memrealtime();
#endif

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const int nkc = I8 ? a.ld8 / (2 * kFilterChunkK) : a.ld / kFilterChunkK;
    for (int t = threadIdx.x; t < kFilterQueries; t += kThreads) {
        float thr = a.thr[t], sqv = a.qscale[t], kev = a.ke[t];
        if (I8) {
            // the assembly computes w = float(I) * (row scale term); the per-query constants are rescaled by sq8 here
            const double sq8 = (double)a.sq8[t];
            if (SPACE == kSpaceCosine) {  // w >= T:  u = w*sq8 + ke >= thr  <=>  w >= (thr - ke)/sq8, rounded down
                if (thr > 1.0e30f) thr = 3.4e38f;
                else if (thr < -1.0e30f) thr = -3.4e38f;
                else thr = float_below(((double)thr - (double)a.ke8[t]) / sq8);
            } else if (SPACE == kSpaceIp) {  // w + ke' |x| >= thr/sq8  (the scatter multiplies the stored value by sq8)
                if (thr > 1.0e30f) thr = 3.4e38f;
                else if (thr < -1.0e30f) thr = -3.4e38f;
                else thr = float_below((double)thr / sq8);
                kev = float_above((double)kev / sq8);
            } else {  // l2: sq' (w + ke' |x|) + p1 >= thr with sq' = 2|q| sq8
                kev = float_above((double)kev / sq8);
                sqv = sqv * a.sq8[t];
                if (scan_code_l2c(QD) && t < a.nq && thr > -1.0e30f && thr < 1.0e30f)
                    thr = float_below((double)thr + (double)l2c_delta(a, t));  // (see l2c_delta)
            }
        }
        thr_l[t] = thr;
        sq_l[t] = sqv;
        ke_l[t] = kev;
    }
    const int64_t ntiles_all = tile_end - tile_begin;
    // Which tiles this workgroup scans: tile0 + i * tstride, i < my_tiles.  Default: the grid walks the range together
    // (workgroup b takes tiles b, b + grid, ...).  xcd_mode (MLVDB_SCAN_XCD=1, tuning): workgroups b, b + 8, ... share an
    // XCD under the observed round-robin placement (speed only, never correctness), so each XCD gets one contiguous
    // eighth of the range and its workgroups walk that eighth together.
    int64_t tile0 = tile_begin + blockIdx.x, my_tiles = ntiles_all > blockIdx.x ? (ntiles_all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    uint32_t tstride = gridDim.x;
    if (xcd_mode) {  // (the launcher only sets it for grids that are multiples of 8)
        const int64_t per = (ntiles_all + 7) / 8, x = blockIdx.x & 7, slot = blockIdx.x >> 3, gs = gridDim.x >> 3;
        const int64_t n_x = per * x < ntiles_all ? (per * (x + 1) <= ntiles_all ? per : ntiles_all - per * x) : 0;
        tile0 = tile_begin + per * x + slot;
        my_tiles = n_x > slot ? (n_x - slot + gs - 1) / gs : 0;
        tstride = (uint32_t)gs;
    }
    if (my_tiles == 0) return;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
    // The assembly toggles the Q buffers with xor 0x8000, so the dynamic LDS must start at 0, i.e. the kernel may have no
    // static LDS: launch_scan_asm checks that on the host (hipFuncGetAttributes) before the first launch.  Should it ever be
    // violated all the same, the workgroup hands its queries to the exact fallback (overflow flags) instead of trapping.
    if (lds_base != 0) {
        for (int t = threadIdx.x; t < a.nq; t += kThreads) a.overflow[t] = 1u;
        return;
    }

    const uint32_t chunk_bytes = (uint32_t)(kChunkVec * sizeof(uint4));
    const uint32_t pb = I8 ? (uint32_t)a.ld8 * 16u : (uint32_t)a.ld * 32u;  // bytes of one shadow panel (16 rows)
    const uint32_t wbytes = MT * pb;           // this wave's panels of a tile
    const uint64_t tile_bytes = (uint64_t)NW * wbytes;
    const int64_t first_tile = tile0;
    const uint64_t xbase = reinterpret_cast<uint64_t>(I8 ? a.X8 : a.Xb) + (uint64_t)first_tile * tile_bytes + (uint64_t)wave * wbytes;
    const uint64_t xstride = (uint64_t)tstride * tile_bytes;
    const uint64_t rnbase = reinterpret_cast<uint64_t>(I8 ? a.rp8 + 2 * (first_tile * kTileRowsV + wave * kWaveRows)  // pairs per row
                                                          : a.rn + first_tile * kTileRowsV + wave * kWaveRows);
    const uint32_t xlo = (uint32_t)xbase, xhi = (uint32_t)(xbase >> 32) & 0xffffu;
    const uint32_t xslo = (uint32_t)xstride, xshi = (uint32_t)(xstride >> 32);
    const uint32_t rnlo = (uint32_t)rnbase, rnhi = (uint32_t)(rnbase >> 32) & 0xffffu;
    const uint32_t rnstride = tstride * (uint32_t)(kTileRowsV * (I8 ? 8 : 4));
    const uint32_t row0 = (uint32_t)(first_tile * kTileRowsV);
    const uint32_t rowstride = tstride * (uint32_t)kTileRowsV;
    const uint32_t ntiles = (uint32_t)my_tiles;
    const uint32_t qbytes = (uint32_t)nkc * chunk_bytes;
    const uint32_t nb = (uint32_t)(2 * nkc / R);
    const uint32_t qcur0 = (uint32_t)(2 % nkc) * chunk_bytes;
    const uint32_t qc1 = (uint32_t)(1 % nkc) * chunk_bytes;
    u32x4s qsrd;
    {
        const uint64_t qb = reinterpret_cast<uint64_t>(I8 ? a.qimg8 : a.qimg);
        qsrd[0] = (uint32_t)qb;
        qsrd[1] = (uint32_t)(qb >> 32) & 0xffffu;
        qsrd[2] = qbytes;
        qsrd[3] = 0x00020000u;
    }
    // l2: p1 = k1 |x|^2; int8 cosine: K = 1.016 / min sq8, the factor of the rows' own errors (filter_prep8_fin_kernel)
    const float k1 = I8 && SPACE == kSpaceCosine ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.ke8[kFilterQueries])))
                                                 : -(1.0f - kSlack);
    // the later-dispatched half of the workgroup's waves (readfirstlane: an "s" operand must live in an SGPR)
    const uint32_t wtype = __builtin_amdgcn_readfirstlane(wave >= NW / 2 ? 1u : 0u);
    (void)wtype;
    // stagger (STAG): the later half runs nkc/2 chunk periods behind, its k origin rotated by half a row
    // (tools/gen_scan_asm.py, generate); needs the R ring steps after the rotated origin inside the panel
    const uint32_t hc = STAG && nkc % 2 == 0 && nkc >= R ? (uint32_t)nkc / 2 : 0u;
    const uint32_t xrot = __builtin_amdgcn_readfirstlane(wtype && hc ? (uint32_t)nkc * 1024u : 0u);
    const uint32_t pbrot = __builtin_amdgcn_readfirstlane(pb + xrot);
    const uint32_t pb2 = 2 * pb;
    (void)hc;
    (void)xrot;
    (void)pbrot;
    (void)pb2;
    // this wave's append buffer in global memory: u[cap], row[cap], q[cap]
    constexpr int kCapW = kWgCap / NW;
    const char* wgb = reinterpret_cast<const char*>(a.wgbuf + ((size_t)blockIdx.x * NW + wave) * kCapW);
    const float* wgbu = reinterpret_cast<const float*>(wgb);
    const int32_t* wgbr = reinterpret_cast<const int32_t*>(wgb + kCapW * 4);
    const uint32_t* wgbq = reinterpret_cast<const uint32_t*>(wgb + kCapW * 8);
    const uint32_t* wgcp = a.wgcnt + blockIdx.x * NW + wave;
    const uint32_t stg = (uint32_t)(kQBufs * chunk_bytes) + 3 * kFilterQueries * sizeof(float) + (uint32_t)wave * (12 * kStageCap);
    const uint32_t* ovfb = a.overflow;
    const uint32_t lane16 = lane * 16;
    // LDS-DMA staging: which fragment of a Q chunk this wave moves (tools/gen_scan_asm.py, dma_pieces): query tile `wave` (+ 8), both
    // k-step halves -- or, when only 4 query tiles are computed, the single fragment (tile wave & 3, half wave >> 2)
    constexpr int kNQT = scan_code_nqt(QD);
    const uint32_t wave_piece = kNQT == 4 ? (uint32_t)(wave & 3) * 2048u + (uint32_t)(wave >> 2) * 1024u : (uint32_t)wave * 2048u;
    const uint32_t qvoff = wave_piece + lane16;  // this thread's uint4 of the wave's fragment
    const uint32_t rnvoff = g * (I8 ? 32 : 16);
    const uint32_t thra = (uint32_t)(kQBufs * chunk_bytes) + c16 * 4;
    const uint32_t c16v = c16;
    const uint32_t crow = (uint32_t)wave * (uint32_t)kWaveRows + g * 4;
    static_assert(kWgCap == 16384 && sizeof(WgEntry) == 16, "the assembly hard-codes the append buffer geometry");

    // l2e bodies: the offsets plane lies 8 x rp8_cap bytes behind the pairs; eo0 = that distance less 4 bytes per row of this
    // wave's first row (the pairs' descriptor base already points 8 bytes per row into the array): mod 2^32, < 2^32 by the
    // launcher's check
    const uint32_t eo0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(8ull * (uint64_t)a.rp8_cap - 4ull * (uint64_t)(first_tile * kTileRowsV + wave * kWaveRows)));
    (void)eo0;
    // l2c bodies: the pass's common query scale SQ and error coefficient KE (filter_l2_offsets_kernel; SGPR operands)
    const float sqc = scan_code_l2e(QD) ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.l2c_out[0]))) : 0.f;
    const float kec = scan_code_l2e(QD) ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.l2c_out[1]))) : 0.f;
    const float krc = scan_code_l2e(QD) ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.l2c_out[2]))) : 0.f;
    (void)sqc;
    (void)kec;
    (void)krc;
    u32x4s veo[MT];
    uint32_t s_eo;
    (void)veo;
    (void)s_eo;
    u32x4s xring[R * MT], qsa[kQPer], qsb[kQPer], qt[scan_code_qd(QD)];
    float vr[4 * MT], vp[4 * MT], vu[4 * MT], vs[4 * MT], vt[16];
    (void)vs;
    (void)vt;
    uint32_t ve[13], ldr, ldw, s_sldw;
    const uint32_t wave2k = __builtin_amdgcn_readfirstlane(wave_piece);  // LDS-DMA staging: this wave's fragment offset inside a Q buffer
    (void)ldw;
    (void)s_sldw;
    (void)wave2k;
    (void)qsa;
    (void)qsb;
    uint32_t s_xso0, s_xso1, s_xso2, s_xso3, s_qcur, s_cnt, s_st0, s_tl, s_trow, s_sn64, s_wcnt, s_sacc0, s_sacc1;
    (void)vp;
    (void)s_xso2;
    (void)s_xso3;
    (void)k1;
#ifdef MLVDB_SCAN_DIAGNOSTICS
    if (stamping) phase()[1] = __builtin_amdgcn_s_memrealtime();
#endif
#include "scan_asm_dispatch.inc"
#ifdef MLVDB_SCAN_DIAGNOSTICS
    if (stamping) phase()[4] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- the workgroup's own scatter: append buffers -> per-query candidate lists (what filter_scatter_kernel did in a
    // launch of its own).  Every wave staged its entries {u[], row[], q[]} in its LDS area (the first kStageCap of
    // them; later ones went to its slice of a.wgbuf, same slot numbering, stores drained); s_wcnt = how many it
    // appended.  Entries are counted per query in LDS first, so the workgroup issues one device-scope atomic per query it
    // has entries for.  The Q buffers are free by now: every wave has passed the last chunk barrier of its last tile.
    {
        uint32_t* hist = reinterpret_cast<uint32_t*>(smem);          // [256]
        uint32_t* lbase = hist + kFilterQueries;                      // [256]
        uint32_t* wcount = lbase + kFilterQueries;                    // [NW]
        if (lane == 0) {
            wcount[wave] = s_wcnt;
            a.wgcnt[blockIdx.x * NW + wave] = s_wcnt;                 // tuning aid (MLVDB_DEBUG_ENTRIES)
        }
        for (int t = threadIdx.x; t < kFilterQueries; t += kThreads) hist[t] = 0;
        __syncthreads();
        constexpr uint32_t kArea = 12u * kStageCap;
        const uint32_t stage0 = (uint32_t)(kQBufs * chunk_bytes) + 3 * kFilterQueries * sizeof(float);
        auto entry_q = [&](int w, uint32_t i) -> uint32_t {
            if (i < (uint32_t)kStageCap) return *reinterpret_cast<const uint32_t*>(smem + stage0 + w * kArea + 8u * kStageCap + 4u * i);
            const char* g = reinterpret_cast<const char*>(a.wgbuf + ((size_t)blockIdx.x * NW + w) * kCapW);
            return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(g + kCapW * 8) + i);
        };
        for (int w = 0; w < NW; ++w) {
            const uint32_t n = min(wcount[w], (uint32_t)kCapW);  // entries past the global slice were flagged by the scan
            for (uint32_t i = threadIdx.x; i < n; i += kThreads) atomicAdd(&hist[entry_q(w, i)], 1u);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < kFilterQueries; t += kThreads) {
            const uint32_t c = hist[t];
            const uint32_t b = c ? atomicAdd(&a.cnt[t], c) : 0u;
            lbase[t] = b;
            if (c && b + c > (uint32_t)a.cand_cap) a.overflow[t] = 1u;
            hist[t] = 0;
        }
        __syncthreads();
        constexpr int kI8Mode = !I8 ? 0 : (SPACE == kSpaceCosine ? 1 : (SPACE == kSpaceIp ? 2 : 0));
        for (int w = 0; w < NW; ++w) {
            const uint32_t n = min(wcount[w], (uint32_t)kCapW);
            const char* g = reinterpret_cast<const char*>(a.wgbuf + ((size_t)blockIdx.x * NW + w) * kCapW);
            for (uint32_t i = threadIdx.x; i < n; i += kThreads) {
                float u;
                int32_t row;
                if (i < (uint32_t)kStageCap) {
                    u = *reinterpret_cast<const float*>(smem + stage0 + w * kArea + 4u * i);
                    row = *reinterpret_cast<const int32_t*>(smem + stage0 + w * kArea + 4u * kStageCap + 4u * i);
                } else {
                    u = __builtin_nontemporal_load(reinterpret_cast<const float*>(g) + i);
                    row = __builtin_nontemporal_load(reinterpret_cast<const int32_t*>(g + kCapW * 4) + i);
                }
                const uint32_t q = entry_q(w, i);
                const uint32_t slot = lbase[q] + atomicAdd(&hist[q], 1u);
                if (slot < (uint32_t)a.cand_cap) {
                    CandEntry e;
                    // int8 scan: the stored value is in units of the query's scale (cosine: w, ip: w + ke' |x|)
                    e.u = kI8Mode == 1 ? __builtin_fmaf(u, a.sq8[q], a.ke8[q]) : (kI8Mode == 2 ? u * a.sq8[q] : u);
                    if (scan_code_l2c(QD)) e.u = float_above((double)u - (double)l2c_delta(a, (int)q));
                    e.row = row;
                    a.cand[(size_t)q * a.cand_cap + slot] = e;
                }
            }
        }
    }
#ifdef MLVDB_SCAN_DIAGNOSTICS
    if (stamping) phase()[5] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------ threshold update + compaction
// eps such that l = u - 2*eps <= s <= u for the entry's (query,row); mirrors the scan epilogue.
__device__ __forceinline__ float entry_eps(int space, float ke, float sq, float nr) {
    if (space == kSpaceCosine) return ke;
    if (space == kSpaceIp) return ke * nr;
    return sq * ke * nr + kSlack * nr * nr;
}


// One block per query.  thr[q] = max(thr[q], k-th largest lower bound l = u - 2 eps); entries with
// u < thr are dropped.  The k-th largest lower bound is found by a 4-pass radix select on the order
// keys through an LDS histogram (no shuffles, 8 barriers), then the list is compacted through LDS.
constexpr int kUpdThreads = 256;
__global__ __launch_bounds__(kUpdThreads) void filter_update_kernel(const FilterArgs a, const int32_t k,
                                                                      const int32_t forced_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CandEntry* stage = reinterpret_cast<CandEntry*>(smem);                                   // [kCandCap]
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem + kCandCap * sizeof(CandEntry));       // [kCandCap]
    uint32_t* hist = keys + kCandCap;                                                        // [256]
    uint32_t* s_scan = hist + 256;                                                           // [16]
    const int q = blockIdx.x;
    if (q >= a.nq || a.overflow[q]) return;
    const uint32_t cnt = forced_cnt >= 0 ? (uint32_t)forced_cnt : a.cnt[q];  // after the dense seeding pass: every slot
    if (cnt > (uint32_t)kCandCap) {  // cannot happen without the flag, but never index past the list
        if (threadIdx.x == 0) a.overflow[q] = 1u;
        return;
    }
    CandEntry* list = a.cand + (int64_t)q * kCandCap;
    const float sq = a.qscale[q];
    const float ke = a.ke[q];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool need_norm = a.space != kSpaceCosine;
    float thr = a.thr[q];
    // order keys of the lower bounds (0 = not a candidate: NaN bound of a tombstoned / padding row)
    uint32_t n_valid = 0, kmin = 0xffffffffu, kmax = 0;
    for (uint32_t idx = threadIdx.x; idx < cnt; idx += kUpdThreads) {
        const CandEntry e = list[idx];
        stage[idx] = e;
        uint32_t key = 0;
        if (e.u == e.u) {
            const float eps = entry_eps(a.space, ke, sq, need_norm ? a.rn[e.row] : 0.f);
            float l = e.u - 2.0f * eps;
            l -= kSlack * (__builtin_fabsf(e.u) + eps);  // rounding of the line above
            key = float_order_key(l);
            if (key == 0) key = 1;
            ++n_valid;
            kmin = min(kmin, key);
            kmax = max(kmax, key);
        }
        keys[idx] = key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_valid += __shfl_xor(n_valid, off);
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, off));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off));
    }
    if (lane == 0) {
        s_scan[wave] = n_valid;
        s_scan[8 + wave] = kmin;
        s_scan[12 + wave] = kmax;
    }
    __syncthreads();
    n_valid = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    kmin = min(min(s_scan[8], s_scan[9]), min(s_scan[10], s_scan[11]));
    kmax = max(max(s_scan[12], s_scan[13]), max(s_scan[14], s_scan[15]));
    if (n_valid >= (uint32_t)k) {
        // Radix select, most significant digit first: prefix/mask narrow the keys, `want` = rank still
        // wanted.  The keys of one list are floats of similar size: their leading bits are all equal, and a
        // histogram over those bits would be 256 threads adding to one LDS word (measured: the bulk of the
        // kernel).  So the first digit starts at the highest bit in which any two keys differ.
        const uint32_t diff = kmin ^ kmax;
        uint32_t prefix = kmax, mask = 0xffffffffu, want = (uint32_t)k;  // all keys equal: the answer is that key
        if (diff != 0) {
            const int hb = 31 - __builtin_clz(diff);
            mask = hb == 31 ? 0u : ~((2u << hb) - 1u);
            prefix = kmax & mask;
            int shift = hb >= 7 ? hb - 7 : 0, width = hb - shift + 1;
            for (;;) {
                __syncthreads();
                hist[threadIdx.x] = 0;
                __syncthreads();
                const uint32_t dmask = (1u << width) - 1u;
                for (uint32_t idx = threadIdx.x; idx < cnt; idx += kUpdThreads) {
                    const uint32_t key = keys[idx];
                    if (key != 0 && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
                }
                __syncthreads();
                if (wave == 0) {
                    // suffix sums over the 256 bins: lane owns bins 4*lane .. 4*lane+3
                    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                    uint32_t above = h0 + h1 + h2 + h3;  // becomes the count in bins of higher lanes
                    uint32_t run = above;
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t v = __shfl_down(run, off);
                        if (lane + off < 64) run += v;
                    }
                    above = run - above;  // keys in bins > 4*lane+3
                    // the wanted bin is the highest bin b with (count in bins >= b) >= want
                    const uint32_t c3 = above + h3, c2 = c3 + h2, c1 = c2 + h1, c0 = c1 + h0;
                    int bin = -1;
                    uint32_t before = 0;  // keys in bins strictly above the chosen one
                    if (above < want && c0 >= want) {
                        if (c3 >= want) { bin = 4 * lane + 3; before = above; }
                        else if (c2 >= want) { bin = 4 * lane + 2; before = c3; }
                        else if (c1 >= want) { bin = 4 * lane + 1; before = c2; }
                        else { bin = 4 * lane; before = c1; }
                    }
                    if (bin >= 0) {
                        s_scan[4] = (uint32_t)bin;
                        s_scan[5] = before;
                    }
                }
                __syncthreads();
                prefix |= s_scan[4] << shift;
                mask |= dmask << shift;
                want -= s_scan[5];
                if (shift == 0) break;
                const int next = shift >= 8 ? shift - 8 : 0;
                width = shift - next;
                shift = next;
            }
        }
        const float lk = float_from_order_key(prefix);
        if (lk > thr) thr = lk;
    }
    // compaction through LDS: survivors (u >= thr) in list order
    uint32_t keep = 0;
    for (uint32_t idx = threadIdx.x * 32; idx < min(cnt, threadIdx.x * 32 + 32); ++idx) keep += stage[idx].u >= thr ? 1u : 0u;
    uint32_t incl = keep;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - keep;
    for (int w = 0; w < wave; ++w) pos += s_scan[w];
    const uint32_t new_cnt = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    if (new_cnt != cnt) {
        for (uint32_t idx = threadIdx.x * 32; idx < min(cnt, threadIdx.x * 32 + 32); ++idx) {
            const CandEntry e = stage[idx];
            if (e.u >= thr) list[pos++] = e;
        }
    }
    if (threadIdx.x == 0) {
        a.thr[q] = thr;
        a.cnt[q] = new_cnt;
    }
}

// ------------------------------------------------------------------ exact rescoring
// Two kernels.  filter_rescore_score_kernel: the surviving rows of ALL queries are scored by the exact-scan arithmetic
// (accumulate_rows) as one flat list of 16-row groups dealt out to the waves of the grid in turn, so that every CU gathers
// the same number of rows; the exact (distance, label) pairs go to a.rs.  Why: a CU pulls ~24 GB/s from HBM
// (MI355X_MICROARCH.md), a row is 48 pieces of 64 bytes in half-used 128-byte lines, so 100 rows cost a CU ~27 us however
// the loads are issued -- one block per query made the kernel as slow as its longest list (70-78 us for the 10M x 768
// wave, mean block 33 us), eight blocks per query left the CUs with 2..4 blocks each (46-56 us):
// profiles/r02/refine_rescore_phase_stamps_10m.txt.
// filter_rescore_rank_kernel: one block per query ranks the pairs by (distance, label) and writes the answer.  (One kernel
// whose last-finishing block ranks was tried first: the agent-scope fences it needs write the XCD's L2 back per block --
// 264 us instead of 87.  The kernel boundary does that once.)
constexpr int kRescoreWaves = 16;      // waves per block (fewer when that many copies of the query do not fit in LDS)
constexpr int kRescoreGrid = 256;      // one block per CU: each asks for more than half of a CU's LDS, so no two share a CU, and the
                                       // 16-row groups are dealt to the blocks in turn -- every CU gathers the same number of rows
                                       // (+-16), each group on a wave of its own while there are at most 4096 of them
constexpr int kRescorePF = 8;          // column groups per prefetch bank (accumulate_rows; the summation order is fixed).  A whole
                                       // 768-column row in flight per lane (24 groups, two banks, 8-wave blocks) was slower: the
                                       // gather is bound by bytes per CU, not by round trips
constexpr int kRescoreRankMax = 2048;  // lists up to this length are ranked by counting in LDS
constexpr int kRankWaves = 16;         // waves of the ranking block
template <int SPACE>
__global__ __launch_bounds__(kRescoreWaves * 64) void filter_rescore_score_kernel(const FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t pre[kFilterQueries + 1];  // 16-row groups of the queries before q (overflowed queries: none)
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    double* qs = reinterpret_cast<double*>(smem) + (size_t)wave * ld;  // this wave's copy of its current query
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: phase stamps (100 MHz) of every wave, read back by api.hip (MLVDB_DEBUG_REFINE)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.wgbuf) + ((size_t)blockIdx.x * nwaves + wave) * 4;
#define SCORE_STAMP(i) do { if (lane == 0 && a.wgbuf) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SCORE_STAMP(i) do { } while (0)
#endif
    SCORE_STAMP(0);
    if (wave == 0) {  // lane l: queries 4l .. 4l+3; inclusive scan over the lanes (a serial loop over LDS here cost 12 us)
        uint32_t n[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 4 * lane + i;
            n[i] = q < a.nq && !a.overflow[q] ? (min(a.cnt[q], (uint32_t)kCandCap) + 15u) >> 4 : 0u;
            sum += n[i];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        uint32_t run = incl - sum;
        if (lane == 0) pre[0] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            run += n[i];
            pre[4 * lane + i + 1] = run;
        }
    }
    __syncthreads();
    const uint32_t total = pre[kFilterQueries];
    SCORE_STAMP(1);
    const int g = lane >> 4, r = lane & 15;
    int cur = -1;
    double qinv = 0.0;
    for (uint32_t u = blockIdx.x + gridDim.x * wave; u < total; u += gridDim.x * nwaves) {
        int lo = 0, hi = kFilterQueries;  // the query with pre[q] <= u < pre[q + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= u) lo = mid;
            else hi = mid;
        }
        const int q = lo;
        if (q != cur) {  // (wave-private LDS: no barrier; the wave's own earlier reads are done -- its loop is in order)
            for (int c = lane; c < ld; c += 64) qs[c] = (double)a.Qpad[(int64_t)q * ld + c];
            qinv = a.qaux[q];
            cur = q;
        }
        SCORE_STAMP(2);
        const uint32_t cnt = min(a.cnt[q], (uint32_t)kCandCap);
        const CandEntry* list = a.cand + (int64_t)q * kCandCap;
        RangeHit* rs = a.rs + (int64_t)q * kCandCap;
        const uint32_t idx = (u - pre[q]) * 16 + r;
        const bool have = idx < cnt;
        const int32_t row = have ? list[idx].row : 0;
        const float* base[1] = {a.X + (int64_t)(row >> 4) * (kPanelRows * ld) + (row & 15) * 16 + g * 4};
        double acc[1][1], nx[1];
        accumulate_rows<SPACE, 1, 1, kRescorePF>(base, qs, ld, g, acc, nx);
        const double dist = finish_distance<SPACE>(acc[0][0], nx[0], qinv);
        if (have && lane < 16) {
            const float nrm = a.rn[row];
            const bool live = nrm == nrm;
            RangeHit hit;
            hit.d = live ? dist : __builtin_inf();
            hit.l = live ? row : kNoLabel;
            hit.pad = 0;
            rs[idx] = hit;
        }
        SCORE_STAMP(3);
    }
#undef SCORE_STAMP
}

// Block 0 also compacts the indices of the overflowed queries (ascending) for the exact fallback that follows when the
// call cannot ask the host (qsel != nullptr; a launch of its own until round 2).
__global__ __launch_bounds__(kRankWaves * 64) void filter_rescore_rank_kernel(const FilterArgs a, const int32_t k, const int32_t q0,
                                                                                  int64_t* out_labels, float* out_dist,
                                                                                  int32_t* out_counts, double* out_d64,
                                                                                  unsigned long long* rescored, int32_t* qsel,
                                                                                  int32_t* nflag) {
    __shared__ double ed[kRescoreRankMax];  // exact distances
    __shared__ int32_t el[kRescoreRankMax];  // labels
    __shared__ double sd[kRankWaves][64];
    __shared__ int32_t sl[kRankWaves][64];
    __shared__ int32_t s_nvalid;
    __shared__ int32_t s_flagged[4];
    const int q = blockIdx.x;
    if (blockIdx.x == 0 && qsel) {  // block-uniform
        const int t = threadIdx.x;
        const bool flagged = t < kFilterQueries && t < a.nq && a.overflow[t] != 0;
        const unsigned long long bal = __ballot(flagged);
        if (t < kFilterQueries && (t & 63) == 0) s_flagged[t >> 6] = __popcll(bal);
        __syncthreads();
        if (t < kFilterQueries) {
            int base = 0;
            for (int w = 0; w < (t >> 6); ++w) base += s_flagged[w];
            if (flagged) qsel[base + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = t;
        }
        if (t == 0) {
            const int n = s_flagged[0] + s_flagged[1] + s_flagged[2] + s_flagged[3];
            *nflag = n;
            if (rescored) rescored[1] += (unsigned long long)n;
        }
    }
#ifdef MLVDB_SCAN_DIAGNOSTICS
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.wgbuf) + 16384 + (size_t)blockIdx.x * 4;
#define RANK_STAMP(i) do { if (threadIdx.x == 0 && a.wgbuf) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RANK_STAMP(i) do { } while (0)
#endif
    RANK_STAMP(0);
    if (q >= a.nq || a.overflow[q]) return;
    const uint32_t cnt = min(a.cnt[q], (uint32_t)kCandCap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RangeHit* rs = a.rs + (int64_t)q * kCandCap;
    const int64_t o = (int64_t)(q0 + q) * k;
    if (threadIdx.x == 0) s_nvalid = 0;
    if (cnt <= (uint32_t)kRescoreRankMax) {
        for (uint32_t i = threadIdx.x; i < cnt; i += kRankWaves * 64) {
            const RangeHit h = rs[i];
            ed[i] = h.d;
            el[i] = h.l;
        }
        __syncthreads();
        RANK_STAMP(1);
        // an entry's rank = how many entries precede it in (distance, label) order; one entry per wave at a time, 64
        // comparisons per step by ballot, given up as soon as k entries precede it (wave-uniform).  (A thread per entry
        // looping over the whole list took 13.5 us for ~100 entries; 4 waves without the early exit 16 us, and 70 us for
        // the longest list of the wave -- a step is ~200 cycles of dependent LDS latency.)
        int mine = 0;
        for (uint32_t i = wave; i < cnt; i += kRankWaves) {
            const double di = ed[i];
            const int32_t li = el[i];
            if (li == kNoLabel) continue;  // wave-uniform
            ++mine;
            int rank = 0;
            for (uint32_t j0 = 0; j0 < cnt && rank < k; j0 += 128) {
                const uint32_t j = j0 + lane, j2 = j + 64;
                const bool l0 = j < cnt && entry_less(ed[j], el[j], di, li);
                const bool l1 = j2 < cnt && entry_less(ed[j2], el[j2], di, li);
                rank += __popcll(__ballot(l0)) + __popcll(__ballot(l1));
            }
            if (lane == 0 && rank < k) {
                out_labels[o + rank] = (int64_t)li;
                out_dist[o + rank] = (float)di;
                if (out_d64) out_d64[o + rank] = di;
            }
        }
        if (lane == 0 && mine) atomicAdd(&s_nvalid, mine);
        __syncthreads();
        RANK_STAMP(2);
        const int n_out = min(s_nvalid, k);
        for (int i = n_out + threadIdx.x; i < k; i += kRankWaves * 64) {
            out_labels[o + i] = -1;
            out_dist[o + i] = __builtin_inff();
            if (out_d64) out_d64[o + i] = __builtin_inf();
        }
        if (threadIdx.x == 0) {
            out_counts[q0 + q] = n_out;
            if (rescored) atomicAdd(rescored, (unsigned long long)cnt);
        }
        RANK_STAMP(3);
#undef RANK_STAMP
        return;
    }
    // long lists (k <= 64 here): per-wave sorted top-k lists, merged by wave 0
    WaveTopK top;
    top.init();
    for (uint32_t b = 0; b < cnt; b += kRankWaves * 64) {
        const uint32_t i = b + wave * 64 + lane;
        RangeHit h;
        h.d = __builtin_inf();
        h.l = kNoLabel;
        if (i < cnt) h = rs[i];
        top.offer(h.l != kNoLabel, h.d, h.l, k, lane);
    }
    sd[wave][lane] = top.d;
    sl[wave][lane] = top.l;
    __syncthreads();
    if (wave != 0) return;
    WaveTopK f;
    f.init();
#pragma unroll
    for (int w2 = 0; w2 < kRankWaves; ++w2)
        f.offer(lane < k && sl[w2][lane] != kNoLabel, sd[w2][lane], sl[w2][lane], k, lane);
    const bool valid = lane < k && f.l != kNoLabel;
    if (lane < k) {
        out_labels[o + lane] = valid ? (int64_t)f.l : -1;
        out_dist[o + lane] = valid ? (float)f.d : __builtin_inff();
        if (out_d64) out_d64[o + lane] = valid ? f.d : __builtin_inf();
    }
    const int n_valid = __popcll(__ballot(valid));
    if (lane == 0) {
        out_counts[q0 + q] = n_valid;
        if (rescored) atomicAdd(rescored, (unsigned long long)cnt);
    }
}

// Range variant, two kernels.  A range pass can hold tens of thousands of candidates for one query and a handful for
// the next (hit counts vary ~1000x with |q| on unnormalised data), so the exact rescoring is spread over
// (query, chunk of kRangeChunk candidates) blocks instead of one block per query:
//   range_score_kernel  exact fp64 distance of every candidate of its chunk; the hits (dist <= radius, live) are
//                       collected in LDS and copied to the query's hit array behind one atomic reservation;
//   range_rank_kernel   blocks over (query, 256 hits): rank of every hit by (distance, label), written to its place;
//                       publishes the exact count.  More than kCandCap hits: flagged, served by the paged exact kNN (api.hip).
template <int SPACE>
__global__ __launch_bounds__(256) void range_score_kernel(const FilterArgs a, const float radius) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);                                  // [ld]
    RangeHit* found = reinterpret_cast<RangeHit*>(qs + a.ld);                      // [kRangeChunk]
    uint32_t& s_n = *reinterpret_cast<uint32_t*>(found + kRangeChunk);
    uint32_t& s_base = *(&s_n + 1);
    const int q = blockIdx.x;
    if (q >= a.nq || a.overflow[q]) return;
    const uint32_t cnt = min(a.cnt[q], (uint32_t)a.cand_cap);
    const uint32_t begin = blockIdx.y * (uint32_t)kRangeChunk;
    if (begin >= cnt) return;
    const uint32_t end = min(cnt, begin + (uint32_t)kRangeChunk);
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    for (int c = threadIdx.x; c < ld; c += 256) qs[c] = (double)a.Qpad[(int64_t)q * ld + c];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const double qinv = a.qaux[q];
    const double rad = (double)radius;
    const CandEntry* list = a.cand + (int64_t)q * a.cand_cap;
    for (uint32_t i0 = begin + wave * 16; i0 < end; i0 += 64) {
        const uint32_t idx = i0 + r;
        const bool have = idx < end;
        const int32_t row = have ? (int32_t)((uint32_t)list[idx].row & ~kRefinedBit) : 0;
        const float* base[1] = {a.X + (int64_t)(row >> 4) * (kPanelRows * ld) + (row & 15) * 16 + g * 4};
        double acc[1][1], nx[1];
        accumulate_rows<SPACE, 1, 1, 8>(base, qs, ld, g, acc, nx);
        const double dist = finish_distance<SPACE>(acc[0][0], nx[0], qinv);
        bool hit = have && lane < 16 && dist <= rad;
        if (hit) {
            const float nrm = a.rn[row];
            hit = nrm == nrm;
        }
        if (hit) {
            const uint32_t slot = atomicAdd(&s_n, 1u);
            found[slot].d = dist;
            found[slot].l = row;
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&a.rhit_cnt[q], n);
    __syncthreads();
    const uint32_t base = s_base;
    RangeHit* out = a.rhits + (int64_t)q * kCandCap;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        if (base + i < (uint32_t)kCandCap) out[base + i] = found[i];  // beyond: counted only (the query is paged exactly)
}

// Round 3: the same work as range_score_kernel, dealt out like the kNN rescoring (filter_rescore_score_kernel): the
// 16-candidate groups of ALL queries form one flat list, group u goes to wave (u mod waves-of-the-grid), one block per CU --
// every CU gathers the same number of rows whatever the spread of the list lengths (a query's hit count varies ~1000x with
// |q|: the (query, chunk) grid launched 65,536 blocks of which ~1,000 had work, 820 us per 256-query wave).  A group's
// hits (dist <= radius, live) are appended to the query's hit array behind one atomic per group.
template <int SPACE>
__global__ __launch_bounds__(kRescoreWaves * 64) void range_score_flat_kernel(const FilterArgs a, const float radius) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t pre[kFilterQueries + 1];  // 16-candidate groups of the queries before q (overflowed queries: none)
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    double* qs = reinterpret_cast<double*>(smem) + (size_t)wave * ld;  // this wave's copy of its current query
    if (wave == 0) {  // lane l: queries 4l .. 4l+3; inclusive scan over the lanes
        uint32_t n[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 4 * lane + i;
            n[i] = q < a.nq && !a.overflow[q] ? (min(a.cnt[q], (uint32_t)a.cand_cap) + 15u) >> 4 : 0u;
            sum += n[i];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        uint32_t run = incl - sum;
        if (lane == 0) pre[0] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            run += n[i];
            pre[4 * lane + i + 1] = run;
        }
    }
    __syncthreads();
    const uint32_t total = pre[kFilterQueries];
    const int g = lane >> 4, r = lane & 15;
    const double rad = (double)radius;
    int cur = -1;
    double qinv = 0.0;
    for (uint32_t u = blockIdx.x + gridDim.x * wave; u < total; u += gridDim.x * nwaves) {
        int lo = 0, hi = kFilterQueries;  // the query with pre[q] <= u < pre[q + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= u) lo = mid;
            else hi = mid;
        }
        const int q = lo;
        if (q != cur) {  // (wave-private LDS: no barrier; the wave's own earlier reads are done -- its loop is in order)
            for (int c = lane; c < ld; c += 64) qs[c] = (double)a.Qpad[(int64_t)q * ld + c];
            qinv = a.qaux[q];
            cur = q;
        }
        const uint32_t cnt = min(a.cnt[q], (uint32_t)a.cand_cap);
        const CandEntry* list = a.cand + (int64_t)q * a.cand_cap;
        const uint32_t idx = (u - pre[q]) * 16 + r;
        const bool have = idx < cnt;
        const int32_t row = have ? (int32_t)((uint32_t)list[idx].row & ~kRefinedBit) : 0;  // (bit 31: the entry carries a mid bound)
        const float* base[1] = {a.X + (int64_t)(row >> 4) * (kPanelRows * ld) + (row & 15) * 16 + g * 4};
        double acc[1][1], nx[1];
        accumulate_rows<SPACE, 1, 1, kRescorePF>(base, qs, ld, g, acc, nx);
        const double dist = finish_distance<SPACE>(acc[0][0], nx[0], qinv);
        bool hit = have && lane < 16 && dist <= rad;
        if (hit) {
            const float nrm = a.rn[row];
            hit = nrm == nrm;
        }
        const unsigned long long bal = __ballot(hit);
        if (bal) {  // wave-uniform
            uint32_t base_slot = 0;
            if (lane == 0) base_slot = atomicAdd(&a.rhit_cnt[q], (uint32_t)__popcll(bal));
            base_slot = __shfl(base_slot, 0);
            if (hit) {
                const uint32_t slot = base_slot + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (slot < (uint32_t)kCandCap) {  // beyond: counted only (the query is paged exactly)
                    RangeHit h;
                    h.d = dist;
                    h.l = row;
                    h.pad = 0;
                    a.rhits[(int64_t)q * kCandCap + slot] = h;
                }
            }
        }
    }
}

// Ordering the hits of a query by (distance, label) -- by RANKING, not sorting: a hit's rank is the number of hits that
// precede it, ranks are a permutation (labels are unique), so every hit is written straight to its place.  Blocks over
// (query, chunk of 256 hits); every block holds the query's whole hit list in LDS as {d[], l[]} and each thread counts the
// predecessors of one hit (all lanes read the same LDS word per step: a broadcast).  Round 2 sorted each list with one
// bitonic network per query in LDS: 91 stages x 8192 x 16-byte records for the longest list of a wave (6,554 hits) are
// 36 MB through one CU's LDS -- 300-380 us per 256-query wave with 255 CUs idle (profiles/r03/config4_l2_range_kernel_stats.csv).
// Here the longest list is spread over 26 blocks of 6,554 steps each.  More than kCandCap hits: flagged, served by the
// paged exact kNN (api.hip), as before.
constexpr int kRankThreads = 256;
constexpr int kRankTargets = 32;   // hits ranked per work item: 32 targets x 8 list segments = 256 threads
constexpr int kRankSegs = kRankThreads / kRankTargets;
constexpr int kRankGrid = 256;  // one block per CU; the (query, 32-hit chunk) work items are dealt to the blocks in turn.  (A grid of
                                // nq x 32 blocks, most of which exit at once, took 2.5 ms just to be dispatched: every block asks
                                // for 98 KB of LDS.)
// knn_k > 0 (big-k passes, launch_knn_rescore_rank): the same ranking as the end of a kNN pass -- capacity = knn_k, counts are
// int32 (min(hits, k)), tails are padded (label -1, distance +inf), the fp64 distances go to out_d64 when asked for.
struct KnnOut {
    int32_t k;                    // 0: range mode
    int32_t* counts;
    double* d64;
    unsigned long long* rescored;
};
__global__ __launch_bounds__(kRankThreads) void range_rank_kernel(const FilterArgs a, const int32_t q0, const int64_t capacity,
                                                                    int64_t* out_labels, float* out_dist, int64_t* out_counts,
                                                                    const KnnOut ko) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t pre[kFilterQueries + 1];  // work items of the queries before q
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {  // lane l: queries 4l .. 4l+3; inclusive scan over the lanes
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 4 * lane + i;
            uint32_t n = 0;
            if (q < a.nq && !a.overflow[q]) {
                n = a.rhit_cnt[q];
                if (blockIdx.x == 0) {
                    if (n > (uint32_t)kCandCap) {  // more hits than a list holds: the exact count is known, the hits come from
                        a.overflow[q] = 2u;         // the paged exact kNN (api.hip reads the flag after the kernel)
                        a.cnt[q] = n;
                    }
                    if (ko.k > 0) {
                        const int32_t nk = n > (uint32_t)kCandCap ? 0 : (int32_t)min(n, (uint32_t)ko.k);
                        ko.counts[q0 + q] = nk;
                        for (int32_t i = nk; i < ko.k; ++i) {  // fewer than k live rows: pad (overflowed queries are redone whole)
                            out_labels[(int64_t)(q0 + q) * capacity + i] = -1;
                            out_dist[(int64_t)(q0 + q) * capacity + i] = __builtin_inff();
                            if (ko.d64) ko.d64[(int64_t)(q0 + q) * capacity + i] = __builtin_inf();
                        }
                        if (ko.rescored) atomicAdd(ko.rescored, (unsigned long long)min(n, (uint32_t)kCandCap));
                    } else {
                        out_counts[q0 + q] = n;
                    }
                }
                if (n > (uint32_t)kCandCap) n = 0;
            }
            c[i] = (n + kRankTargets - 1) / kRankTargets;
            sum += c[i];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        uint32_t run = incl - sum;
        if (lane == 0) pre[0] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            run += c[i];
            pre[4 * lane + i + 1] = run;
        }
    }
    __syncthreads();
    const uint32_t total = pre[kFilterQueries];
    const int tgt = threadIdx.x / kRankSegs, seg = threadIdx.x % kRankSegs;  // the 8 segment threads of a target are neighbours
    int cur = -1;
    uint32_t n = 0, n8 = 0;
    for (uint32_t u = blockIdx.x; u < total; u += gridDim.x) {  // block-uniform
        int lo = 0, hi = kFilterQueries;  // the query with pre[q] <= u < pre[q + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= u) lo = mid;
            else hi = mid;
        }
        const int q = lo;
        double* ed = reinterpret_cast<double*>(smem);  // [n8]
        if (q != cur) {  // (consecutive items of a block usually belong to different queries; the longest list's to many blocks)
            n = a.rhit_cnt[q];  // (<= kCandCap: larger lists have no items)
            n8 = (n + 63u) & ~63u;  // padded with (+inf, kNoLabel), which precedes nothing: every segment a whole number of 8-entry steps
            int32_t* el_w = reinterpret_cast<int32_t*>(ed + n8);
            const RangeHit* src = a.rhits + (int64_t)q * kCandCap;
            __syncthreads();  // the previous item's readers are done with the LDS copy
            for (uint32_t i = threadIdx.x; i < n8; i += kRankThreads) {
                RangeHit h;
                h.d = __builtin_inf();
                h.l = kNoLabel;
                if (i < n) h = src[i];
                ed[i] = h.d;
                el_w[i] = h.l;
            }
            __syncthreads();
            cur = q;
        }
        const int32_t* el = reinterpret_cast<const int32_t*>(ed + n8);  // [n8], 16-byte aligned
        const uint32_t i = (u - pre[q]) * (uint32_t)kRankTargets + tgt;
        // Rank of hit i = how many list entries precede it by (distance, label).  The list is cut into 8 segments, one per
        // thread of the target; 8 entries per step, read as 4 + 2 16-byte LDS words and compared without branches (with one
        // entry per step and entry_less's short-circuit the loop was two exposed LDS round trips per entry: 873 us for the
        // 6,554-hit list of the benchmark wave; one thread per target over the whole list, 8 entries per step: 268 us -- the
        // loop is bound by its ~64 compare / select instructions per step, so the longest list is spread over 8x more threads)
        const bool have = i < n;
        const double di = have ? ed[i] : 0.0;
        const int32_t li = have ? el[i] : 0;
        const double2* e2 = reinterpret_cast<const double2*>(ed);
        const int4* l4 = reinterpret_cast<const int4*>(el);
        uint32_t rank = 0;
        auto before = [&](double d, int32_t l) __attribute__((always_inline)) {
            return (uint32_t)((d < di) | ((d == di) & (l < li)));
        };
        const uint32_t per = n8 / kRankSegs;  // entries per segment: a multiple of 8
        for (uint32_t j = seg * per; j < (seg + 1) * per; j += 8) {
            const double2 d0 = e2[j / 2], d1 = e2[j / 2 + 1], d2 = e2[j / 2 + 2], d3 = e2[j / 2 + 3];
            const int4 la = l4[j / 4], lb = l4[j / 4 + 1];
            rank += before(d0.x, la.x) + before(d0.y, la.y) + before(d1.x, la.z) + before(d1.y, la.w) +
                    before(d2.x, lb.x) + before(d2.y, lb.y) + before(d3.x, lb.z) + before(d3.y, lb.w);
        }
        rank += __shfl_xor(rank, 1);  // the 8 partial ranks of a target
        rank += __shfl_xor(rank, 2);
        rank += __shfl_xor(rank, 4);
        if (have && seg == 0 && (int64_t)rank < capacity) {
            out_labels[(int64_t)(q0 + q) * capacity + rank] = li;
            out_dist[(int64_t)(q0 + q) * capacity + rank] = (float)di;
            if (ko.d64) ko.d64[(int64_t)(q0 + q) * capacity + rank] = di;
        }
    }
}

// ------------------------------------------------------------------ launchers
hipError_t launch_filter_prep(const FilterArgs& a, hipStream_t s) {
    const int nkc = a.ld / kFilterChunkK;
    filter_prep_kernel<<<nkc > 0 ? nkc * 16 : 1, 256, 0, s>>>(a);  // nkc == 0: only the per-query state
    return hipGetLastError();
}

hipError_t launch_filter_seed_thr(const FilterArgs& a, const double* seed_d64, int32_t k, hipStream_t s) {
    filter_seed_thr_kernel<<<1, 256, 0, s>>>(a, seed_d64, k);
    return hipGetLastError();
}

hipError_t launch_filter_range_thr(const FilterArgs& a, float radius, hipStream_t s) {
    filter_range_thr_kernel<<<1, 256, 0, s>>>(a, radius);
    return hipGetLastError();
}

// ================================================================== int8 shadow (cosine; MLVDB_I8=0 disables)
// v_mfma_i32_16x16x64_i8 issues at the bf16 instruction's rate under the power cap (tools/probe/mfma_i8_probe):
// half the MFMA instructions, shadow bytes and LDS reads per row.  Rows and queries are quantised with one scale
// per vector, x ~ sx * x8, q^ ~ sq * q8; the integer dot product I is exact, and
//   |<q^,x> - sq sx I| <= (eq8 + (1 + eq8) rmax8) |x|      (Cauchy-Schwarz on the two rounding errors, |q^| <= 1)
// with eq8 = |q^ - sq q8| measured per query and rmax8 = max over rows of |x - sx x8| / |x| measured at build time.
// The errors are ~7x those of bf16, so thresholds from lower bounds (u - 2 eps) would be far too loose:
// filter_refine_thr_kernel sets them from EXACT scores of the k best bounds of every round instead.
__device__ __forceinline__ int64_t layout_offset_i8(int64_t row, int32_t col, int32_t ld) {
    return (row >> 4) * (int64_t)(kPanelRows * ld) + (int64_t)(col >> 6) * 1024 + ((col & 63) >> 4) * 256 + (row & 15) * 16 +
           (col & 15);
}

// One wave per 32-row slab (two 16-row panels): lane 16g + r owns row r of each panel, columns 4g..4g+3 of every
// 16-column group (one coalesced 1 KiB load per group, as everywhere); pass 1 finds the row maxima, pass 2 (the slab is
// in L2 by then) quantises and measures the error.  Whole slabs are (re)written: idempotent for rows converted before.
//
// Scales.  l2: one scale per row, sx = max|x_i| / 127.  ip / cosine: the 8 rows that one lane of the scan holds for a
// query tile -- rows 4g'..4g'+3 of both panels of the slab -- share ONE scale (ip: the group's largest |x_i| / 127) or,
// cosine, ONE ratio sx / (|x| + 1e-30), the largest of their own ratios (so nobody clips; the others quantise ~10 %
// coarser, and every row's error is still measured, not assumed).  The scan's row constant a = sx/(|x|+1e-30) is then the same for the lane's 8 rows, which makes the
// one-compare pre-test of the folded admission test (tools/gen_scan_asm.py, gen_pretest: max_j(I_j) * max_j(a_j) +
// max_j(b_j K) >= T) as sharp as the 8 exact tests it stands for.
__global__ __launch_bounds__(256) void shadow8_rows_kernel(const float* X, const float* rn, int8_t* X8, float* rp8,
                                                           unsigned int* row_err8, int64_t slab_begin, int64_t slab_end,
                                                           int64_t panel_end, int64_t full_slabs, int32_t ld, int32_t ld8,
                                                           int32_t space) {
    const int lane = threadIdx.x & 63;
    const int64_t slab = slab_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slab >= slab_end) return;
    const int g = lane >> 4, r = lane & 15;
    const int ngroups = ld / 16;
    float amax[2] = {0.f, 0.f}, nrm[2], ratio = 0.f, gmax = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int64_t panel = slab * 2 + p;
        nrm[p] = __builtin_nanf("");
        if (panel >= panel_end) continue;  // the capacity is a whole number of panels, not of slabs
        const float4* src = reinterpret_cast<const float4*>(X + panel * (int64_t)(kPanelRows * ld) + lane_group_offset(lane));
        float m = 0.f;
        for (int cg = 0; cg < ngroups; ++cg) {
            const float4 v = src[cg * (kGroupFloats / 4)];
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))),
                                __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
        }
        m = __builtin_fmaxf(m, __shfl_xor(m, 16));
        m = __builtin_fmaxf(m, __shfl_xor(m, 32));
        amax[p] = m;
        nrm[p] = rn[panel * kPanelRows + r];  // NaN: tombstoned / not a row
        if (nrm[p] == nrm[p]) {
            ratio = __builtin_fmaxf(ratio, m / (nrm[p] + 1e-30f));
            gmax = __builtin_fmaxf(gmax, m);
        }
    }
    // the group's ratio (cosine) / largest component (l2, ip): rows r with the same r >> 2, both panels
    ratio = __builtin_fmaxf(ratio, __shfl_xor(ratio, 1));
    ratio = __builtin_fmaxf(ratio, __shfl_xor(ratio, 2));
    gmax = __builtin_fmaxf(gmax, __shfl_xor(gmax, 1));
    gmax = __builtin_fmaxf(gmax, __shfl_xor(gmax, 2));
    float bg = 0.f;  // l2: the largest relative error among the group's live rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int64_t panel = slab * 2 + p;
        if (panel >= panel_end) continue;
        float sx = amax[p] > 0.f ? amax[p] / 127.0f : 1.0f;
        if (space == kSpaceCosine && nrm[p] == nrm[p] && ratio > 0.f) {
            const float shared = ratio * (nrm[p] + 1e-30f) * (1.0f / 127.0f) * 1.0000005f;
            if (shared > sx) sx = shared;  // never finer than the row's own scale: nothing clips
        }
        // ip, l2: the group's scale (>= the row's own).  (l2 since round 4: the folded admission test needs one scale per lane)
        if ((space == kSpaceIp || space == kSpaceL2) && nrm[p] == nrm[p] && gmax > amax[p]) sx = gmax / 127.0f;
        const float inv = 1.0f / sx;
        double err2 = 0.0, n2 = 0.0;
        const float4* src = reinterpret_cast<const float4*>(X + panel * (int64_t)(kPanelRows * ld) + lane_group_offset(lane));
        uint32_t* dst = reinterpret_cast<uint32_t*>(X8 + panel * (int64_t)(kPanelRows * ld8)) + r * 4 + g;  // columns ld..ld8 stay zero
        for (int cg = 0; cg < ngroups; ++cg) {
            const float4 v = src[cg * (kGroupFloats / 4)];
            const float x[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = __builtin_rintf(x[i] * inv);
                t = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, t));
                packed |= ((uint32_t)(int)t & 0xffu) << (8 * i);
                const double e = (double)x[i] - (double)sx * (double)t;
                err2 += e * e;
                n2 += (double)x[i] * (double)x[i];
            }
            dst[(cg >> 2) * 256 + (cg & 3) * 64] = packed;  // bytes: (cg>>2)*1024 + (cg&3)*256 + r*16 + 4g (layout_offset_i8)
        }
        err2 += __shfl_xor(err2, 16);
        err2 += __shfl_xor(err2, 32);
        n2 += __shfl_xor(n2, 16);
        n2 += __shfl_xor(n2, 32);
        float rel = 0.f;  // (every lane of the row holds the reduced sums)
        if (n2 > 0.0) {
            rel = (float)(__builtin_sqrt(err2 / n2) * 1.000001);
            rel = __uint_as_float(__float_as_uint(rel) + 1u);
        }
        if (nrm[p] == nrm[p]) bg = __builtin_fmaxf(bg, rel);
        if (g == 0) {
            const int64_t row = panel * kPanelRows + r;
            if (n2 > 0.0) atomicMax(row_err8, __float_as_uint(rel));  // non-negative floats order like their bits
            if (nrm[p] == nrm[p]) atomicMin(row_err8 + 1, __float_as_uint(nrm[p]));  // smallest norm of a row ever held (l2c_delta)
            // per-row pair: cosine {sx/(|x|+1e-30), this row's error}, ip {sx, |x|}; NaN marks a tombstone.  (l2: below)
            if (space != kSpaceL2) {
                float2 pr;
                if (space == kSpaceCosine) pr = make_float2(sx / (nrm[p] + 1e-30f), nrm[p] == nrm[p] ? rel : nrm[p]);
                else pr = make_float2(nrm[p] == nrm[p] ? sx : nrm[p], nrm[p]);
                reinterpret_cast<float2*>(rp8)[row] = pr;
            }
        }
    }
    if (space == kSpaceL2) {
        // l2 pairs (round 4): {x-slot, |x|} with |x| = NaN for a dead row and the x-slot NEVER NaN: the group's scale S in the
        // rows of the slab's first panel, the group's largest relative row error Bg in the rows of its second panel -- a scan
        // lane holds the 8 rows of a group and reads S from its first row, Bg from its fifth.  Both panels of the slab are
        // written even when the second lies beyond the last row (the capacity is a whole number of slabs): all dead there.
        bg = __builtin_fmaxf(bg, __shfl_xor(bg, 1));
        bg = __builtin_fmaxf(bg, __shfl_xor(bg, 2));
        const float sg = gmax > 0.f ? gmax / 127.0f : 1.0f;  // (= the sx every live row of the group was quantised with)
        // odd groups (api.hip: i8_bounds_usable).  Counted once: when the slab is complete -- a partly filled last slab is
        // converted again by the next update (single-row inserts: up to 31 times) and counts then.
        if (g == 0 && (r & 3) == 0 && bg > 0.03f && slab < full_slabs) atomicAdd(row_err8 + 2, 1u);
        if (g == 0) {
            reinterpret_cast<float2*>(rp8)[(slab * 2) * kPanelRows + r] = make_float2(sg, nrm[0]);
            reinterpret_cast<float2*>(rp8)[(slab * 2 + 1) * kPanelRows + r] = make_float2(bg, nrm[1]);
        }
    }
}

// Median machinery for the 256 per-query values of a pass (non-negative floats; negative = not a query).  A value's order key is
// its bit pattern (monotone for non-negative floats) with the query index below it -- distinct 64-bit integers, so the ranks of the
// real queries are a permutation and one u64 compare per pair decides (the float compares with index tie-break compiled to ~50
// instructions per pair: 25 us per launch); non-queries get the largest key and order after everything.
__device__ __forceinline__ unsigned long long rank_key_256(const float v, const int t) {
    return v >= 0.f ? ((unsigned long long)__float_as_uint(v) << 8) | (unsigned long long)t : ~0ull;
}
__device__ __forceinline__ int rank_among_256(const unsigned long long* keys, const unsigned long long mine) {
    const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(keys);  // (16-byte aligned LDS: two keys per read)
    int below = 0;
#pragma unroll 16
    for (int j = 0; j < kFilterQueries / 2; ++j) {
        const ulonglong2 o = k2[j];
        below += o.x < mine ? 1 : 0;
        below += o.y < mine ? 1 : 0;
    }
    return below;
}

// l2, once per pass: e[row] = ceil((p_row - P0) / (SQ S)) + 1 (<= 1), the integer that, added to the row's int8 dot product,
// stands for the difference between the row's own term p = -(1 - slack) |x|^2 and the largest such term P0 among the 8 rows one
// scan lane holds (rows 4g..4g+3 of both panels of a 32-row slab), in units of one quantum SQ S of the score: SQ = the largest
// sq = 2 |q| sq8 of the pass's queries (so the bound holds for every query: sq <= SQ and p - P0 <= 0), S = the lane's scale.
// p, P0, S and SQ are formed by the very float operations the scan uses (tools/gen_scan_asm.py: gen_rowmax_l2ip, the wrapper's
// preamble); dead rows (NaN pairs) drop out of the maxima and get 0.  One thread per row; a slab's 32 rows are 32 adjacent lanes.
__global__ __launch_bounds__(256) void filter_l2_offsets_kernel(const FilterArgs a, const int64_t rows) {
    __shared__ float s_sq[4], s_ke[4], s_kr[4];
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[kFilterQueries];
    __shared__ float s_med;
    const int t = threadIdx.x;  // 256 threads = kFilterQueries
    {
        float sq = t < a.nq ? a.qscale[t] * a.sq8[t] : 0.f;  // (the scan's preamble: sqv * a.sq8[t])
        if (!(sq == sq)) sq = 0.f;
        for (int off = 32; off > 0; off >>= 1) sq = __builtin_fmaxf(sq, __shfl_xor(sq, off));
        if ((t & 63) == 0) s_sq[t >> 6] = sq;
    }
    if (blockIdx.x == 0) {  // (block-uniform) the pass's scalars for the l2c scan bodies; every block needs only SQ
        // Odd queries out.  The coefficients below are maxima over the pass's queries: one query whose image is useless (it clipped,
        // or it is tiny beside the others: eq8 ~ 1) would loosen every query's bounds.  A query whose measured error exceeds 4 x the
        // pass's median (and 0.03) is taken off the filter here -- flagged like a list overflow, threshold +inf so that no scan admits
        // anything for it -- and served by the exact fallback; the maxima run over the others.
        const float e8 = t < a.nq ? a.ke8[t] : -1.f;
        const unsigned long long key = rank_key_256(e8, t);
        s_key[t] = key;
        __syncthreads();
        if (e8 >= 0.f && rank_among_256(s_key, key) == (a.nq - 1) / 2) s_med = e8;  // the (lower) median of the pass's errors
        __syncthreads();
        const bool odd = t < a.nq && e8 > __builtin_fmaxf(0.03f, 4.0f * s_med);
        if (odd) {
            a.overflow[t] = 1u;
            a.thr[t] = 3.4e38f;
        }
        // l2c: one error coefficient for the pass, KE >= sq_q ke'_q = 2 |q| ke_q of every query, with room for the roundings of
        // sq_q against SQ and of S SQ (each <= 2e-7 relative of 2 |q| |x|)
        // per-row-group errors (round 4): the bound's error term is 2 |q| (eq8_q + (1 + eq8_q) Bg) N_j = (KEq + KEr Bg) N_j with
        // KEq >= 2 |q| eq8_q (a.ke8: the query's own measured error + roundings + slack) and KEr >= 2 |q| (1 + eq8_q)
        float ke = t < a.nq && !odd ? float_above((double)a.qscale[t] * ((double)a.ke8[t] * 1.000002 + 1.0e-6)) : 0.f;
        float kr = t < a.nq && !odd ? float_above((double)a.qscale[t] * (1.0 + (double)a.ke8[t]) * 1.000002) : 0.f;
        if (!(ke == ke)) ke = 0.f;
        if (!(kr == kr)) kr = 0.f;
        for (int off = 32; off > 0; off >>= 1) {
            ke = __builtin_fmaxf(ke, __shfl_xor(ke, off));
            kr = __builtin_fmaxf(kr, __shfl_xor(kr, off));
        }
        if ((t & 63) == 0) {
            s_ke[t >> 6] = ke;
            s_kr[t >> 6] = kr;
        }
    }
    __syncthreads();
    const float SQ = __builtin_fmaxf(__builtin_fmaxf(s_sq[0], s_sq[1]), __builtin_fmaxf(s_sq[2], s_sq[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // what the l2c scan bodies take as scalars (the kernel boundary publishes them)
        a.l2c_out[0] = SQ;
        a.l2c_out[1] = __builtin_fmaxf(__builtin_fmaxf(s_ke[0], s_ke[1]), __builtin_fmaxf(s_ke[2], s_ke[3]));
        a.l2c_out[2] = __builtin_fmaxf(__builtin_fmaxf(s_kr[0], s_kr[1]), __builtin_fmaxf(s_kr[2], s_kr[3]));
    }
    // The plane is kept across passes (round 4): it was computed for the scale SQh >= the pass scale it was first needed for, and an
    // offset computed for a LARGER scale stays a valid (slightly looser) bound for a smaller one -- (p - P0) <= 0, so
    // ceil((p - P0) / (SQh S)) >= ceil((p - P0) / (SQ S)).  A pass whose SQ lies within 3e-6 below the tag reuses it (the prep
    // quantises the pass's common step on a 2^(1/4) grid, so batches of similar queries produce the same SQ up to roundings);
    // whatever rewrites the pairs zeroes the tag (api.hip forget_l2_offsets).  Identical decision in every block: the tag is
    // only written by the block that finishes last, after every block has read it.
    const float tag = a.l2tag ? __uint_as_float(a.l2tag[0]) : 0.f;
    const bool hit = a.l2tag && SQ <= tag && SQ >= tag * (1.0f - 3.0e-6f) && a.l2tag[1] == (uint32_t)rows;
    if (hit) return;
    const float SQh = a.l2tag ? float_above((double)SQ * (1.0 + 1.0e-6)) : SQ;
    const float k1 = -(1.0f - kSlack);
    int32_t* eoff = reinterpret_cast<int32_t*>(const_cast<float*>(a.rp8) + 2 * a.rp8_cap);
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (int64_t)gridDim.x * 256) {
        const float2 pr = reinterpret_cast<const float2*>(a.rp8)[row];  // {scale, |x|}, NaN = dead
        float p = pr.y * pr.y;
        p = k1 * p;
        // maxima over the lane group: rows r ^ 1, r ^ 2, r ^ 3 (same panel) and r ^ 16 (the other panel); the group's scale is the
        // x-slot of its first-panel rows (the second panel's holds the group's error: shadow8_rows_kernel)
        float P0 = p, S = (row & 16) ? 0.f : pr.x;
        P0 = __builtin_fmaxf(P0, __shfl_xor(P0, 1));
        P0 = __builtin_fmaxf(P0, __shfl_xor(P0, 2));
        P0 = __builtin_fmaxf(P0, __shfl_xor(P0, 16));
        S = __builtin_fmaxf(S, __shfl_xor(S, 1));
        S = __builtin_fmaxf(S, __shfl_xor(S, 2));
        S = __builtin_fmaxf(S, __shfl_xor(S, 16));
        int32_t e = 0;
        const double quantum = (double)SQh * (double)S;
        if (p == p && P0 == P0 && quantum > 0.0) {
            const double d = __builtin_ceil(((double)p - (double)P0) / quantum);  // <= 0
            e = (int32_t)(d < -1073741824.0 ? -1073741824.0 : d) + 1;
        }
        eoff[row] = e;
    }
    if (a.l2tag) {  // the block that finishes last publishes what the plane now holds
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(&a.l2tag[2], 1u) == gridDim.x - 1u) {
            a.l2tag[2] = 0u;
            a.l2tag[1] = (uint32_t)rows;
            a.l2tag[0] = __float_as_uint(SQh);
        }
    }
}

hipError_t launch_filter_l2_offsets(const FilterArgs& a, int64_t rows, hipStream_t s) {
    if (a.rp8_cap <= 0 || rows <= 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>(2048, (rows + 255) / 256);
    filter_l2_offsets_kernel<<<(unsigned)blocks, 256, 0, s>>>(a, rows);
    return hipGetLastError();
}

hipError_t launch_shadow8_rows(const float* X, const float* rn, void* X8, float* rp8, float* row_err8, int64_t row_begin,
                               int64_t row_end, int32_t ld, int32_t ld8, int32_t space, hipStream_t s) {
    const int64_t pe = (row_end + kPanelRows - 1) / kPanelRows;
    const int64_t sb = row_begin / (2 * kPanelRows), se = (pe + 1) / 2;
    if (se <= sb) return hipSuccess;
    shadow8_rows_kernel<<<(unsigned)((se - sb + 3) / 4), 256, 0, s>>>(X, rn, static_cast<int8_t*>(X8), rp8,
                                                                       reinterpret_cast<unsigned int*>(row_err8), sb, se, pe,
                                                                       row_end / (2 * kPanelRows), ld, ld8, space);
    return hipGetLastError();
}

// Query image: Qimg8[kc][n][ks][lane][j] = q8[16n + (lane&15)][128kc + 64ks + 16(lane>>4) + j], one block per query
// slot; also sq8, and ke = the int8 error term (overrides filter_prep_kernel's, so that every kernel of the pass --
// the bf16 seeding pass included -- uses the same, larger eps).
__global__ __launch_bounds__(256) void filter_prep8_kernel(const FilterArgs a) {
    __shared__ float red[4];
    __shared__ double dred[4];
    const int q = blockIdx.x;
    const int ld = a.ld;
    int8_t* img = reinterpret_cast<int8_t*>(a.qimg8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float inv = 0.f;
    if (q < a.nq) inv = a.space == kSpaceCosine ? (float)a.qaux[q] : (float)(1.0 / (a.qaux[q] + 1e-30));  // as filter_prep_kernel
    float amax = 0.f;
    for (int c = threadIdx.x; c < ld; c += 256) {
        const float v = q < a.nq ? a.Qpad[(int64_t)q * ld + c] * inv : 0.f;
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    const float sq = amax > 0.f ? amax / 127.0f : 1.0f;
    const float isq = 1.0f / sq;
    double err2 = 0.0;
    for (int c = threadIdx.x; c < a.ld8; c += 256) {  // (columns ld..ld8 of the image: zeros, like the shadow's)
        const float v = q < a.nq && c < ld ? a.Qpad[(int64_t)q * ld + c] * inv : 0.f;
        float t = __builtin_rintf(v * isq);
        t = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, t));
        const double e = (double)v - (double)sq * (double)t;
        err2 += e * e;
        const int kc = c >> 7, ks = (c >> 6) & 1, g = (c >> 4) & 3, j = c & 15;
        const int n = q >> 4, l = (q & 15) + 16 * g;
        img[((((int64_t)kc * 16 + n) * 2 + ks) * 64 + l) * 16 + j] = (int8_t)t;
    }
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
    if (lane == 0) dred[wave] = err2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double eq8 = __builtin_sqrt(dred[0] + dred[1] + dred[2] + dred[3]) * 1.000001 + 1e-12;
        a.sq8[q] = sq;
        a.ke8[q] = q < a.nq ? float_above(eq8) : 0.f;  // finished by filter_prep8_fin_kernel
        if (q < a.nq) {
            atomicMin(a.sqmin, __float_as_uint(sq));                    // positive floats order like their bits
            atomicMax(a.sqmin + 1, __float_as_uint(float_above(eq8)));  // the largest query error of the pass
        }
    }
}

// Second phase (one block): the error terms that need the smallest query scale of the pass.
//   cosine  rows carry their own error b: the scan tests  w + b K >= (thr - ke8)/sq8  with K = (1 + max eq8)/min sq8,
//           i.e. the bound u = sq8 w + ke8 + (1 + max eq8) b sq8/min sq8 >= <q^,x>/|x|;  ke8 = eq8 + roundings + slack
//           (|q8 image| <= 1 + eq8: a query with one dominant component quantises its small ones badly, eq8 ~ 0.1)
//   l2, ip  index-wide row error: ke = eq8 + (1 + eq8) rmax8 + roundings + slack (x |x|)
//   keb     the bf16 term (filter_prep_kernel's ke): what the bf16 seeding pass adds to its bounds
//   ke      covers both kinds of entry (the update kernel's lower bounds u - 2 eps): the larger of the two, with the
//           int8 row term at its index-wide maximum
// (device function: also called by filter_prep_fused_kernel for one-query passes; needs a.ke8[q] = eq8 and a.ke[q] = the bf16 term)
__device__ __forceinline__ void prep8_fin_query(const FilterArgs& a, const int q, const float sqmin, const float eqmax_f) {
    const double eqmax = (double)eqmax_f;
    const double eq8 = (double)a.ke8[q];
    const double ratio = a.space == kSpaceCosine && q < a.nq ? (double)a.sq8[q] / (double)sqmin * 1.000001 : 1.0;
    const double rnd = 4.0 * 5.9604644775390625e-08;  // float(I) * rp8 (+ b K) * sq8: roundings of a value <= ~1
    const double slack = (a.space == kSpaceCosine ? 2.0 : 1.0) * (double)kSlack;
    const double e8 = eq8 + (1.0 + (a.space == kSpaceCosine ? eqmax : eq8)) * 1.000001 * (double)*a.row_err8 * ratio + rnd;
    const double eb = (double)a.ke[q];  // bf16 term incl. its slack (filter_prep_kernel)
    a.keb[q] = a.ke[q];
    const double big = (e8 * 1.000001 + slack) > eb ? (e8 * 1.000001 + slack) : eb;
    a.ke[q] = float_above(big);
    a.ke8[q] = float_above((eq8 + rnd) * 1.000001 + slack);
    if (q == 0) a.ke8[kFilterQueries] = float_above((1.0 + eqmax) / (double)sqmin * 1.000001);
}

__global__ __launch_bounds__(256) void filter_prep8_fin_kernel(const FilterArgs a, const int reset) {
    const int q = threadIdx.x;
    const float sqmin = __uint_as_float(a.sqmin[0]);
    const float eqmax = __uint_as_float(a.sqmin[1]);
    if (reset) {  // fused prep: nobody re-initialises the two scalars before the next pass's atomics -- this kernel does, once
        __syncthreads();  // every thread of this (only) block has read them
        if (q == 0) {
            a.sqmin[0] = 0x7f7f7f7fu;
            a.sqmin[1] = 0u;
        }
    }
    prep8_fin_query(a, q, sqmin, eqmax);
}

// ------------------------------------------------------------------ one launch for everything a pass needs of its queries
// query_prep_kernel (padded copy, norm, bf16 rounding error) + filter_prep_kernel (bf16 image, per-query state) +
// filter_prep8_kernel (int8 image, scale, error) as ONE kernel, block q = query q of the pass: three launches and their
// two kernel boundaries less per pass (round 3; ~14 us of a 256-query wave, ~20 us of a batch-1 call).  Same arithmetic,
// value for value, as the three kernels it replaces (they still serve the range path and the exact scans).
__global__ __launch_bounds__(256) void filter_prep_fused_kernel(const FilterArgs a, const float* __restrict__ queries, const int32_t dim,
                                                                float* __restrict__ Qpad, double* __restrict__ qaux,
                                                                float* __restrict__ qerr, const int want_i8) {
    __shared__ double dred[4];
    __shared__ float fred[4];
    __shared__ double s_inv;
    const int q = blockIdx.x;
    const bool real = q < a.nq;
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* dst = Qpad + (int64_t)q * ld;
    // 1. zero-padded copy and norm (query_prep_kernel)
    double s = 0.0;
    if (real) {
        const float* src = queries + (int64_t)q * dim;
        for (int c = threadIdx.x; c < ld; c += 256) {
            const float v = c < dim ? src[c] : 0.f;
            dst[c] = v;
            s = __builtin_fma((double)v, (double)v, s);
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) dred[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nrm = __builtin_sqrt((dred[0] + dred[1]) + (dred[2] + dred[3]));
        const double aux = a.space == kSpaceCosine ? 1.0 / (nrm + 1e-30) : nrm;
        if (real) qaux[q] = aux;
        s_inv = a.space == kSpaceCosine ? aux : 1.0 / (aux + 1e-30);
    }
    __syncthreads();
    const double inv = s_inv;
    const float invf = (float)inv;
    // 2. bf16 image in B-fragment order (filter_prep_kernel) and its measured rounding error (query_prep_kernel)
    __bf16* img = reinterpret_cast<__bf16*>(a.qimg);
    const bool want_bf16 = ld % kFilterChunkK == 0;  // (another ld has no bf16 / fp32 filter body: only the int8 image is used)
    const int n16 = q >> 4, c16 = q & 15;
    double e2 = 0.0;
    float amax = 0.f;
    for (int c = threadIdx.x; c < ld; c += 256) {
        const float x = real ? dst[c] : 0.f;  // (this thread wrote dst[c] itself)
        const float v = x * invf;
        const __bf16 b = (__bf16)v;
        int kc = c >> 6, ks, g, j;
        if (a.Xb) {  // shadow k order: 8 consecutive columns per lane group
            ks = (c >> 5) & 1; g = (c >> 3) & 3; j = c & 7;
        } else {     // fp32 panels: columns {4g..4g+3} of two 16-column groups
            const int t = (c >> 4) & 3;
            ks = t >> 1; g = (c >> 2) & 3; j = (t & 1) * 4 + (c & 3);
        }
        if (want_bf16) img[((((int64_t)kc * 16 + n16) * 2 + ks) * 64 + (c16 + 16 * g)) * 8 + j] = b;
        if (c < dim) {
            const double e = (double)x * inv - (double)(float)(__bf16)(x * invf);
            e2 = __builtin_fma(e, e, e2);
        }
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) {
        e2 += __shfl_xor(e2, off);
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, off));
    }
    __syncthreads();  // (dred is reused)
    if (lane == 0) {
        dred[wave] = e2;
        fred[wave] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float e = (float)(__builtin_sqrt((dred[0] + dred[1]) + (dred[2] + dred[3])) * 1.000001);
        e = __uint_as_float(__float_as_uint(e) + 1u);  // never below the true error
        if (real) qerr[q] = e;
        // per-query state of the pass (filter_prep_kernel, block 0)
        const double nrm = real ? (a.space == kSpaceCosine ? 0.0 : qaux[q]) : 0.0;
        a.qscale[q] = a.space == kSpaceL2 ? (float)(2.0 * nrm) : 1.0f;
        const double e1q = (real ? (double)e : 0.0) + 1.00390625 * (double)*a.row_err + (double)ld * 2.384185791015625e-07;
        const float ke = (float)(e1q * 1.000001) + (a.space == kSpaceCosine ? 2.0f : 1.0f) * kSlack;
        a.ke[q] = __uint_as_float(__float_as_uint(ke) + 1u);
        a.thr[q] = real ? -3.0e38f : 3.4e38f;  // padded queries never admit anything
        a.cnt[q] = 0;
        a.overflow[q] = 0;
    }
    if (!want_i8) return;
    amax = __builtin_fmaxf(__builtin_fmaxf(fred[0], fred[1]), __builtin_fmaxf(fred[2], fred[3]));
    if (a.l2c) {
        // l2 with one quantisation step for the whole pass (round 4: filter_prep8_l2c_kernel builds the images once the pass's
        // largest raw component is known): here only this query's largest |q_i| = max |q^_i| x |q|, rounded up
        if (threadIdx.x == 0) a.rmaxq[q] = real ? float_above((double)amax * (qaux[q] + 1e-30) * 1.000001) : 0.f;
        return;
    }
    // 3. int8 image, scale and error (filter_prep8_kernel); a.sqmin[] was left initialised by the previous pass's fin kernel
    const float sq = amax > 0.f ? amax / 127.0f : 1.0f;
    const float isq = 1.0f / sq;
    int8_t* img8 = reinterpret_cast<int8_t*>(a.qimg8);
    double err2 = 0.0;
    for (int c = threadIdx.x; c < a.ld8; c += 256) {  // (columns ld..ld8 of the image: zeros, like the shadow's)
        const float v = real && c < ld ? dst[c] * invf : 0.f;
        float t = __builtin_rintf(v * isq);
        t = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, t));
        const double e = (double)v - (double)sq * (double)t;
        err2 += e * e;
        const int kc = c >> 7, ks = (c >> 6) & 1, g = (c >> 4) & 3, j = c & 15;
        img8[((((int64_t)kc * 16 + n16) * 2 + ks) * 64 + (c16 + 16 * g)) * 16 + j] = (int8_t)t;
    }
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
    __syncthreads();
    if (lane == 0) dred[wave] = err2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double eq8 = __builtin_sqrt(dred[0] + dred[1] + dred[2] + dred[3]) * 1.000001 + 1e-12;
        a.sq8[q] = sq;
        a.ke8[q] = real ? float_above(eq8) : 0.f;  // finished by filter_prep8_fin_kernel
        if (real && a.nq > 1) {
            atomicMin(a.sqmin, __float_as_uint(sq));
            atomicMax(a.sqmin + 1, __float_as_uint(float_above(eq8)));
        }
        if (real && a.nq == 1) {
            // a one-query pass (BASELINE configs[1]): the smallest scale / largest error "of the pass" are this query's own,
            // so the fin kernel's arithmetic for it runs here and its launch is saved (prep8_fin_query: one formula, two callers)
            prep8_fin_query(a, 0, sq, float_above(eq8));
        }
    }
    if (want_i8 && a.nq == 1 && q > 0 && threadIdx.x == 0) prep8_fin_query(a, q, 1.0f, 0.0f);  // padded slots (never admitted)
}

// l2, common query scale (a.l2c = 2 + parity of the pass; round 4).  The l2 score in the scan's units is
//   s = 2 |q| <q^, x> - |x|^2 ~ sq (S I) - |x|^2,   sq = 2 |q| sq8   (sq8 = the scale of the query's int8 image),
// and the folded admission test (tools/gen_scan_asm.py, l2c) wants ONE sq for every query of the pass: then the per-row
// integer offsets are exact for every query and the test needs one per-query constant (the threshold), like cosine's.  So
// the images are built with sq8_q = SQ / (2 |q|), SQ = 2 QMAX / 127 and QMAX = the largest |q_i| of the pass's typical raw
// queries (a.rmaxq, left by the fused kernel's blocks; see below): every query is quantised with the same absolute step SQ / 2.
// A query whose own largest component is smaller uses fewer of the 255 levels, one whose is larger clips; its measured error
// eq8 (as always: measured, rounded up) says so.
// Block q: image, scale, error and the per-query error terms (prep8_fin_query: l2 needs nothing of the other queries).
__global__ __launch_bounds__(256) void filter_prep8_l2c_kernel(const FilterArgs a) {
    __shared__ double dred[4];
    const int q = blockIdx.x;
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool real = q < a.nq;
    // The pass's common step follows the largest raw component of its TYPICAL queries: every block finds, from the 256 per-query
    // maxima the fused kernel left in a.sq8, the median m of the real queries' and takes QMAX = the largest maximum <= 4 m.  One
    // query 100x the others would otherwise leave the other 255 with a handful of levels each; this way only the odd one clips
    // (its measured error eq8 says so: looser bounds, at worst its own exact fallback).  Identical in every block: no atomics.
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[kFilterQueries];
    __shared__ float s_q[2];
    const float rmax_t = (int)threadIdx.x < a.nq ? a.rmaxq[threadIdx.x] : -1.f;  // 256 threads = kFilterQueries; -1: not a query
    {
        const unsigned long long key = rank_key_256(rmax_t, (int)threadIdx.x);
        s_key[threadIdx.x] = key;
        __syncthreads();
        // real queries ordered before this one: rank == (nq - 1) / 2 is the (lower) median
        if (rmax_t >= 0.f && rank_among_256(s_key, key) == (a.nq - 1) / 2) s_q[0] = rmax_t;
    }
    __syncthreads();
    {
        const float cap = s_q[0] > 0.f ? 4.0f * s_q[0] : 3.0e38f;
        float v = rmax_t;
        v = (v >= 0.f && v <= cap) ? v : 0.f;
        for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off));
        if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = (double)v;
    }
    __syncthreads();
    float qmax = (float)__builtin_fmax(__builtin_fmax(dred[0], dred[1]), __builtin_fmax(dred[2], dred[3]));
    __syncthreads();  // (dred is reused below)
    // ... rounded up to a 2^(1/4) grid: batches of similar queries then share one step (<= 19 % coarser), and with it the l2
    // offsets plane of the previous pass (filter_l2_offsets_kernel).  Same instructions on the same input in every block.
    if (qmax > 0.f && qmax < 1.0e30f) {
        float g = exp2f(ceilf(log2f(qmax) * 4.0f) * 0.25f);
        if (g < qmax) g *= 1.18920712f;
        if (g >= qmax && g <= qmax * 1.5f) qmax = g;  // (anything odd in the float functions: keep the exact maximum)
    }
    const float SQ = qmax > 0.f ? float_above((double)qmax * (2.0 / 127.0) * 1.000001) : 1.0f;
    const double nrm = real ? a.qaux[q] : 0.0;  // l2: qaux = |q|
    const float invf = (float)(1.0 / (nrm + 1e-30));  // as filter_prep_kernel forms q^ = q / (|q| + 1e-30)
    const float sq = nrm > 0.0 ? (float)((double)SQ / (2.0 * nrm)) : 1.0f;
    const float isq = 1.0f / sq;
    int8_t* img8 = reinterpret_cast<int8_t*>(a.qimg8);
    const int n16 = q >> 4, c16 = q & 15;
    double err2 = 0.0;
    for (int c = threadIdx.x; c < a.ld8; c += 256) {  // (columns ld..ld8 of the image: zeros, like the shadow's)
        const float v = real && c < ld ? a.Qpad[(int64_t)q * ld + c] * invf : 0.f;
        float t = __builtin_rintf(v * isq);
        t = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, t));
        const double e = (double)v - (double)sq * (double)t;
        err2 += e * e;
        const int kc = c >> 7, ks = (c >> 6) & 1, g = (c >> 4) & 3, j = c & 15;
        img8[((((int64_t)kc * 16 + n16) * 2 + ks) * 64 + (c16 + 16 * g)) * 16 + j] = (int8_t)t;
    }
    for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
    if (lane == 0) dred[wave] = err2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double eq8 = __builtin_sqrt(dred[0] + dred[1] + dred[2] + dred[3]) * 1.000001 + 1e-12;
        a.sq8[q] = sq;
        a.ke8[q] = real ? float_above(eq8) : 0.f;
        prep8_fin_query(a, q, 1.0f, 0.0f);  // (l2: the query's own error only; a.ke[q] = the bf16 term the fused kernel left)
    }
}

hipError_t launch_filter_prep_fused(const FilterArgs& a, const float* queries, int32_t dim, float* Qpad, double* qaux, float* qerr,
                                    hipStream_t s) {
    const int want_i8 = a.X8 != nullptr;
    if (!want_i8 && !filter_supported(a.ld)) return hipErrorInvalidValue;
    filter_prep_fused_kernel<<<kFilterQueries, 256, 0, s>>>(a, queries, dim, Qpad, qaux, qerr, want_i8);
    if (want_i8 && a.l2c) filter_prep8_l2c_kernel<<<kFilterQueries, 256, 0, s>>>(a);
    else if (want_i8 && a.nq > 1) filter_prep8_fin_kernel<<<1, kFilterQueries, 0, s>>>(a, 1);  // (one query: done inside the kernel above)
    return hipGetLastError();
}

hipError_t launch_filter_prep8(const FilterArgs& a, hipStream_t s) {
    // a.sqmin[0..1] were initialised by filter_prep_kernel (always launched first: launch_filter_prep)
    filter_prep8_kernel<<<kFilterQueries, 256, 0, s>>>(a);
    filter_prep8_fin_kernel<<<1, kFilterQueries, 0, s>>>(a, 1);
    return hipGetLastError();
}

// Radix select for the refine kernel (256 threads): a key T such that the non-zero keys[0..cnt) that are >= T number at
// least `want` and at most `cap` -- the caller takes ALL of them, so no ties have to be split and one or two histogram
// passes usually do (the digits start at the highest bit in which two keys differ: see filter_update_kernel).  Only when
// more than cap - want + 1 keys are exactly equal at the boundary does the search run to the last bit; then *ties is how
// many of the entries with key == T belong to the selection (the caller takes those with the lowest list indices), else
// *ties = 0xffffffff (take every key >= T).  *total = size of the selection.  kmin / kmax: smallest / largest non-zero
// key (block-uniform).  hist: [256], s_sel: [3].  Needs want <= #non-zero keys.
__device__ __forceinline__ uint32_t radix_select_atleast256(const uint32_t* keys, uint32_t cnt, uint32_t want, uint32_t cap,
                                                            uint32_t kmin, uint32_t kmax, uint32_t* hist, uint32_t* s_sel,
                                                            uint32_t* ties, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t want0 = want;
    const uint32_t diff = kmin ^ kmax;
    uint32_t prefix = kmax, mask = 0xffffffffu;
    if (diff != 0) {
        const int hb = 31 - __builtin_clz(diff);
        mask = hb == 31 ? 0u : ~((2u << hb) - 1u);
        prefix = kmax & mask;
        int shift = hb >= 7 ? hb - 7 : 0, width = hb - shift + 1;
        for (;;) {
            __syncthreads();
            hist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t dmask = (1u << width) - 1u;
            for (uint32_t idx = threadIdx.x; idx < cnt; idx += 256) {
                const uint32_t key = keys[idx];
                if (key != 0 && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
            }
            __syncthreads();
            if (wave == 0) {  // suffix sums over the 256 bins: lane owns bins 4*lane .. 4*lane+3
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                uint32_t above = h0 + h1 + h2 + h3;
                uint32_t run = above;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_down(run, off);
                    if (lane + off < 64) run += v;
                }
                above = run - above;  // keys in bins > 4*lane+3
                const uint32_t c3 = above + h3, c2 = c3 + h2, c1 = c2 + h1, c0 = c1 + h0;
                if (above < want && c0 >= want) {  // the highest bin b with (count in bins >= b) >= want
                    if (c3 >= want) { s_sel[0] = 4 * lane + 3; s_sel[1] = above; s_sel[2] = h3; }
                    else if (c2 >= want) { s_sel[0] = 4 * lane + 2; s_sel[1] = c3; s_sel[2] = h2; }
                    else if (c1 >= want) { s_sel[0] = 4 * lane + 1; s_sel[1] = c2; s_sel[2] = h1; }
                    else { s_sel[0] = 4 * lane; s_sel[1] = c1; s_sel[2] = h0; }
                }
            }
            __syncthreads();
            prefix |= s_sel[0] << shift;
            mask |= dmask << shift;
            want -= s_sel[1];
            // everything in the chosen bin and above: (want0 - want) strictly above the bin + the bin's own s_sel[2]
            if (want0 - want + s_sel[2] <= cap) {
                *ties = 0xffffffffu;
                *total = want0 - want + s_sel[2];
                return prefix;  // the bin's lowest possible key: its undecided low bits are zero
            }
            if (shift == 0) break;
            const int next = shift >= 8 ? shift - 8 : 0;
            width = shift - next;
            shift = next;
        }
        *ties = want;  // more equal keys at the boundary than fit: `want` of them
        *total = want0;
        return prefix;
    }
    *ties = want;  // all keys equal (and more of them than cap: the caller takes short lists whole)
    *total = want0;
    return prefix;
}

// thr[q] from the exact distances of a prefix of the rows (small batches: api.hip, launch_prefix_exact): the threshold the
// k-th nearest of them gives, exactly as filter_seed_thr_kernel forms it -- t(d) = float_below(s(d) - 1e-9 mag) is monotone
// in d, so the k-th largest t is t(k-th smallest d).  Selection without histograms (a radix select run to the last bit
// took 14 us here, per-wave sorted lists with fp64 shuffles 34): every thread keeps its <= 15 keys in registers; L = the
// k-th largest of the 256 per-thread maxima is at most the k-th largest key (k keys are >= L), the few keys >= L (about k,
// unless many are equal) are collected and ranked by counting.
template <int kPer>  // keys per thread: 15 (m <= kSeedRows) or 45 (m <= 3 kSeedRows: the one-round path of small corpora)
__global__ __launch_bounds__(256) void filter_prefix_thr_kernel(const FilterArgs a, const double* __restrict__ d64, const int32_t m,
                                                                const int32_t k) {
    __shared__ __attribute__((aligned(16))) uint32_t tmax[256];
    __shared__ uint32_t sel[256];
    __shared__ uint32_t s_n[4];
    const int q = blockIdx.x;
    const double aux = a.qaux[q];
    if (a.l2c && a.overflow[q]) return;  // (l2: taken off the filter by the pass, its threshold stays +inf; others: no flag is set yet)
    if (a.space == kSpaceIp && !(aux > 0.0)) return;  // |q| = 0: every distance is 1 (block-uniform)
    uint32_t key[kPer], best = 0;
    double dk[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = j * 256 + threadIdx.x;
        dk[j] = i < m ? d64[(int64_t)q * m + i] : __builtin_inf();
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        key[j] = 0;
        if (dk[j] < 1.0e300) {
            double s;
            if (a.space == kSpaceCosine) s = 1.0 - dk[j];
            else if (a.space == kSpaceIp) s = (1.0 - dk[j]) / aux;
            else s = aux * aux - dk[j];
            const double mag = a.space == kSpaceL2 ? aux * aux + __builtin_fabs(dk[j]) : __builtin_fabs(s) + 1.0;
            key[j] = float_order_key(float_below(s - 1e-9 * mag));
            if (key[j] == 0) key[j] = 1;
            best = max(best, key[j]);
        }
    }
    tmax[threadIdx.x] = best;
    if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
    __syncthreads();
    // rank of this thread's maximum among the 256 (ties by thread index: a permutation); the k-th largest is L
    if (k <= 256) {  // (four maxima per LDS read, eight reads in flight: one read per step made this loop most of the kernel)
        uint32_t rank = 0;
        const uint4* t4 = reinterpret_cast<const uint4*>(tmax);
        const uint32_t me = threadIdx.x;
#pragma unroll 8
        for (uint32_t j = 0; j < 64; ++j) {
            const uint4 o = t4[j];
            rank += (o.x > best || (o.x == best && 4 * j < me)) ? 1u : 0u;
            rank += (o.y > best || (o.y == best && 4 * j + 1 < me)) ? 1u : 0u;
            rank += (o.z > best || (o.z == best && 4 * j + 2 < me)) ? 1u : 0u;
            rank += (o.w > best || (o.w == best && 4 * j + 3 < me)) ? 1u : 0u;
        }
        if (rank == (uint32_t)k - 1u) s_n[1] = best;
    }
    __syncthreads();
    const uint32_t L = s_n[1];
    if (L == 0) return;  // fewer than k threads hold a live row: leave the threshold open (rare: the rounds' refines set it)
#pragma unroll
    for (int j = 0; j < kPer; ++j)
        if (key[j] >= L) {
            const uint32_t pos = atomicAdd(&s_n[0], 1u);
            if (pos < 256u) sel[pos] = key[j];
        }
    __syncthreads();
    const uint32_t n = s_n[0];
    uint32_t tkey = L;  // more than 256 keys at or above L (masses of equal rows): L itself is a valid threshold
    if (n <= 256u) {
        if (threadIdx.x < n) {
            const uint32_t mine = sel[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t o = sel[j];
                rank += (o > mine || (o == mine && j < threadIdx.x)) ? 1u : 0u;
            }
            if (rank == (uint32_t)k - 1u) s_n[2] = mine;
        }
        __syncthreads();
        tkey = s_n[2];
    }
    if (threadIdx.x == 0) a.thr[q] = float_from_order_key(tkey);
}

hipError_t launch_filter_prefix_thr(const FilterArgs& a, const double* d64, int32_t m, int32_t k, hipStream_t s) {
    if (m < 1 || m > 3 * kSeedRows || k < 1 || k > 256) return hipErrorInvalidValue;
    if (m <= kSeedRows) filter_prefix_thr_kernel<kSeedRows / 256><<<a.nq, 256, 0, s>>>(a, d64, m, k);
    else filter_prefix_thr_kernel<3 * kSeedRows / 256><<<a.nq, 256, 0, s>>>(a, d64, m, k);
    return hipGetLastError();
}

// Exact thresholds: the entries with the largest bounds -- at least `picks` (>= k), at most max(picks, 32) of them -- are
// scored exactly (fp64); the k-th largest of those exact scores is a lower bound of the final k-th best score, however
// loose the bounds are (k rows are known to score at least that).  With int8 bounds (error ~ one sigma of the score
// distribution) the k largest BOUNDS are not the k best rows; a few more picks than k put most of the true top k among
// them, and the threshold moves up to (nearly) the exact k-th best of the rows seen so far.  Picks: one or two histogram
// passes find a bound with 16..32 entries at or above it, all of which are taken (the set is deterministic, no ties are
// split); gather: 8 lanes per row, all of a row's pieces in flight at once (a row is one page walk: latency, not bytes).
// The sums here are only compared against bounds (1e-9 relative slack below), so their order is free.
//
// FINISH (batches of <= 8 queries, after the last scan round; fuse must be set): the block goes on to do what the rescoring
// kernels do for its query -- exact distances of the surviving rows by accumulate_rows (the arithmetic of every other exact
// path), ranking by (distance, label), the answer written out -- and block 0 compacts the overflow flags for the device-
// decided fallback, as the ranking kernel does.  Three dependent launches (refine 13 + score 12 + rank 5 us at batch 1, of
// which ~4.5 us each are the launch itself) become one.
struct FinishOut {
    int32_t q0;
    int64_t* labels;
    float* dist;
    int32_t* counts;
    double* d64;
    unsigned long long* rescored;
    int32_t* qsel;
    int32_t* nflag;
};

template <int SPACE, bool FINISH = false>
__global__ __launch_bounds__(256) void filter_refine_thr_kernel(const FilterArgs a, const int32_t k, const int32_t forced_cnt,
                                                                const bool fuse, const int32_t picks, const FinishOut fo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);                                   // [ld]
    double* sc = qs + a.ld;                                                         // [64] exact scores of the picks
    double* smin = sc + 64;                                                         // [1]
    int32_t* pick = reinterpret_cast<int32_t*>(smin + 1);                           // [64] rows picked
    uint32_t* hist = reinterpret_cast<uint32_t*>(pick + 64);                        // [256]
    uint32_t* s_scan = hist + 256;                                                  // [24]
    uint32_t* keys = s_scan + 24;                                                   // [kCandCap]
    CandEntry* stage = reinterpret_cast<CandEntry*>(keys + kCandCap);               // [kCandCap] (fuse only)
    const int q = blockIdx.x;
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: phase stamps (100 MHz) of every block, read back by api.hip (MLVDB_DEBUG_REFINE)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.wgbuf) + (size_t)blockIdx.x * 8;
#define REFINE_STAMP(i) do { if (threadIdx.x == 0 && a.wgbuf) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define REFINE_STAMP(i) do { } while (0)
#endif
    REFINE_STAMP(0);
    if (FINISH && blockIdx.x == 0 && fo.qsel) {  // block-uniform; 256 threads = kFilterQueries
        uint32_t* s_flagged = s_scan + 18;  // [4] (no static LDS here: the dynamic allocation is sized to the CU's limit)
        const int t = threadIdx.x;
        // (a list longer than its capacity without the flag cannot happen; the block of that query would set the flag below)
        const bool flagged = t < a.nq && (a.overflow[t] != 0 || a.cnt[t] > (uint32_t)kCandCap);
        const unsigned long long bal = __ballot(flagged);
        if ((t & 63) == 0) s_flagged[t >> 6] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (t >> 6); ++w) base += (int)s_flagged[w];
        if (flagged) fo.qsel[base + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = t;
        if (t == 0) {
            const int n = (int)(s_flagged[0] + s_flagged[1] + s_flagged[2] + s_flagged[3]);
            *fo.nflag = n;
            if (fo.rescored) fo.rescored[1] += (unsigned long long)n;
        }
    }
    if (q >= a.nq || a.overflow[q]) return;
    const uint32_t cnt = forced_cnt >= 0 ? (uint32_t)forced_cnt : a.cnt[q];
    if (cnt > (uint32_t)kCandCap) {  // cannot happen without the flag, but never index past the list
        if (fuse && threadIdx.x == 0) a.overflow[q] = 1u;
        return;
    }
    bool ok = cnt >= (uint32_t)k && k <= 64;  // block-uniform: enough entries to take an exact threshold from
    if (!ok && !fuse) return;
    const int ld = a.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CandEntry* list = a.cand + (int64_t)q * kCandCap;
    for (int c = threadIdx.x; c < ld; c += 256) qs[c] = (double)a.Qpad[(int64_t)q * ld + c];
    // order keys of the bounds into LDS; 0 = not a candidate (NaN bound: tombstoned / padding row).  Eight entries per
    // thread are in flight at once: one load per iteration made this loop 15 dependent round trips on a seed list.
    uint32_t n_valid = 0, kmin = 0xffffffffu, kmax = 0;
    for (uint32_t base = 0; base < cnt; base += 256 * 8) {
        CandEntry e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t idx = base + j * 256 + threadIdx.x;
            e[j] = list[idx < cnt ? idx : cnt - 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t idx = base + j * 256 + threadIdx.x;
            if (idx >= cnt) continue;
            if (fuse) stage[idx] = e[j];
            uint32_t key = 0;
            if (e[j].u == e[j].u) {
                key = float_order_key(e[j].u);
                if (key == 0) key = 1;
                ++n_valid;
                kmin = min(kmin, key);
                kmax = max(kmax, key);
            }
            keys[idx] = key;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_valid += __shfl_xor(n_valid, off);
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, off));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off));
    }
    if (lane == 0) {
        s_scan[wave] = n_valid;
        s_scan[4 + wave] = kmin;
        s_scan[8 + wave] = kmax;
    }
    __syncthreads();
    n_valid = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    kmin = min(min(s_scan[4], s_scan[5]), min(s_scan[6], s_scan[7]));
    kmax = max(max(s_scan[8], s_scan[9]), max(s_scan[10], s_scan[11]));
    if (n_valid < (uint32_t)k) ok = false;  // fewer than k valid entries (block-uniform)
    REFINE_STAMP(1);
    uint32_t m = 0;  // rows picked
    if (ok) {
        // at least `want` and at most `cap` rows (32 = one gather step): every entry whose key is >= tkey
        const uint32_t want = min((uint32_t)max(picks, k), 64u), cap = max(want, 32u);
        uint32_t tkey = 1, ties = 0xffffffffu;
        m = n_valid;  // short lists whole
        if (n_valid > cap) tkey = radix_select_atleast256(keys, cnt, want, cap, kmin, kmax, hist, s_scan + 12, &ties, &m);
        if (threadIdx.x == 0) s_scan[16] = 0;
        __syncthreads();
        const bool split = ties != 0xffffffffu;  // more equal keys at the boundary than fit (block-uniform)
        for (uint32_t idx = threadIdx.x; idx < cnt; idx += 256) {
            const uint32_t key = keys[idx];
            if (key != 0 && (split ? key > tkey : key >= tkey)) {
                const uint32_t pos = atomicAdd(&s_scan[16], 1u);
                if (pos < 64u) pick[pos] = fuse ? stage[idx].row : list[idx].row;  // (pos < m by construction: never past pick[])
            }
        }
        if (split && wave == 0) {  // ... the `ties` entries with that key and the lowest list indices (rare: duplicates)
            uint32_t taken = 0;
            const uint32_t above = m - ties;
            for (uint32_t base = 0; base < cnt && taken < ties; base += 64) {
                const uint32_t idx = base + lane;
                const bool hit = idx < cnt && keys[idx] == tkey;
                const unsigned long long bal = __ballot(hit);
                const uint32_t pos = taken + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (hit && pos < ties && above + pos < 64u) pick[above + pos] = fuse ? stage[idx].row : list[idx].row;
                taken += (uint32_t)__popcll(bal);
            }
        }
    }
    __syncthreads();
    if (ok) m = min(m, 64u);  // (the selection's own count; the gather below never reads past pick[])
    REFINE_STAMP(2);
    // exact scores of the picked rows: 32 rows per step, lane l8 of a row takes the 16-column groups l8, l8 + 8, ...
    const int rslot = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const double qinv = a.qaux[q];
    const int ngroups = ld >> 4;
    constexpr int G = 6;  // groups in flight per lane (d = 768: all of them)
    for (uint32_t p0 = 0; p0 < m; p0 += 32) {
        const uint32_t pidx = p0 + rslot;
        const bool have = pidx < m;
        const int32_t row = have ? pick[pidx] : pick[0];
        const float* rb = a.X + (int64_t)(row >> 4) * (kPanelRows * ld) + (row & 15) * 16;
        double acc = 0.0, nx = 0.0;
        for (int c0 = l8; c0 < ngroups; c0 += 8 * G) {
            float4 v[G][4];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int cg = c0 + 8 * j < ngroups ? c0 + 8 * j : c0;  // clamped: the loads stay unconditional
                const float4* src = reinterpret_cast<const float4*>(rb + (int64_t)cg * kGroupFloats);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[j][i] = src[i];
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (c0 + 8 * j >= ngroups) continue;
                const double* qp = qs + (c0 + 8 * j) * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double x[4] = {(double)v[j][i].x, (double)v[j][i].y, (double)v[j][i].z, (double)v[j][i].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (SPACE == kSpaceL2) {
                            const double e = qp[4 * i + t] - x[t];
                            acc = __builtin_fma(e, e, acc);
                        } else {
                            acc = __builtin_fma(qp[4 * i + t], x[t], acc);
                        }
                        if (SPACE == kSpaceCosine) nx = __builtin_fma(x[t], x[t], nx);
                    }
                }
            }
        }
        for (int off = 1; off < 8; off <<= 1) {
            acc += __shfl_xor(acc, off);
            if (SPACE == kSpaceCosine) nx += __shfl_xor(nx, off);
        }
        const double dk = finish_distance<SPACE>(acc, nx, qinv);
        double s;
        if (SPACE == kSpaceCosine) s = 1.0 - dk;
        else if (SPACE == kSpaceIp) s = (1.0 - dk) / qinv;  // qaux = |q| for ip / l2
        else s = qinv * qinv - dk;
        if (have && l8 == 0) sc[pidx] = s == s ? s : -__builtin_inf();
    }
    if (threadIdx.x == 0) *smin = -__builtin_inf();
    __syncthreads();
    REFINE_STAMP(3);
    if (ok && threadIdx.x < m) {  // the k-th largest of the m exact scores (ranks are a permutation: one thread writes)
        const double si = sc[threadIdx.x];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const double sj = sc[j];
            rank += (sj > si || (sj == si && j < threadIdx.x)) ? 1u : 0u;
        }
        if (rank == (uint32_t)k - 1u) *smin = si;
    }
    __syncthreads();
    float thr = a.thr[q];
    if (ok) {
        const double sm = *smin;
        if (sm > -1.0e300 && sm < 1.0e300) {
            const double mag = SPACE == kSpaceL2 ? qinv * qinv + __builtin_fabs(sm) : __builtin_fabs(sm) + 1.0;
            const float t = float_below(sm - 1e-9 * mag);
            if (t > thr) thr = t;
        }
    }
    REFINE_STAMP(4);
    if (!fuse) {
        if (threadIdx.x == 0) a.thr[q] = thr;
        return;
    }
    // fused threshold update (what filter_update_kernel does after its own, bound-derived threshold): survivors
    // (u >= thr; NaN bounds drop out) are compacted through LDS in list order
    // (wave w owns the contiguous quarter [w per, w per + per) of the list, 64 entries per step: conflict-free LDS reads)
    const uint32_t per = ((cnt + 255u) / 256u) * 64u;
    const uint32_t wb = min(cnt, (uint32_t)wave * per), we = min(cnt, wb + per);
    uint32_t keep = 0;
    for (uint32_t base = wb; base < we; base += 64) {
        const uint32_t idx = base + lane;
        keep += (uint32_t)__popcll(__ballot(idx < we && stage[idx].u >= thr));
    }
    if (lane == 0) s_scan[wave] = keep;
    __syncthreads();
    uint32_t pos = 0;
    for (int w = 0; w < wave; ++w) pos += s_scan[w];
    const uint32_t new_cnt = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    uint32_t* surv = keys;  // FINISH: rows of the survivors, in list order (the keys are done with)
    if (FINISH || new_cnt != cnt) {
        for (uint32_t base = wb; base < we; base += 64) {
            const uint32_t idx = base + lane;
            const bool kept = idx < we && stage[idx].u >= thr;
            const unsigned long long bal = __ballot(kept);
            const uint32_t at = pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (kept) {
                if (FINISH) surv[at] = (uint32_t)stage[idx].row;
                else list[at] = stage[idx];
            }
            pos += (uint32_t)__popcll(bal);
        }
    }
    if (threadIdx.x == 0) {
        a.thr[q] = thr;
        a.cnt[q] = new_cnt;
    }
    REFINE_STAMP(5);
#undef REFINE_STAMP
    if (!FINISH) return;
    __syncthreads();  // surv[] complete; stage[] is free from here on
    // ---- exact distances of the survivors: 16 rows per wave and step (filter_rescore_score_kernel's inner step)
    constexpr int kFinWaves = 4;
    double* ed = reinterpret_cast<double*>(stage);                    // [kRescoreRankMax] short lists: ranked in LDS
    int32_t* el = reinterpret_cast<int32_t*>(ed + kRescoreRankMax);   // [kRescoreRankMax]
    RangeHit* rs = a.rs + (int64_t)q * kCandCap;                      // long lists: through global memory
    const bool in_lds = new_cnt <= (uint32_t)kRescoreRankMax;
    {
        // (two rows per lane group -- 128 rows per block and step, the ~70 survivors in one round of gathers -- was slower:
        // 25.0 vs 22.6 us for the kernel)
        const int g = lane >> 4, r = lane & 15;
        for (uint32_t base = (uint32_t)wave * 16; base < new_cnt; base += kFinWaves * 16) {
            const uint32_t idx = base + r;
            const bool have = idx < new_cnt;
            const int32_t row = (int32_t)surv[have ? idx : base];
            const float* rbase[1] = {a.X + (int64_t)(row >> 4) * (kPanelRows * ld) + (row & 15) * 16 + g * 4};
            double acc[1][1], nx[1];
            accumulate_rows<SPACE, 1, 1, kRescorePF>(rbase, qs, ld, g, acc, nx);
            const double dist = finish_distance<SPACE>(acc[0][0], nx[0], qinv);
            if (have && lane < 16) {
                const float nrm = a.rn[row];
                const bool live = nrm == nrm;
                if (in_lds) {
                    ed[idx] = live ? dist : __builtin_inf();
                    el[idx] = live ? row : kNoLabel;
                } else {
                    RangeHit hit;
                    hit.d = live ? dist : __builtin_inf();
                    hit.l = live ? row : kNoLabel;
                    hit.pad = 0;
                    rs[idx] = hit;
                }
            }
        }
    }
    if (threadIdx.x == 0) s_scan[17] = 0;
    __syncthreads();
    // ---- ranking + output (filter_rescore_rank_kernel's two cases, four waves)
    const int64_t o = (int64_t)(fo.q0 + q) * k;
    if (in_lds) {
        int mine = 0;
        for (uint32_t i = wave; i < new_cnt; i += kFinWaves) {
            const double di = ed[i];
            const int32_t li = el[i];
            if (li == kNoLabel) continue;  // wave-uniform
            ++mine;
            int rank = 0;
            for (uint32_t j0 = 0; j0 < new_cnt && rank < k; j0 += 128) {
                const uint32_t j = j0 + lane, j2 = j + 64;
                const bool l0 = j < new_cnt && entry_less(ed[j], el[j], di, li);
                const bool l1 = j2 < new_cnt && entry_less(ed[j2], el[j2], di, li);
                rank += __popcll(__ballot(l0)) + __popcll(__ballot(l1));
            }
            if (lane == 0 && rank < k) {
                fo.labels[o + rank] = (int64_t)li;
                fo.dist[o + rank] = (float)di;
                if (fo.d64) fo.d64[o + rank] = di;
            }
        }
        if (lane == 0 && mine) atomicAdd(&s_scan[17], (uint32_t)mine);
        __syncthreads();
        const int n_out = min((int)s_scan[17], k);
        for (int i = n_out + threadIdx.x; i < k; i += 256) {
            fo.labels[o + i] = -1;
            fo.dist[o + i] = __builtin_inff();
            if (fo.d64) fo.d64[o + i] = __builtin_inf();
        }
        if (threadIdx.x == 0) {
            fo.counts[fo.q0 + q] = n_out;
            if (fo.rescored) atomicAdd(fo.rescored, (unsigned long long)new_cnt);
        }
        return;
    }
    // long lists (masses of near-equal rows): per-wave sorted top-k lists, merged by wave 0
    double* sd = reinterpret_cast<double*>(stage);                  // [kFinWaves][64]
    int32_t* sl = reinterpret_cast<int32_t*>(sd + kFinWaves * 64);  // [kFinWaves][64]
    WaveTopK top;
    top.init();
    for (uint32_t b = 0; b < new_cnt; b += kFinWaves * 64) {
        const uint32_t i = b + wave * 64 + lane;
        RangeHit hh;
        hh.d = __builtin_inf();
        hh.l = kNoLabel;
        if (i < new_cnt) hh = rs[i];
        top.offer(hh.l != kNoLabel, hh.d, hh.l, k, lane);
    }
    sd[wave * 64 + lane] = top.d;
    sl[wave * 64 + lane] = top.l;
    __syncthreads();
    if (wave != 0) return;
    WaveTopK f;
    f.init();
#pragma unroll
    for (int w2 = 0; w2 < kFinWaves; ++w2)
        f.offer(lane < k && sl[w2 * 64 + lane] != kNoLabel, sd[w2 * 64 + lane], sl[w2 * 64 + lane], k, lane);
    const bool valid = lane < k && f.l != kNoLabel;
    if (lane < k) {
        fo.labels[o + lane] = valid ? (int64_t)f.l : -1;
        fo.dist[o + lane] = valid ? (float)f.d : __builtin_inff();
        if (fo.d64) fo.d64[o + lane] = valid ? f.d : __builtin_inf();
    }
    const int n_valid_out = __popcll(__ballot(valid));
    if (lane == 0) {
        fo.counts[fo.q0 + q] = n_valid_out;
        if (fo.rescored) atomicAdd(fo.rescored, (unsigned long long)new_cnt);
    }
}

// fuse: also do the threshold update's pruning (then filter_update_kernel is not needed for the round); possible while
// the lists' LDS copy fits beside the query (ld <= 2048)
bool filter_refine_can_fuse(const FilterArgs& a) { return a.ld <= 2048; }

hipError_t launch_filter_refine_thr(const FilterArgs& a, int32_t k, int32_t forced_cnt, bool fuse, hipStream_t s) {
    const size_t lds = (size_t)a.ld * sizeof(double) + 65 * sizeof(double) + 64 * 4 + 256 * 4 + 24 * 4 + (size_t)kCandCap * 4 +
                       (fuse ? (size_t)kCandCap * sizeof(CandEntry) : 0);
    auto kern = a.space == kSpaceL2 ? filter_refine_thr_kernel<kSpaceL2>
                : a.space == kSpaceCosine ? filter_refine_thr_kernel<kSpaceCosine> : filter_refine_thr_kernel<kSpaceIp>;
    static std::atomic<uint64_t> configured[3];
    if (hipError_t e = ensure_dynamic_lds(configured[a.space], reinterpret_cast<const void*>(kern), 160 * 1024); e != hipSuccess)
        return e;
    // rows scored exactly per query and round: 1.5 k (at least 16, at most 64); MLVDB_REFINE_PICKS overrides (tuning; = k
    // gives "the smallest exact score of the k largest bounds").  10M x 768, k = 10, per wave: 10 picks 1.962 ms, 16 1.935,
    // 30 1.953, 48 1.976 (profiles/r02/scan_ab_refine_picks_10m.txt): more picks are more page walks per round
    const int pv = a.tn ? a.tn->refine_picks : 0;
    int picks = pv > 0 ? pv : k + k / 2;
    if (pv <= 0 && picks < 16) picks = 16;
    picks = picks < k ? k : (picks > 64 ? 64 : picks);
    kern<<<a.nq, 256, lds, s>>>(a, k, forced_cnt, fuse, picks, FinishOut{});
    return hipGetLastError();
}

// The last refine of a pass of <= 8 queries + rescoring + ranking in one launch (filter_refine_can_fuse(a) must hold, k <= 64).
hipError_t launch_filter_finish_small(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                      int32_t* out_counts, double* out_d64, unsigned long long* rescored, int32_t* qsel,
                                      int32_t* nflag, hipStream_t s) {
    if (!filter_refine_can_fuse(a) || k < 1 || k > 64) return hipErrorInvalidValue;
    const size_t lds = (size_t)a.ld * sizeof(double) + 65 * sizeof(double) + 64 * 4 + 256 * 4 + 24 * 4 + (size_t)kCandCap * 4 +
                       (size_t)kCandCap * sizeof(CandEntry);
    auto kern = a.space == kSpaceL2 ? filter_refine_thr_kernel<kSpaceL2, true>
                : a.space == kSpaceCosine ? filter_refine_thr_kernel<kSpaceCosine, true> : filter_refine_thr_kernel<kSpaceIp, true>;
    static std::atomic<uint64_t> configured[3];
    if (hipError_t e = ensure_dynamic_lds(configured[a.space], reinterpret_cast<const void*>(kern), 160 * 1024); e != hipSuccess)
        return e;
    const int pv = a.tn ? a.tn->refine_picks : 0;
    int picks = pv > 0 ? pv : k + k / 2;
    if (pv <= 0 && picks < 16) picks = 16;
    picks = picks < k ? k : (picks > 64 ? 64 : picks);
    FinishOut fo{q0, out_labels, out_dist, out_counts, out_d64, rescored, qsel, nflag};
    kern<<<a.nq, 256, lds, s>>>(a, k, -1, true, picks, fo);
    return hipGetLastError();
}

template <int SPACE, bool XB, bool DENSE = false>
static hipError_t launch_scan_one(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s,
                                  ScanInfo* info = nullptr) {
    constexpr int NW = 4, MT = 2;
    constexpr int tile_rows = NW * 16 * MT;
    const int64_t tile_begin = row_begin / tile_rows;
    const int64_t tile_end = (row_end + tile_rows - 1) / tile_rows;
    if (tile_end <= tile_begin) return hipSuccess;
    const size_t lds = 2 * kChunkVec * sizeof(uint4) + 3 * kFilterQueries * sizeof(float) +
                       (size_t)NW * 4 * MT * 64 * sizeof(float);
    const int64_t ntiles = tile_end - tile_begin;
    const int max_grid = 256 * 2;  // workgroups resident per launch
    const int grid = (int)(ntiles < max_grid ? ntiles : max_grid);
    auto kern = filter_scan_kernel<SPACE, XB, DENSE>;
    static std::atomic<uint64_t> configured{0};  // per instantiation
    if (hipError_t e = ensure_dynamic_lds(configured, reinterpret_cast<const void*>(kern), (int)lds); e != hipSuccess)
        return e;
    kern<<<grid, NW * 64, lds, s>>>(a, tile_begin, tile_end);
    return hipGetLastError();
}

// narrow kernel: batches of <= 64 queries whose image fits in LDS beside the scratch
constexpr int kNarrowMaxQueries = 64;
constexpr size_t kNarrowLdsMax = 152 * 1024;
static int narrow_nqt(int nq) { return nq <= 16 ? 1 : (nq <= 32 ? 2 : 4); }
static size_t narrow_lds(int32_t ld, int nqt, int nw, bool i8 = false) {
    return (size_t)(ld / (i8 ? 64 : 32)) * nqt * 1024 + 3 * kFilterQueries * sizeof(float) + (size_t)nw * 4 * 2 * 64 * sizeof(float);
}
bool filter_narrow_ok(const FilterArgs& a) {
    if (!(a.Xb || a.X8) || a.nq > kNarrowMaxQueries) return false;  // streams the int8 shadow when the pass has one, else the bf16 one
    if (a.tn->scan_narrow == 0) return false;
    // int8 shadow: the assembly body generated for 4 query tiles (round 4) streams it at the copy ceiling and stages its hits
    // per wave; this compiler-scheduled kernel appends them one atomic at a time and is slower at every batch size (10M x 768:
    // 1.25-1.47 vs 1.20-1.21 ms for 1-8 queries, profiles/r04/small_batch_nqt4_vs_narrow_*.txt).  NARROW_I8_MAX > 0 brings it back (A/B).
    if (a.X8 && a.nq > a.tn->narrow_i8_max) return false;
    return narrow_lds(a.X8 ? a.ld8 : a.ld, narrow_nqt(a.nq), 8, a.X8 != nullptr) <= kNarrowLdsMax;
}
template <int SPACE, int NQT, bool DENSE, int R, int NW, bool I8 = false>
static hipError_t launch_scan_narrow_n(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s, int qgroups = 1) {
    constexpr int tile_rows = NW * 32;
    const int64_t tile_begin = row_begin / tile_rows;
    const int64_t tile_end = (row_end + tile_rows - 1) / tile_rows;
    if (tile_end <= tile_begin) return hipSuccess;
    const size_t lds = narrow_lds(I8 ? a.ld8 : a.ld, NQT, NW, I8);
    const int64_t ntiles = tile_end - tile_begin;
    const int per_cu = (int)std::min<size_t>(32 / NW, (160 * 1024) / lds);  // workgroups resident per CU
    const int max_grid = 256 * (a.tn->narrow_wgs > 0 ? a.tn->narrow_wgs : per_cu);
    // equal tile counts per workgroup: the kernel is a pure stream, a last round with a few busy workgroups is all tail
    const int64_t rounds = (ntiles + max_grid - 1) / max_grid;
    const int grid = a.tn->narrow_balance ? (int)((ntiles + rounds - 1) / rounds)
                                                        : (int)(ntiles < max_grid ? ntiles : max_grid);
    auto kern = filter_scan_narrow_kernel<SPACE, NQT, DENSE, R, NW, I8>;
    static std::atomic<uint64_t> configured{0};  // per instantiation
    if (hipError_t e = ensure_dynamic_lds(configured, reinterpret_cast<const void*>(kern), (int)kNarrowLdsMax); e != hipSuccess)
        return e;
    kern<<<dim3((unsigned)grid, (unsigned)qgroups), NW * 64, lds, s>>>(a, tile_begin, tile_end);
    return hipGetLastError();
}
template <int SPACE, int NQT, bool DENSE>
static hipError_t launch_scan_narrow_q(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s) {
    if (a.X8) {  // int8 shadow: k-steps of 64 columns; ld % 256 == 0, so their count is a multiple of 4
        if constexpr (DENSE) return launch_scan_narrow_n<SPACE, NQT, true, 4, 4, true>(a, row_begin, row_end, s);
        else return launch_scan_narrow_n<SPACE, NQT, false, 4, 8, true>(a, row_begin, row_end, s);
    }
    const bool r4 = (a.ld / 32) % 4 == 0;
    if constexpr (DENSE) {  // the seeding pass is a few tiles: one geometry
        return r4 ? launch_scan_narrow_n<SPACE, NQT, true, 4, 4>(a, row_begin, row_end, s)
                  : launch_scan_narrow_n<SPACE, NQT, true, 2, 4>(a, row_begin, row_end, s);
    } else {
        // 8 waves per workgroup: measured 2-6 % faster than 4 on the 10M x 768 corpus (fewer copies of the image)
        return r4 ? launch_scan_narrow_n<SPACE, NQT, false, 4, 8>(a, row_begin, row_end, s)
                  : launch_scan_narrow_n<SPACE, NQT, false, 2, 8>(a, row_begin, row_end, s);
    }
}
template <int SPACE, bool DENSE>
static hipError_t launch_scan_narrow(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s) {
    switch (narrow_nqt(a.nq)) {
        case 1: return launch_scan_narrow_q<SPACE, 1, DENSE>(a, row_begin, row_end, s);
        case 2: return launch_scan_narrow_q<SPACE, 2, DENSE>(a, row_begin, row_end, s);
        default: return launch_scan_narrow_q<SPACE, 4, DENSE>(a, row_begin, row_end, s);
    }
}

template <int SPACE, int R, int NW, bool NT = false, int QD = 4, bool PRIO = false, int MT = 2, bool DMA = false, bool STAG = false>
static hipError_t launch_scan_asm(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s, ScanInfo* info) {
    constexpr int tile_rows = NW * 16 * MT;
    const int64_t tile_begin = row_begin / tile_rows;
    const int64_t tile_end = (row_end + tile_rows - 1) / tile_rows;
    if (tile_end <= tile_begin) return hipSuccess;
    const size_t lds = scan_code_qbufs(QD) * kChunkVec * sizeof(uint4) + 3 * kFilterQueries * sizeof(float) +
                       (size_t)NW * 12 * scan_code_stage_cap(QD, NW, MT);
    const int64_t ntiles = tile_end - tile_begin;
    const int max_grid = 256 * ((16 / MT) / NW);  // two waves per SIMD on every CU (<= kScanMaxGrid)  // two waves per SIMD on every CU
    const int grid = (int)(ntiles < max_grid ? ntiles : max_grid);
    auto kern = filter_scan_asm_kernel<SPACE, R, NW, NT, QD, PRIO, MT, DMA, STAG>;
    static std::atomic<uint64_t> configured{0};  // per instantiation
    static std::atomic<int> lds_base_ok{0};      // 0 = not checked yet, 1 = ok, -1 = the kernel has static LDS
    if (lds_base_ok.load(std::memory_order_acquire) == 0) {
        hipFuncAttributes attr{};
        if (hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
        lds_base_ok.store(attr.sharedSizeBytes == 0 ? 1 : -1, std::memory_order_release);
    }
    if (lds_base_ok.load(std::memory_order_acquire) < 0) return hipErrorInvalidConfiguration;  // dynamic LDS would not start at 0
    if (hipError_t e = ensure_dynamic_lds(configured, reinterpret_cast<const void*>(kern), (int)lds); e != hipSuccess)
        return e;
    const int xcd_mode = grid % 8 == 0 && ntiles >= grid && a.tn->scan_xcd ? 1 : 0;
    kern<<<grid, NW * 64, lds, s>>>(a, tile_begin, tile_end, xcd_mode);
    // (the kernel's own tail moves the entries into the candidate lists: there is no scatter launch)
    info->nw = NW;
    info->dbg = QD == 108 ? 1 : 0;
    info->i8 = !scan_code_i8(QD) ? 0 : (SPACE == kSpaceCosine ? 1 : (SPACE == kSpaceIp ? 2 : 0));  // how the scatter turns stored values into bounds
    return hipGetLastError();
}

// Picks the scan kernel for a launch.  Default library: the int8 body (ArchVGPR accumulators) wherever the pass has an int8
// shadow, the bf16 body (8 waves, LDS-DMA staging) for an index that keeps a bf16 shadow, the compiler-scheduled kernel for
// an index without shadow.  `make AB=1` adds the tuning variants (Tuning::scan_*; tools/scan_ab.py and the `ab`-marked
// tests); the handle's tuning state picks among them -- nothing here reads the environment.
template <int SPACE>
static hipError_t launch_scan_space(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s, ScanInfo* info) {
    const Tuning& tn = *a.tn;
    const int nkc = a.ld / kFilterChunkK;
    if (filter_narrow_ok(a)) return launch_scan_narrow<SPACE, false>(a, row_begin, row_end, s);  // appends to the lists itself
    if ((a.Xb || a.X8) && tn.scan_asm) {
        // hand-written body (tools/gen_scan_asm.py): one 8-wave workgroup per CU (256-row tiles: the query image is staged
        // once per CU, by LDS-DMA), non-temporal X loads, ring of 4 k-steps -- measured fastest (profiles/r01/scan_ab_*.txt);
        // a ring of R k-steps needs the tile's k-steps to be a multiple of R
#ifdef MLVDB_AB
        if (a.Xb && tn.scan_mt == 4) {  // one wave per SIMD, 64 rows per wave
            if (nkc % 2 == 0) return launch_scan_asm<SPACE, 4, 4, true, 4, false, 4>(a, row_begin, row_end, s, info);
            return launch_scan_asm<SPACE, 2, 4, true, 4, false, 4>(a, row_begin, row_end, s, info);
        }
#endif
        if (a.X8 && a.ld8 > 0) {  // int8 shadow (the caller attached it): ld8/128 chunks, an even number
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: timing diagnostics of the int8 body (wrong results by design)
            if constexpr (SPACE == kSpaceCosine) {
                switch (tn.scan_diag) {
                    case 209: return launch_scan_asm<SPACE, 4, 8, true, 209, true, 2, true>(a, row_begin, row_end, s, info);
                    case 210: return launch_scan_asm<SPACE, 4, 8, true, 210, true, 2, true>(a, row_begin, row_end, s, info);
                    case 212: return launch_scan_asm<SPACE, 4, 8, true, 212, true, 2, true>(a, row_begin, row_end, s, info);
                    case 213: return launch_scan_asm<SPACE, 4, 8, true, 213, true, 2, true>(a, row_begin, row_end, s, info);
                    case 234: return launch_scan_asm<SPACE, 4, 8, true, 234, true, 2, true>(a, row_begin, row_end, s, info);
                    case 223: return launch_scan_asm<SPACE, 4, 8, true, 223, true, 2, true>(a, row_begin, row_end, s, info);
                    case 224: return launch_scan_asm<SPACE, 4, 8, true, 224, true, 2, true>(a, row_begin, row_end, s, info);
                    case 225: return launch_scan_asm<SPACE, 4, 8, true, 225, true, 2, true>(a, row_begin, row_end, s, info);
                    case 226: return launch_scan_asm<SPACE, 4, 8, true, 226, true, 2, true>(a, row_begin, row_end, s, info);
                    case 227: return launch_scan_asm<SPACE, 4, 8, true, 227, true, 2, true>(a, row_begin, row_end, s, info);
                    default: break;
                }
            }
#endif
#ifdef MLVDB_AB
            if constexpr (SPACE != kSpaceL2) {  // round 1: AccVGPR accumulators, serial admission phase; with / without wave priorities
                if (!tn.scan_va) {
                    if (tn.scan_prio != 0) return launch_scan_asm<SPACE, 4, 8, true, 208, true, 2, true>(a, row_begin, row_end, s, info);
                    return launch_scan_asm<SPACE, 4, 8, true, 208, false, 2, true>(a, row_begin, row_end, s, info);
                }
            }
            if constexpr (SPACE == kSpaceCosine) {  // tuning variants of the folded body
                const int var = tn.scan_var;
                if (var == 229 && (a.ld8 / 64) % 6 == 0) return launch_scan_asm<SPACE, 6, 8, true, 229, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 228 && (a.ld8 / 64) % 6 == 0) return launch_scan_asm<SPACE, 6, 8, true, 228, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 214 && (a.ld8 / 64) % 6 == 0) return launch_scan_asm<SPACE, 6, 8, true, 214, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 215) return launch_scan_asm<SPACE, 4, 8, true, 215, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 216) return launch_scan_asm<SPACE, 4, 8, true, 216, false, 2, true>(a, row_begin, row_end, s, info);
                if (var == 219) return launch_scan_asm<SPACE, 4, 8, true, 219, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 222) return launch_scan_asm<SPACE, 4, 8, true, 222, true, 2, true, true>(a, row_begin, row_end, s, info);
                if (var == 220) return launch_scan_asm<SPACE, 4, 8, true, 220, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 221) return launch_scan_asm<SPACE, 4, 8, true, 221, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 230) return launch_scan_asm<SPACE, 4, 4, true, 230, false, 4, true>(a, row_begin, row_end, s, info);
                if (var == 231) return launch_scan_asm<SPACE, 4, 8, true, 231, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 232) return launch_scan_asm<SPACE, 4, 8, true, 232, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 233) return launch_scan_asm<SPACE, 4, 8, true, 233, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 235) return launch_scan_asm<SPACE, 4, 8, true, 235, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 236) return launch_scan_asm<SPACE, 4, 8, true, 236, true, 2, true>(a, row_begin, row_end, s, info);
                if (var == 217) return launch_scan_asm<SPACE, 4, 4, true, 217, false, 2, true>(a, row_begin, row_end, s, info);
                if (var == 218) return launch_scan_asm<SPACE, 4, 4, true, 218, true, 2, true>(a, row_begin, row_end, s, info);
            }
#endif
            if constexpr (SPACE == kSpaceCosine) {  // round 2's default body stays in the default library as the A/B reference
                if (tn.scan_var == 237) return launch_scan_asm<SPACE, 4, 8, true, 237, true, 2, true>(a, row_begin, row_end, s, info);
            }
            // passes of <= 64 / <= 128 queries: the same body computing 4 / 8 of the 16 query tiles (round 4; SCAN_NQT=16 pads)
            const int nqt = tn.scan_nqt > 0 ? tn.scan_nqt : (a.nq <= 64 ? 4 : (a.nq <= 128 ? 8 : 16));
            if constexpr (SPACE == kSpaceL2) {  // folded admission test with per-row integer offsets (the pass computed them: api.hip prep_pass)
                // (the only int8 bodies of l2: api.hip attaches the int8 shadow to an l2 pass only with rp8_cap and l2c set)
                if (nqt <= 4 && a.nq <= 64) return launch_scan_asm<SPACE, 4, 8, true, 245, true, 2, true>(a, row_begin, row_end, s, info);
                if (nqt <= 8 && a.nq <= 128) return launch_scan_asm<SPACE, 4, 8, true, 244, true, 2, true>(a, row_begin, row_end, s, info);
                return launch_scan_asm<SPACE, 4, 8, true, 243, true, 2, true>(a, row_begin, row_end, s, info);
            } else {
                if (nqt <= 4 && a.nq <= 64) return launch_scan_asm<SPACE, 4, 8, true, 242, true, 2, true>(a, row_begin, row_end, s, info);
                if (nqt <= 8 && a.nq <= 128) return launch_scan_asm<SPACE, 4, 8, true, 241, true, 2, true>(a, row_begin, row_end, s, info);
                return launch_scan_asm<SPACE, 4, 8, true, 211, true, 2, true>(a, row_begin, row_end, s, info);
            }
        }
        // everything below streams the bf16 shadow (an int8-only index without usable int8 bounds has none: its
        // fp32 rows are converted in registers by the compiler-scheduled kernel)
        if (!a.Xb) return launch_scan_one<SPACE, false>(a, row_begin, row_end, s, info);
#ifdef MLVDB_AB
        const int nw = tn.scan_nw;
        if constexpr (SPACE == kSpaceCosine) {
            if (nw == 8 && nkc % 2 == 0 && tn.scan_prio == 1)
                return launch_scan_asm<SPACE, 4, 8, true, 4, true>(a, row_begin, row_end, s, info);
            if (nkc % 2 == 0 && tn.scan_nt == 0)
                return nw == 8 ? launch_scan_asm<SPACE, 4, 8, false>(a, row_begin, row_end, s, info)
                               : launch_scan_asm<SPACE, 4, 4, false>(a, row_begin, row_end, s, info);
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: timing diagnostics (tools/scan_ab.py --no-check), wrong results by design
            switch (nw == 8 && nkc % 2 == 0 ? tn.scan_diag : 0) {
                case 101: return launch_scan_asm<SPACE, 4, 8, true, 101>(a, row_begin, row_end, s, info);
                case 102: return launch_scan_asm<SPACE, 4, 8, true, 102>(a, row_begin, row_end, s, info);
                case 103: return launch_scan_asm<SPACE, 4, 8, true, 103>(a, row_begin, row_end, s, info);
                case 104: return launch_scan_asm<SPACE, 4, 8, true, 104>(a, row_begin, row_end, s, info);
                case 107: return launch_scan_asm<SPACE, 4, 8, true, 107>(a, row_begin, row_end, s, info);
                case 108: return launch_scan_asm<SPACE, 4, 8, true, 108>(a, row_begin, row_end, s, info);
                case 109: return launch_scan_asm<SPACE, 4, 8, true, 109>(a, row_begin, row_end, s, info);
                default: break;
            }
#endif
        }
        if (nw == 8 && nkc % 2 == 0 && tn.scan_dma && tn.scan_stag)  // later half of the waves half a tile behind
            return launch_scan_asm<SPACE, 4, 8, true, 4, false, 2, true, true>(a, row_begin, row_end, s, info);
        if (nw == 8 && !tn.scan_dma) {  // query image staged through registers
            if (nkc % 2 == 0) return launch_scan_asm<SPACE, 4, 8, true>(a, row_begin, row_end, s, info);
            return launch_scan_asm<SPACE, 2, 8, true>(a, row_begin, row_end, s, info);
        }
        if (nw != 8) {  // two 4-wave workgroups per CU
            if (nkc % 2 == 0) return launch_scan_asm<SPACE, 4, 4, true>(a, row_begin, row_end, s, info);
            return launch_scan_asm<SPACE, 2, 4, true>(a, row_begin, row_end, s, info);
        }
#endif
        // query image staged by LDS-DMA (+2 % over the register-staged variant)
        if (nkc % 2 == 0) return launch_scan_asm<SPACE, 4, 8, true, 4, false, 2, true>(a, row_begin, row_end, s, info);
        return launch_scan_asm<SPACE, 2, 8, true, 4, false, 2, true>(a, row_begin, row_end, s, info);
    }
    // compiler-scheduled kernel: corpora without shadow (fp32 rows converted in registers), and the A/B reference
    if (a.Xb) return launch_scan_one<SPACE, true>(a, row_begin, row_end, s, info);
    return launch_scan_one<SPACE, false>(a, row_begin, row_end, s, info);
}

// Seeding pass over rows [0, row_end): every bound goes into the candidate lists (slot = row),
// row_end <= kCandCap; then the update kernel (told that every list holds row_end rounded up to
// the tile entries) turns the lists into thresholds and drops what is below them.
static hipError_t launch_update(const FilterArgs& a, int32_t k, int32_t forced_cnt, hipStream_t s);

hipError_t launch_filter_dense_scan(const FilterArgs& a, int64_t rows, hipStream_t s) {
    const bool xb = a.Xb != nullptr;
    if (filter_narrow_ok(a)) {  // (round 3: the narrow kernel runs on the int8 shadow for every space)
        return a.space == kSpaceL2       ? launch_scan_narrow<kSpaceL2, true>(a, 0, rows, s)
               : a.space == kSpaceCosine ? launch_scan_narrow<kSpaceCosine, true>(a, 0, rows, s)
                                         : launch_scan_narrow<kSpaceIp, true>(a, 0, rows, s);
    }
    if (a.X8 && a.tn->seed_i8) {
        // Round 3: the dense pass of a 256-query batch on the int8 shadow too -- the narrow kernel (64 queries' image in LDS)
        // once per group of 64 queries (grid.y): 30 row tiles x 4 groups = 120 workgroups instead of the 30 of the bf16
        // kernel below, half the MFMAs, and no pass of the default path reads the bf16 shadow any more (so an index need
        // not keep one: 1.25x instead of 1.75x the corpus in HBM).  Every bound goes into the lists either way; the refine
        // that follows takes the threshold from exact scores.
        // queries per group: as many as the image leaves room for in LDS (64 up to ld8 = 2304, 32 up to 4736, else 16)
        const int nqt = narrow_lds(a.ld8, 4, 4, true) <= kNarrowLdsMax ? 4 : (narrow_lds(a.ld8, 2, 4, true) <= kNarrowLdsMax ? 2 : 1);
        const int groups = (a.nq + 16 * nqt - 1) / (16 * nqt);
#define MLVDB_SEED_I8(SP)                                                                                              \
    (nqt == 4 ? launch_scan_narrow_n<SP, 4, true, 4, 4, true>(a, 0, rows, s, groups)                                  \
              : nqt == 2 ? launch_scan_narrow_n<SP, 2, true, 4, 4, true>(a, 0, rows, s, groups)                       \
                         : launch_scan_narrow_n<SP, 1, true, 4, 4, true>(a, 0, rows, s, groups))
        return a.space == kSpaceL2 ? MLVDB_SEED_I8(kSpaceL2) : a.space == kSpaceCosine ? MLVDB_SEED_I8(kSpaceCosine) : MLVDB_SEED_I8(kSpaceIp);
#undef MLVDB_SEED_I8
    }
    FilterArgs b = a;
    if (a.X8) b.ke = a.keb;  // bf16 bounds get the bf16 error term; `a.ke` (update, below) covers int8 entries too
    switch (a.space) {
        case kSpaceL2:
            return xb ? launch_scan_one<kSpaceL2, true, true>(b, 0, rows, s) : launch_scan_one<kSpaceL2, false, true>(b, 0, rows, s);
        case kSpaceCosine:
            return xb ? launch_scan_one<kSpaceCosine, true, true>(b, 0, rows, s)
                      : launch_scan_one<kSpaceCosine, false, true>(b, 0, rows, s);
        default:
            return xb ? launch_scan_one<kSpaceIp, true, true>(b, 0, rows, s) : launch_scan_one<kSpaceIp, false, true>(b, 0, rows, s);
    }
}

hipError_t launch_filter_seed_scan(const FilterArgs& a, int64_t row_end, int32_t k, hipStream_t s) {
    const int64_t rows = (row_end + kSeedTileRows - 1) / kSeedTileRows * kSeedTileRows;
    hipError_t e = launch_filter_dense_scan(a, rows, s);
    if (e != hipSuccess) return e;
    if (a.X8) {  // int8 bounds are loose: the threshold comes from exact scores of the best bounds (the same kernel prunes)
        const bool fuse = filter_refine_can_fuse(a);
        if ((e = launch_filter_refine_thr(a, k, (int32_t)rows, fuse, s)) != hipSuccess || fuse) return e;
    }
    return launch_update(a, k, (int32_t)rows, s);  // lists -> thresholds; cnt[q] = survivors
}

hipError_t launch_filter_scan(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s, ScanInfo* info) {
    *info = ScanInfo{};
    switch (a.space) {
        case kSpaceL2: return launch_scan_space<kSpaceL2>(a, row_begin, row_end, s, info);
        case kSpaceCosine: return launch_scan_space<kSpaceCosine>(a, row_begin, row_end, s, info);
        default: return launch_scan_space<kSpaceIp>(a, row_begin, row_end, s, info);
    }
}


static hipError_t launch_update(const FilterArgs& a, int32_t k, int32_t forced_cnt, hipStream_t s) {
    const size_t lds = (size_t)kCandCap * (sizeof(CandEntry) + sizeof(uint32_t)) + 256 * 4 + 64;  // hist + s_scan[16]
    static std::atomic<uint64_t> configured{0};
    if (hipError_t e = ensure_dynamic_lds(configured, reinterpret_cast<const void*>(filter_update_kernel), (int)lds); e != hipSuccess)
        return e;
    filter_update_kernel<<<a.nq, kUpdThreads, lds, s>>>(a, k, forced_cnt);
    return hipGetLastError();
}

hipError_t launch_filter_update(const FilterArgs& a, int32_t k, hipStream_t s) { return launch_update(a, k, -1, s); }

hipError_t launch_filter_rescore(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                 int32_t* out_counts, double* out_d64, unsigned long long* rescored, int32_t* qsel,
                                 int32_t* nflag, hipStream_t s) {
    // every wave keeps its current query in LDS as fp64; at least 96 KiB per block, so that the dispatcher cannot put two on one CU
    int waves = kRescoreWaves;
    while (waves > 1 && (size_t)waves * a.ld * sizeof(double) > 144 * 1024) waves >>= 1;
    const size_t lds = std::max((size_t)waves * a.ld * sizeof(double), (size_t)96 * 1024);
    hipError_t e = hipSuccess;
#define MLVDB_LAUNCH_RESCORE(SP)                                                                                     \
    do {                                                                                                             \
        auto kern = filter_rescore_score_kernel<SP>;                                                                 \
        if (lds > 48 * 1024)                                                                                         \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds);                                                                       \
        if (e == hipSuccess) kern<<<kRescoreGrid, waves * 64, lds, s>>>(a);                                          \
    } while (0)
    switch (a.space) {
        case kSpaceL2: MLVDB_LAUNCH_RESCORE(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_RESCORE(kSpaceCosine); break;
        default: MLVDB_LAUNCH_RESCORE(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_RESCORE
    if (e != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    filter_rescore_rank_kernel<<<a.nq, kRankWaves * 64, 0, s>>>(a, k, q0, out_labels, out_dist, out_counts, out_d64, rescored, qsel,
                                                                nflag);
    return hipGetLastError();
}

hipError_t launch_range_rescore(const FilterArgs& a, float radius, int32_t q0, int64_t capacity, int64_t* out_labels,
                                float* out_dist, int64_t* out_counts, hipStream_t s) {
    const size_t lds_sort = (size_t)kCandCap * (sizeof(double) + sizeof(int32_t));  // the ranking kernel's {d[], l[]} (kCandCap % 64 == 0)
    hipError_t e = hipMemsetAsync(a.rhit_cnt, 0, kFilterQueries * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    // MLVDB_RANGE_FLAT=0: round 2's (query, 256-candidate chunk) grid (A/B)
    const bool flat = a.tn->range_flat != 0;
    int waves = kRescoreWaves;
    while (waves > 1 && (size_t)waves * a.ld * sizeof(double) > 144 * 1024) waves >>= 1;
    const size_t lds_score = flat ? std::max((size_t)waves * a.ld * sizeof(double), (size_t)96 * 1024)
                                  : (size_t)a.ld * sizeof(double) + (size_t)kRangeChunk * sizeof(RangeHit) + 16;
    const dim3 grid = flat ? dim3(kRescoreGrid) : dim3((unsigned)a.nq, (unsigned)((a.cand_cap + kRangeChunk - 1) / kRangeChunk));
    const int threads = flat ? waves * 64 : 256;
#define MLVDB_LAUNCH_RANGE(SP)                                                                                        \
    do {                                                                                                              \
        auto kern = flat ? range_score_flat_kernel<SP> : range_score_kernel<SP>;                                      \
        if (lds_score > 48 * 1024)                                                                                    \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                    (int)lds_score);                                                                  \
        if (e == hipSuccess) kern<<<grid, threads, lds_score, s>>>(a, radius);                                        \
    } while (0)
    switch (a.space) {
        case kSpaceL2: MLVDB_LAUNCH_RANGE(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_RANGE(kSpaceCosine); break;
        default: MLVDB_LAUNCH_RANGE(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_RANGE
    if (e != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(range_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_sort);
    if (e != hipSuccess) return e;
    range_rank_kernel<<<kRankGrid, kRankThreads, lds_sort, s>>>(a, q0, capacity, out_labels, out_dist, out_counts, KnnOut{});
    return hipGetLastError();
}

hipError_t launch_knn_rescore_rank(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                   int32_t* out_counts, double* out_d64, unsigned long long* rescored, hipStream_t s) {
    const size_t lds_sort = (size_t)kCandCap * (sizeof(double) + sizeof(int32_t));
    hipError_t e = hipMemsetAsync(a.rhit_cnt, 0, kFilterQueries * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    int waves = kRescoreWaves;
    while (waves > 1 && (size_t)waves * a.ld * sizeof(double) > 144 * 1024) waves >>= 1;
    const size_t lds_score = std::max((size_t)waves * a.ld * sizeof(double), (size_t)96 * 1024);
    const float inf = __builtin_inff();  // every live entry is a "hit"
#define MLVDB_LAUNCH_KNN_SCORE(SP)                                                                                    \
    do {                                                                                                              \
        auto kern = range_score_flat_kernel<SP>;                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)lds_score);                                                                      \
        if (e == hipSuccess) kern<<<kRescoreGrid, waves * 64, lds_score, s>>>(a, inf);                                \
    } while (0)
    switch (a.space) {
        case kSpaceL2: MLVDB_LAUNCH_KNN_SCORE(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_KNN_SCORE(kSpaceCosine); break;
        default: MLVDB_LAUNCH_KNN_SCORE(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_KNN_SCORE
    if (e != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(range_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_sort);
    if (e != hipSuccess) return e;
    range_rank_kernel<<<kRankGrid, kRankThreads, lds_sort, s>>>(a, q0, (int64_t)k, out_labels, out_dist, nullptr,
                                                                KnnOut{k, out_counts, out_d64, rescored});
    return hipGetLastError();
}

}  // namespace mlvdb
