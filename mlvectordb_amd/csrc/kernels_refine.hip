// Second-level ("mid") bounds and the candidate-list kernels of passes that keep many candidates per query:
// top_k in (64, 1024] ("big-k" passes, api.hip run_bigk_pass) and range queries.
//
// Why.  The int8 scan's bound u of a (query, row) score s is loose by a deterministic offset: eps ~ half a sigma of the
// score distribution (Cauchy-Schwarz on the two quantisation errors), so the rows with u >= thr are 6-12 x the rows with
// s >= thr.  For k = 10 that band is ~120 rows per query and the exact fp64 rescoring swallows it (44 us per 256-query
// wave).  For k = 1000, or a range query with 128 hits, the band is thousands of rows per query, each a gather of 48
// half-used 128-byte lines of the fp32 panels: 0.28 ms of a 2.47 ms range wave (round 3), seconds for big k.
//
// The mid level: a ROW-MAJOR fp16 copy of the corpus, x ~ s16 * h with one scale per row (h in [-32768, 32768]: eleven
// mantissa bits for every component that matters), |x - s16 h| / |x| measured per row when the copy is built and kept
// as an index-wide maximum E16 (~3e-4, rounded up).  A candidate's row is ld16 * 2 contiguous bytes -- whole 128-byte
// lines, half the bytes of the fp32 row and a quarter of its lines -- and
//       |<q, x> - s16 <q, h>| <= |q| E16 |x|                     (Cauchy-Schwarz; the dot product in fp64)
// bounds the score to +-0.01 sigma: the band shrinks to ~1 % of the hits, thresholds taken from these lower bounds are
// (nearly) the exact k-th best, and only the rows that survive them are gathered from the fp32 panels and scored
// exactly (fp64, the arithmetic of every other exact path), which is what is returned.  Nothing is approximated: a row
// is dropped only if an UPPER bound of its score is below a proven LOWER bound of the k-th best score (or of the radius).
//
// The copy costs +50 % of the corpus in HBM (15.4 GB per 10M x 768; an MI355X has 288 GB), is built lazily by the first
// call that wants it (12 ms per 10M rows) and can be switched off (Tuning L2_SHADOW=0: range queries then rescore their
// whole band, big-k searches take the paged exact scan).
//
// Reference: none of this exists there (hnswlib's knn_query, index.py:111, is approximate and capped at 10k rows); the
// contract is SURVEY 8(a) a2 / a8: ids of the exact scan, distances within 1e-5.
#include <algorithm>
#include <type_traits>

#include "bound_common.h"
#include "internal.h"
#include "scan_common.h"

namespace mlvdb {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ the fp16 shadow
// One wave per panel (16 rows): lane 16g + r owns row r, columns 4g..4g+3 of every 16-column group (one coalesced 1 KiB
// load per group, as everywhere).  Pass 1: the row's largest |component|; pass 2 (the panel is in L2 by then): convert,
// store, measure the error.  Whole panels are (re)written: idempotent for rows converted before.
__global__ __launch_bounds__(256) void shadow16_rows_kernel(const float* __restrict__ X, _Float16* __restrict__ X16,
                                                            float* __restrict__ s16, unsigned int* row_err16, int64_t panel_begin,
                                                            int64_t panel_end, int32_t ld, int32_t ld16) {
    const int lane = threadIdx.x & 63;
    const int64_t panel = panel_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (panel >= panel_end) return;
    const int g = lane >> 4, r = lane & 15;
    const int ngroups = ld / 16;
    const float4* src = reinterpret_cast<const float4*>(X + panel * (int64_t)(kPanelRows * ld) + lane_group_offset(lane));
    float m = 0.f;
    for (int cg = 0; cg < ngroups; ++cg) {
        const float4 v = src[cg * (kGroupFloats / 4)];
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))),
                            __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
    }
    m = __builtin_fmaxf(m, __shfl_xor(m, 16));
    m = __builtin_fmaxf(m, __shfl_xor(m, 32));
    const float sx = m > 0.f ? m * (1.0f / 32768.0f) : 1.0f;  // |x_i / sx| <= 32768 (1 + 2^-23): inside fp16's range
    const float inv = 1.0f / sx;
    const int64_t row = panel * kPanelRows + r;
    _Float16* dst = X16 + row * (int64_t)ld16 + 4 * g;
    double err2 = 0.0, n2 = 0.0;
    for (int cg = 0; cg < ngroups; ++cg) {
        const float4 v = src[cg * (kGroupFloats / 4)];
        const float x[4] = {v.x, v.y, v.z, v.w};
        _Float16 hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hv[i] = (_Float16)(x[i] * inv);
            const double e = (double)x[i] - (double)sx * (double)(float)hv[i];
            err2 += e * e;
            n2 += (double)x[i] * (double)x[i];
        }
        *reinterpret_cast<uint2*>(dst + cg * 16) = *reinterpret_cast<const uint2*>(hv);
    }
    err2 += __shfl_xor(err2, 16);
    err2 += __shfl_xor(err2, 32);
    n2 += __shfl_xor(n2, 16);
    n2 += __shfl_xor(n2, 32);
    if (g == 0) {
        s16[row] = sx;
        if (n2 > 0.0) {
            float rel = (float)(__builtin_sqrt(err2 / n2) * 1.000001);
            rel = __uint_as_float(__float_as_uint(rel) + 1u);  // never below the true relative error
            atomicMax(row_err16, __float_as_uint(rel));        // non-negative floats order like their bits
        }
    }
}

hipError_t launch_shadow16_rows(const float* X, void* X16, float* s16, float* row_err16, int64_t row_begin, int64_t row_end,
                                int32_t ld, int32_t ld16, hipStream_t s) {
    const int64_t pb = row_begin / kPanelRows, pe = (row_end + kPanelRows - 1) / kPanelRows;
    if (pe <= pb) return hipSuccess;
    shadow16_rows_kernel<<<(unsigned)((pe - pb + 3) / 4), 256, 0, s>>>(X, static_cast<_Float16*>(X16), s16,
                                                                         reinterpret_cast<unsigned int*>(row_err16), pb, pe, ld, ld16);
    return hipGetLastError();
}

// ------------------------------------------------------------------ mid bounds
// Half-width of the mid bound in score units (kernels_filter.hip's header: cosine s = <q^,x>/(|x|+1e-30), ip s = <q^,x>,
// l2 s = 2 <q,x> - |x|^2):  |s - s_mid| <= mid_halfwidth.  e16 = E16, qn = |q|, xn = |x| as stored (fp32 of the fp64 norm).
//   <q,x> - s16 <q,h> = <q, x - s16 h>,  |.| <= |q| E16 |x|;  the fp64 sum over <= 8192 terms adds < 1e-12 relative;
//   cosine: the stored norm is off by <= 6e-8 relative, the score by as much (|s| <= 1);  l2: |x|^2 from the stored norm is
//   off by <= 1.3e-7 |x|^2;  everything is rounded up with room to spare.
template <int SPACE>
__device__ __forceinline__ double mid_halfwidth(double e16, double qn, double xn) {
    if (SPACE == kSpaceCosine) return e16 * 1.000001 + 2.0e-7;
    if (SPACE == kSpaceIp) return (e16 * 1.000001 + 1.0e-7) * xn;
    return 2.0 * qn * xn * (e16 * 1.000001 + 1.0e-7) + 4.0e-7 * xn * xn;
}

constexpr int kMidWaves = 16;  // waves per block: each keeps its current query in LDS as fp64
constexpr int kMidGrid = 256;  // one block per CU (each asks for more than half of a CU's LDS)

// Work items: the 8-entry groups of every query's pick list (picks != nullptr: indices into the candidate list, written by
// bigk_select_kernel) or of its whole list (picks == nullptr: every entry that is not refined yet), as one flat list dealt
// to the waves of the grid in turn -- the flat distribution of filter_rescore_score_kernel.  A wave step = 8 rows, 8 lanes
// per row, every lane's 16-byte pieces of the row 128 bytes apart: whole lines, all of a row's loads in flight.
// Result: the entry's u becomes the mid UPPER bound (rounded up to fp32), its row gets kRefinedBit; a row that is dead
// (tombstoned / masked out: NaN norm) gets u = NaN and drops out of every later selection.
template <int SPACE>
__global__ __launch_bounds__(kMidWaves * 64) void mid_score_flat_kernel(const FilterArgs a, const MidArgs m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t pre[kFilterQueries + 1];
    const int ld16 = m.ld16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    double* qs = reinterpret_cast<double*>(smem) + (size_t)wave * ld16;
    if (wave == 0) {  // lane l: queries 4l .. 4l+3; inclusive scan over the lanes
        uint32_t n[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 4 * lane + i;
            uint32_t c = 0;
            if (q < a.nq && !a.overflow[q]) c = m.picks ? min(m.npicks[q], (uint32_t)m.picks_cap) : min(a.cnt[q], (uint32_t)a.cand_cap);
            n[i] = (c + 7u) >> 3;
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
    const double e16 = (double)*m.row_err16;
    const int r8 = lane >> 3, l8 = lane & 7;
    const int nsteps = ld16 >> 6;  // 64 columns (128 bytes) per row and step
    int cur = -1;
    double qaux = 0.0, qn = 0.0;
    // every wave takes a CONTIGUOUS run of groups (waves numbered block-first, so a short list still spreads over the CUs):
    // consecutive groups mostly belong to one query, whose fp64 copy in LDS is then loaded once per run -- dealt out one by
    // one (the rescoring kernel's order) nearly every group of a long list meant a new query: 6 KB from L2 and a dependent
    // round trip before the gather could start (145 us per range wave for 256k rows; profiles/r04)
    const uint32_t nw_all = gridDim.x * nwaves, gw = (uint32_t)wave * gridDim.x + blockIdx.x;
    const uint32_t per = (total + nw_all - 1) / nw_all;
    const uint32_t u_end = min(total, (gw + 1) * per);
    for (uint32_t u = gw * per; u < u_end; ++u) {
        int lo = 0, hi = kFilterQueries;  // the query with pre[q] <= u < pre[q + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= u) lo = mid;
            else hi = mid;
        }
        const int q = lo;
        const uint32_t nq_items = m.picks ? min(m.npicks[q], (uint32_t)m.picks_cap) : min(a.cnt[q], (uint32_t)a.cand_cap);
        CandEntry* list = a.cand + (int64_t)q * a.cand_cap;
        const uint32_t item = (u - pre[q]) * 8 + r8;
        bool have = item < nq_items;
        uint32_t li = 0;
        CandEntry e{};
        if (have) {
            li = m.picks ? m.picks[(int64_t)q * m.picks_cap + item] : item;
            e = list[li];
            have = !((uint32_t)e.row & kRefinedBit) && e.u == e.u;  // (refined before, or never a candidate: padding / dead row)
        }
        if (!__any(have)) continue;  // wave-uniform
        if (q != cur) {  // (wave-private LDS: the wave's own earlier reads are done -- its loop is in order)
            for (int c = lane; c < ld16; c += 64) qs[c] = c < a.ld ? (double)a.Qpad[(int64_t)q * a.ld + c] : 0.0;
            qaux = a.qaux[q];
            qn = SPACE == kSpaceCosine ? 0.0 : qaux;  // ip / l2: qaux = |q|
            cur = q;
        }
        const int32_t row = have ? (int32_t)((uint32_t)e.row & ~kRefinedBit) : 0;
        const uint4* rp = reinterpret_cast<const uint4*>(m.X16 + (int64_t)row * ld16) + l8;
        double acc = 0.0;
        constexpr int G = 12;  // pieces in flight per lane: a whole 768-column row in one round trip (the gather is a chain of
                               // dependent round trips per wave, each a page walk per random row: 6 in flight made it two chains)
        for (int s0 = 0; s0 < nsteps; s0 += G) {
            uint4 v[G];
#pragma unroll
            for (int j = 0; j < G; ++j) v[j] = rp[(s0 + j < nsteps ? s0 + j : s0) * 8];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (s0 + j >= nsteps) continue;
                const half8 hv = __builtin_bit_cast(half8, v[j]);
                const double* qp = qs + ((s0 + j) * 8 + l8) * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc = __builtin_fma(qp[t], (double)(float)hv[t], acc);
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if (have && l8 == 0) {
            const float nrm = a.rn[row];
            CandEntry o;
            o.row = (int32_t)((uint32_t)row | kRefinedBit);
            if (!(nrm == nrm)) {
                o.u = __builtin_nanf("");  // tombstoned / masked out: not a candidate
            } else {
                const double xn = (double)nrm;
                const double dot = acc * (double)m.s16[row];
                double s;
                if (SPACE == kSpaceCosine) s = dot * qaux / (xn + 1e-30);
                else if (SPACE == kSpaceIp) s = qaux > 0.0 ? dot / qaux : 0.0;
                else s = 2.0 * dot - xn * xn;
                const double up = s + mid_halfwidth<SPACE>(e16, qn, xn);
                o.u = up < -3.0e38 ? -3.0e38f : (up > 3.0e38 ? 3.0e38f : float_above(up));
            }
            list[li] = o;
        }
    }
}

hipError_t launch_mid_score(const FilterArgs& a, const MidArgs& m, hipStream_t s) {
    int waves = kMidWaves;
    while (waves > 1 && (size_t)waves * m.ld16 * sizeof(double) > 144 * 1024) waves >>= 1;
    const size_t lds = std::max((size_t)waves * m.ld16 * sizeof(double), (size_t)96 * 1024);
    hipError_t e = hipSuccess;
#define MLVDB_LAUNCH_MID(SP)                                                                                         \
    do {                                                                                                             \
        auto kern = mid_score_flat_kernel<SP>;                                                                       \
        static std::atomic<uint64_t> configured{0};                                                                  \
        e = ensure_dynamic_lds(configured, reinterpret_cast<const void*>(kern), 156 * 1024);                         \
        if (e == hipSuccess) kern<<<kMidGrid, waves * 64, lds, s>>>(a, m);                                            \
    } while (0)
    switch (a.space) {
        case kSpaceL2: MLVDB_LAUNCH_MID(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_MID(kSpaceCosine); break;
        default: MLVDB_LAUNCH_MID(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_MID
    return e != hipSuccess ? e : hipGetLastError();
}

// ------------------------------------------------------------------ block-wide k-th largest key
// T with #(key > T) < kth <= #(key >= T) over the non-zero keys key(i), i < cnt (0 = not a candidate); needs kth <= their
// number; kmin / kmax = smallest / largest non-zero key (block-uniform).  Radix select, most significant digit first, the
// digits starting at the highest bit in which two keys differ (the keys of one list are floats of similar size: a
// histogram over their common leading bits would be every thread adding to one LDS word).  hist: [256], s_sel: [2].
template <int NT, class KeyFn>
__device__ __forceinline__ uint32_t block_kth_largest(KeyFn key, uint32_t cnt, uint32_t kth, uint32_t kmin, uint32_t kmax,
                                                      uint32_t* hist, uint32_t* s_sel) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t diff = kmin ^ kmax;
    uint32_t prefix = kmax, mask = 0xffffffffu, want = kth;  // all keys equal: the answer is that key
    if (diff == 0) return prefix;
    const int hb = 31 - __builtin_clz(diff);
    mask = hb == 31 ? 0u : ~((2u << hb) - 1u);
    prefix = kmax & mask;
    int shift = hb >= 7 ? hb - 7 : 0, width = hb - shift + 1;
    for (;;) {
        __syncthreads();
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t dmask = (1u << width) - 1u;
        for (uint32_t idx = threadIdx.x; idx < cnt; idx += NT) {
            const uint32_t k = key(idx);
            if (k != 0 && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & dmask], 1u);
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
                if (c3 >= want) { s_sel[0] = 4 * lane + 3; s_sel[1] = above; }
                else if (c2 >= want) { s_sel[0] = 4 * lane + 2; s_sel[1] = c3; }
                else if (c1 >= want) { s_sel[0] = 4 * lane + 1; s_sel[1] = c2; }
                else { s_sel[0] = 4 * lane; s_sel[1] = c1; }
            }
        }
        __syncthreads();
        prefix |= s_sel[0] << shift;
        mask |= dmask << shift;
        want -= s_sel[1];
        if (shift == 0) break;
        const int next = shift >= 8 ? shift - 8 : 0;
        width = shift - next;
        shift = next;
    }
    return prefix;
}

// block-wide sum / min / max of three per-thread values (scratch: [3 * 16] words)
template <int NT>
__device__ __forceinline__ void block_count_min_max(uint32_t& n, uint32_t& kmin, uint32_t& kmax, uint32_t* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off);
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, off));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, off));
    }
    __syncthreads();
    if (lane == 0) {
        scratch[wave] = n;
        scratch[16 + wave] = kmin;
        scratch[32 + wave] = kmax;
    }
    __syncthreads();
    n = 0;
    kmin = 0xffffffffu;
    kmax = 0;
    for (int w = 0; w < NT / 64; ++w) {
        n += scratch[w];
        kmin = min(kmin, scratch[16 + w]);
        kmax = max(kmax, scratch[32 + w]);
    }
}

// ------------------------------------------------------------------ big-k: which entries get a mid bound this round
// One block per query.  The `want` entries with the largest bounds (refined or not: a refined entry's u is its tight mid
// bound, an unrefined one's the scan's loose bound, so the unrefined rows that could still be among the k best sort on top)
// -- of those, the ones without a mid bound yet go to the pick list.  int8 bounds are loose by a near-constant offset, not
// by noise, so ranking by bound is ranking by score to ~0.01 sigma: after these picks are refined, the k-th largest mid
// lower bound is (nearly) the exact k-th best score of the rows seen so far (bigk_thr_prune_kernel).
constexpr int kSelThreads = 1024;
__global__ __launch_bounds__(kSelThreads) void bigk_select_kernel(const FilterArgs a, const int32_t want, const int32_t forced_cnt,
                                                                   uint32_t* picks, uint32_t* npicks, const int32_t picks_cap) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t scratch[48];
    __shared__ uint32_t s_sel[2];
    __shared__ uint32_t s_n;
    const int q = blockIdx.x;
    if (q >= a.nq) return;
    if (threadIdx.x == 0) s_n = 0;
    const uint32_t raw = forced_cnt >= 0 ? (uint32_t)forced_cnt : a.cnt[q];
    if (a.overflow[q] || raw > (uint32_t)a.cand_cap) {  // (a list longer than its capacity: the scan's tail set the flag too)
        if (threadIdx.x == 0) {
            npicks[q] = 0;
            if (raw > (uint32_t)a.cand_cap) a.overflow[q] = 1u;
        }
        return;
    }
    const uint32_t cnt = raw;
    const CandEntry* list = a.cand + (int64_t)q * a.cand_cap;
    auto key = [&](uint32_t idx) -> uint32_t {
        const float u = list[idx].u;
        if (!(u == u)) return 0u;
        const uint32_t k = float_order_key(u);
        return k ? k : 1u;
    };
    uint32_t n = 0, kmin = 0xffffffffu, kmax = 0;
    for (uint32_t idx = threadIdx.x; idx < cnt; idx += kSelThreads) {
        const uint32_t k = key(idx);
        if (k) {
            ++n;
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
    }
    block_count_min_max<kSelThreads>(n, kmin, kmax, scratch);
    uint32_t T = 1u;  // short lists whole
    if (n > (uint32_t)want) T = block_kth_largest<kSelThreads>(key, cnt, (uint32_t)want, kmin, kmax, hist, s_sel);
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < cnt; idx += kSelThreads) {
        const uint32_t k = key(idx);
        if (k >= T && k != 0 && !((uint32_t)list[idx].row & kRefinedBit)) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < (uint32_t)picks_cap) picks[(int64_t)q * picks_cap + pos] = idx;  // (beyond: masses of equal bounds; they stay unrefined)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) npicks[q] = min(s_n, (uint32_t)picks_cap);
}

hipError_t launch_bigk_select(const FilterArgs& a, int32_t want, int32_t forced_cnt, uint32_t* picks, uint32_t* npicks,
                              int32_t picks_cap, hipStream_t s) {
    bigk_select_kernel<<<a.nq, kSelThreads, 0, s>>>(a, want, forced_cnt, picks, npicks, picks_cap);
    return hipGetLastError();
}

// ------------------------------------------------------------------ threshold from the mid lower bounds + pruning
// One block per query.  k > 0: thr[q] = max(thr[q], k-th largest mid LOWER bound among the refined entries) -- k rows are
// known to score at least that, so it is a proven lower bound of the final k-th best score.  Then the list is compacted in
// place to the entries with u >= thr (refined: tight upper bound; unrefined: the scan's bound; NaN = dead rows drop out).
// k == 0 (range passes): the threshold is the radius' and stays; only the pruning.
template <int SPACE>
__global__ __launch_bounds__(kSelThreads) void bigk_thr_prune_kernel(const FilterArgs a, const MidArgs m, const int32_t k,
                                                                      const int32_t forced_cnt) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t scratch[48];
    __shared__ uint32_t s_sel[2];
    __shared__ uint32_t s_w[kSelThreads / 64];
    const int q = blockIdx.x;
    if (q >= a.nq || a.overflow[q]) return;
    const uint32_t raw = forced_cnt >= 0 ? (uint32_t)forced_cnt : a.cnt[q];
    if (raw > (uint32_t)a.cand_cap) {
        if (threadIdx.x == 0) a.overflow[q] = 1u;
        return;
    }
    const uint32_t cnt = raw;
    CandEntry* list = a.cand + (int64_t)q * a.cand_cap;
    float thr = a.thr[q];
    if (k > 0) {
        const double e16 = (double)*m.row_err16;
        const double qn = SPACE == kSpaceCosine ? 0.0 : a.qaux[q];
        auto key = [&](uint32_t idx) -> uint32_t {  // order key of the entry's mid lower bound (0: not refined / dead)
            const CandEntry e = list[idx];
            if (!((uint32_t)e.row & kRefinedBit) || !(e.u == e.u)) return 0u;
            const double xn = SPACE == kSpaceCosine ? 0.0 : (double)a.rn[(uint32_t)e.row & ~kRefinedBit];
            // u = fl_up(s_mid + hw): the lower bound is s_mid - hw >= u - 2 hw - (one fp32 ulp of u)
            const double lo = (double)e.u - 2.0 * mid_halfwidth<SPACE>(e16, qn, xn) - 1.3e-7 * __builtin_fabs((double)e.u);
            const uint32_t kk = float_order_key(lo < -3.0e38 ? -3.0e38f : float_below(lo));
            return kk ? kk : 1u;
        };
        uint32_t n = 0, kmin = 0xffffffffu, kmax = 0;
        for (uint32_t idx = threadIdx.x; idx < cnt; idx += kSelThreads) {
            const uint32_t kk = key(idx);
            if (kk) {
                ++n;
                kmin = min(kmin, kk);
                kmax = max(kmax, kk);
            }
        }
        block_count_min_max<kSelThreads>(n, kmin, kmax, scratch);
        if (n >= (uint32_t)k) {
            const uint32_t T = block_kth_largest<kSelThreads>(key, cnt, (uint32_t)k, kmin, kmax, hist, s_sel);
            const float lk = float_from_order_key(T);
            if (lk > thr) thr = lk;
        }
    }
    // in-place compaction, chunk by chunk: a chunk is read whole before any of it is written, and the write positions
    // never pass the chunk's own indices
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t out = 0;
    for (uint32_t base = 0; base < cnt; base += kSelThreads) {
        const uint32_t idx = base + threadIdx.x;
        CandEntry e{};
        bool keep = false;
        if (idx < cnt) {
            e = list[idx];
            keep = e.u >= thr;
        }
        const unsigned long long bal = __ballot(keep);
        __syncthreads();  // (s_w of the previous chunk has been read by everyone)
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
        __syncthreads();  // every thread has read its entry; the wave counts are visible
        uint32_t pos = out, tot = 0;
        for (int w = 0; w < kSelThreads / 64; ++w) {
            if (w < wave) pos += s_w[w];
            tot += s_w[w];
        }
        if (keep) list[pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = e;
        out += tot;
    }
    if (threadIdx.x == 0) {
        a.thr[q] = thr;
        a.cnt[q] = out;
    }
}

hipError_t launch_bigk_thr_prune(const FilterArgs& a, const MidArgs& m, int32_t k, int32_t forced_cnt, hipStream_t s) {
    switch (a.space) {
        case kSpaceL2: bigk_thr_prune_kernel<kSpaceL2><<<a.nq, kSelThreads, 0, s>>>(a, m, k, forced_cnt); break;
        case kSpaceCosine: bigk_thr_prune_kernel<kSpaceCosine><<<a.nq, kSelThreads, 0, s>>>(a, m, k, forced_cnt); break;
        default: bigk_thr_prune_kernel<kSpaceIp><<<a.nq, kSelThreads, 0, s>>>(a, m, k, forced_cnt); break;
    }
    return hipGetLastError();
}

}  // namespace mlvdb
