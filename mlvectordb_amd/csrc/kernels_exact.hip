// Exact streaming scan: fp64 distances of every live row against a tile of QT queries, fused
// wave-level top-k, per-block partial lists, final merge.
//
// This is the universal exact path of the library: it serves small query batches at HBM
// speed (the fp64 arithmetic for <= 8 queries hides under the corpus stream), it seeds the
// thresholds of the bf16 filter path, and it is the fallback for any query the filter path
// gives up on.  It takes the place of hnswlib's knn_query as called from the reference
// (src/mlvectordb/implementations/index.py:111), computed exhaustively.
#include <algorithm>
#include <cstdlib>

#include "internal.h"
#include "scan_common.h"

namespace mlvdb {

template <int SPACE, int QT, int PW, int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void exact_scan_kernel(const ExactArgs a, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);  // [QT][ld]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int r = lane & 15;
    const int ld = a.ld;
    const int nq_sel = a.nq_sel_dev ? min(*a.nq_sel_dev, a.nq_sel) : a.nq_sel;
    if ((int)blockIdx.y * QT >= nq_sel) return;  // block-uniform: nothing selected for this query tile

    int qid[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int sel = blockIdx.y * QT + t;
        qid[t] = sel < nq_sel ? (a.qsel ? a.qsel[sel] : sel) : -1;
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        for (int c = threadIdx.x; c < ld; c += NW * 64)
            qs[t * ld + c] = qid[t] >= 0 ? (double)a.Qpad[(int64_t)qid[t] * ld + c] : 0.0;
    }
    __syncthreads();

    double qinv[QT];
    double cur_d[QT];
    int32_t cur_l[QT];
    WaveTopK top[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qinv[t] = qid[t] >= 0 ? a.qaux[qid[t]] : 0.0;
        const bool has_cur = a.cursor_d != nullptr && qid[t] >= 0;
        cur_d[t] = has_cur ? a.cursor_d[qid[t]] : -__builtin_inf();
        cur_l[t] = has_cur ? a.cursor_l[qid[t]] : -1;
        top[t].init();
    }

    const int64_t panel_begin = a.row_begin >> 4;
    const int64_t panel_end = (a.row_end + 15) >> 4;
    const int64_t ntasks = (panel_end - panel_begin + PW - 1) / PW;
    for (int64_t task = (int64_t)blockIdx.x * NW + wave; task < ntasks; task += (int64_t)nblk * NW) {
        const float* base[PW];
        int64_t panel[PW];
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            panel[p] = panel_begin + task * PW + p;
            const int64_t pp = panel[p] < panel_end ? panel[p] : panel_begin;  // keep the address valid
            base[p] = a.X + pp * (int64_t)(kPanelRows * ld) + lane_group_offset(lane);
        }
        double acc[PW][QT];
        double nx[PW];
        accumulate_rows<SPACE, QT, PW, (QT == 8 ? 4 : 0), NT>(base, qs, ld, g, acc, nx);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int64_t row = panel[p] * kPanelRows + r;
            bool live = lane < 16 && panel[p] < panel_end && row >= a.row_begin && row < a.row_end;
            if (live) {
                const float nrm = a.rn[row];
                live = nrm == nrm;  // NaN marks a tombstoned row
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const double dist = finish_distance<SPACE>(acc[p][t], nx[p], qinv[t]);
                const bool want = live && qid[t] >= 0 && entry_less(cur_d[t], cur_l[t], dist, (int32_t)row);
                top[t].offer(want, dist, (int32_t)row, a.k, lane);
            }
        }
    }

    // ---- block merge: lists of all waves through LDS (aliases the query tile)
    __syncthreads();
    double* ld_d = reinterpret_cast<double*>(smem);                                  // [NW][QT][64]
    int32_t* ld_l = reinterpret_cast<int32_t*>(smem + (size_t)NW * QT * 64 * sizeof(double));  // [NW][QT][64]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        ld_d[(wave * QT + t) * 64 + lane] = top[t].d;
        ld_l[(wave * QT + t) * 64 + lane] = top[t].l;
    }
    __syncthreads();
    for (int t = wave; t < QT; t += NW) {
        WaveTopK m;
        m.init();
        for (int w2 = 0; w2 < NW; ++w2) {
            const double cd = ld_d[(w2 * QT + t) * 64 + lane];
            const int32_t cl = ld_l[(w2 * QT + t) * 64 + lane];
            m.offer(lane < a.k && cl != kNoLabel, cd, cl, a.k, lane);
        }
        const int sel = blockIdx.y * QT + t;
        if (sel < nq_sel && lane < a.k) {
            TopEntry e;
            e.d = m.d;
            e.l = m.l;
            e.pad = 0;
            a.partial[((int64_t)sel * nblk + blockIdx.x) * a.k + lane] = e;
        }
    }
}

// One block (4 waves) per selected query: each wave folds a quarter of the partial entries,
// wave 0 folds the four lists and writes the final answer.
__global__ __launch_bounds__(256) void exact_merge_kernel(const TopEntry* __restrict__ partial,
                                                          const int32_t* nq_sel_dev, const int32_t* qsel,
                                                          int32_t nblk, int32_t k, int64_t* out_labels,
                                                          float* out_dist, int32_t* out_counts, double* out_d64) {
    __shared__ double sd[4][64];
    __shared__ int32_t sl[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sel = blockIdx.x;
    if (nq_sel_dev && sel >= *nq_sel_dev) return;
    const int q = qsel ? qsel[sel] : sel;
    const TopEntry* src = partial + (int64_t)sel * nblk * k;
    const int64_t n = (int64_t)nblk * k;
    WaveTopK m;
    m.init();
    for (int64_t i0 = (int64_t)wave * 64; i0 < n; i0 += 256) {
        const int64_t i = i0 + lane;
        TopEntry e;
        e.d = __builtin_inf();
        e.l = kNoLabel;
        if (i < n) e = src[i];
        m.offer(e.l != kNoLabel, e.d, e.l, k, lane);
    }
    sd[wave][lane] = m.d;
    sl[wave][lane] = m.l;
    __syncthreads();
    if (wave != 0) return;
    WaveTopK f;
    f.init();
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
        const double cd = sd[w2][lane];
        const int32_t cl = sl[w2][lane];
        f.offer(lane < k && cl != kNoLabel, cd, cl, k, lane);
    }
    const bool valid = lane < k && f.l != kNoLabel;
    if (lane < k) {
        out_labels[(int64_t)q * k + lane] = valid ? (int64_t)f.l : -1;
        out_dist[(int64_t)q * k + lane] = valid ? (float)f.d : __builtin_inff();
        if (out_d64) out_d64[(int64_t)q * k + lane] = valid ? f.d : __builtin_inf();
    }
    const int cnt = __popcll(__ballot(valid));
    if (lane == 0) out_counts[q] = cnt;
}

// Range-query candidate generator: same scan, but every live row with dist <= radius is
// appended to its query's candidate list (rescored and sorted by range_rescore_kernel).
template <int SPACE>
__global__ __launch_bounds__(512) void exact_range_kernel(const FilterArgs a, const double radius, const int nblk,
                                                          const int32_t* qsel, const int32_t nsel) {
    constexpr int QT = 4, PW = 2, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);  // [QT][ld]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int r = lane & 15;
    const int ld = a.ld;
    int qid[QT];
    double qinv[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int sel = blockIdx.y * QT + t;
        qid[t] = sel < nsel ? (qsel ? qsel[sel] : sel) : -1;
        for (int c = threadIdx.x; c < ld; c += NW * 64)
            qs[t * ld + c] = qid[t] >= 0 ? (double)a.Qpad[(int64_t)qid[t] * ld + c] : 0.0;
        qinv[t] = qid[t] >= 0 ? a.qaux[qid[t]] : 0.0;
    }
    __syncthreads();
    const int64_t panel_end = (a.total + 15) >> 4;
    const int64_t ntasks = (panel_end + PW - 1) / PW;
    for (int64_t task = (int64_t)blockIdx.x * NW + wave; task < ntasks; task += (int64_t)nblk * NW) {
        const float* base[PW];
        int64_t panel[PW];
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            panel[p] = task * PW + p;
            const int64_t pp = panel[p] < panel_end ? panel[p] : 0;
            base[p] = a.X + pp * (int64_t)(kPanelRows * ld) + lane_group_offset(lane);
        }
        double acc[PW][QT];
        double nx[PW];
        accumulate_rows<SPACE, QT, PW>(base, qs, ld, g, acc, nx);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int64_t row = panel[p] * kPanelRows + r;
            bool live = lane < 16 && panel[p] < panel_end && row < a.total;
            if (live) {
                const float nrm = a.rn[row];
                live = nrm == nrm;
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const double dist = finish_distance<SPACE>(acc[p][t], nx[p], qinv[t]);
                if (live && qid[t] >= 0 && dist <= radius) {
                    const uint32_t slot = atomicAdd(&a.cnt[qid[t]], 1u);
                    if (slot < (uint32_t)a.cand_cap) {
                        CandEntry e;
                        e.u = 0.f;
                        e.row = (int32_t)row;
                        a.cand[(int64_t)qid[t] * a.cand_cap + slot] = e;
                    } else {
                        a.overflow[qid[t]] = 1u;
                    }
                }
            }
        }
    }
}

hipError_t launch_exact_range_scan(const FilterArgs& a, float radius, const int32_t* qsel, int32_t nsel, hipStream_t s) {
    if (!qsel) nsel = a.nq;
    if (nsel <= 0) return hipSuccess;
    const int nqtiles = (nsel + 3) / 4;
    const int64_t ntasks = ((a.total + 15) / 16 + 1) / 2;
    int64_t nblk = (ntasks + 7) / 8;
    const int64_t cap = std::max<int64_t>(8, 1024 / nqtiles);
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    const size_t lds = (size_t)4 * a.ld * sizeof(double);
    hipError_t e = hipSuccess;
#define MLVDB_LAUNCH_ER(SP)                                                                                      \
    do {                                                                                                         \
        auto kern = exact_range_kernel<SP>;                                                                      \
        if (lds > 48 * 1024)                                                                                     \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds);                                                                   \
        if (e == hipSuccess) kern<<<dim3((unsigned)nblk, nqtiles), 512, lds, s>>>(a, (double)radius, (int)nblk, qsel, nsel); \
    } while (0)
    switch (a.space) {
        case kSpaceL2: MLVDB_LAUNCH_ER(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_ER(kSpaceCosine); break;
        default: MLVDB_LAUNCH_ER(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_ER
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ host side
// ------------------------------------------------------------------ exact distances of given (query, label) pairs
// The vector-level face of the same arithmetic (reference README.md:30-41,178-181: SimpleVector.distance / similarity):
// out[q][j] = distance of query q to row labels[q][j] in the index's space, fp64, by the very accumulate_rows the exact
// scan and the rescoring use -- so a pair scored here is bit-identical to the score a search returns for it.
// Block (q, chunk): 4 waves, each scores 16 pairs per step.  label < 0: +inf (search padding passes through).
template <int SPACE>
__global__ __launch_bounds__(256) void pair_distance_kernel(const float* __restrict__ X, const float* __restrict__ Qpad,
                                                            const double* __restrict__ qaux, const int64_t* __restrict__ labels,
                                                            int32_t m, int32_t ld, double* __restrict__ out64,
                                                            float* __restrict__ out32) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);  // [ld]
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    for (int c = threadIdx.x; c < ld; c += 256) qs[c] = (double)Qpad[(int64_t)q * ld + c];
    __syncthreads();
    const double qinv = qaux[q];
    const int64_t* lab = labels + (int64_t)q * m;
    for (int j0 = (blockIdx.y * 4 + wave) * 16; j0 < m; j0 += gridDim.y * 64) {
        const int j = j0 + r;
        const bool have = j < m;
        const int64_t row = have ? lab[j] : -1;
        const int64_t rr = row >= 0 ? row : 0;
        const float* base[1] = {X + (rr >> 4) * (int64_t)(kPanelRows * ld) + (rr & 15) * 16 + g * 4};
        double acc[1][1], nx[1];
        accumulate_rows<SPACE, 1, 1, 8>(base, qs, ld, g, acc, nx);
        const double d = row >= 0 ? finish_distance<SPACE>(acc[0][0], nx[0], qinv) : __builtin_inf();
        if (have && lane < 16) {
            out64[(int64_t)q * m + j] = d;
            if (out32) out32[(int64_t)q * m + j] = (float)d;
        }
    }
}

// ------------------------------------------------------------------ exact k-th best of a PREFIX of the rows (small batches)
// The seed of a filter pass of <= 8 queries: exact fp64 distances of rows 0..m-1 (accumulate_rows: the scores every other
// exact path gives them), tombstoned rows +inf, one 16-row group per wave so the few thousand rows spread over the chip as
// m/64 blocks per query; filter_prefix_thr_kernel (kernels_filter.hip) then takes the k-th best per query by a radix select
// on order keys (per-wave sorted top-k lists with fp64 lane shuffles took 34 us for 3840 rows: profiles/r03).
// Replaces the dense int8 seeding pass + exact-threshold refine (two latency chains, 23 us at batch 1) where the fp64
// arithmetic of m x nq rows costs nothing.
template <int SPACE, int PF>
__global__ __launch_bounds__(256) void prefix_exact_kernel(const float* __restrict__ X, const float* __restrict__ rn,
                                                           const float* __restrict__ Qpad, const double* __restrict__ qaux,
                                                           int32_t m, int32_t ld, double* __restrict__ out64) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = reinterpret_cast<double*>(smem);  // [ld]
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int nw = blockDim.x >> 6;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) qs[c] = (double)Qpad[(int64_t)q * ld + c];
    __syncthreads();
    const double qinv = qaux[q];
    for (int j0 = (blockIdx.y * nw + wave) * 16; j0 < m; j0 += gridDim.y * nw * 16) {
        const int j = j0 + r;
        const bool have = j < m;
        const int64_t rr = have ? j : 0;
        const float* base[1] = {X + (rr >> 4) * (int64_t)(kPanelRows * ld) + (rr & 15) * 16 + g * 4};
        double acc[1][1], nx[1];
        accumulate_rows<SPACE, 1, 1, PF>(base, qs, ld, g, acc, nx);
        if (have && lane < 16) {
            const float nrm = rn[rr];
            const double d = finish_distance<SPACE>(acc[0][0], nx[0], qinv);
            out64[(int64_t)q * m + j] = nrm == nrm && d == d ? d : __builtin_inf();
        }
    }
}

hipError_t launch_prefix_exact(const float* X, const float* rn, const float* Qpad, const double* qaux, int32_t nq, int32_t m,
                               int32_t ld, int32_t space, double* d64, const Tuning& tn, hipStream_t s) {
    if (nq <= 0 || m <= 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)ld * sizeof(double);
    // a whole 768-column row in flight per lane (24 groups x two banks) when the row is a whole number of such halves:
    // one round trip per row instead of three (the kernel is one latency chain per wave); MLVDB_PREFIX_PF=8: tuning
    const bool pf24 = (ld / 16) % 24 == 0 && tn.prefix_pf == 24;
    const int nw = std::max(1, std::min(4, tn.prefix_waves));  // waves per block, 16 rows each
    const dim3 grid((unsigned)nq, (unsigned)(((int64_t)m + 16 * nw - 1) / (16 * nw)));
    hipError_t e = hipSuccess;
#define MLVDB_LAUNCH_PREFIX(SP)                                                                                        \
    do {                                                                                                               \
        auto kern = pf24 ? prefix_exact_kernel<SP, 24> : prefix_exact_kernel<SP, 8>;                                   \
        if (lds > 48 * 1024)                                                                                           \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) kern<<<grid, nw * 64, lds, s>>>(X, rn, Qpad, qaux, m, ld, d64);                               \
    } while (0)
    switch (space) {
        case kSpaceL2: MLVDB_LAUNCH_PREFIX(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_PREFIX(kSpaceCosine); break;
        default: MLVDB_LAUNCH_PREFIX(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_PREFIX
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_pair_distances(const float* X, const float* Qpad, const double* qaux, const int64_t* labels, int32_t nq,
                                 int32_t m, int32_t ld, int32_t space, double* out64, float* out32, hipStream_t s) {
    if (nq <= 0 || m <= 0) return hipSuccess;
    const size_t lds = (size_t)ld * sizeof(double);
    const dim3 grid((unsigned)nq, (unsigned)std::min<int64_t>(64, ((int64_t)m + 63) / 64));
    hipError_t e = hipSuccess;
#define MLVDB_LAUNCH_PAIRS(SP)                                                                                         \
    do {                                                                                                               \
        auto kern = pair_distance_kernel<SP>;                                                                          \
        if (lds > 48 * 1024)                                                                                           \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) kern<<<grid, 256, lds, s>>>(X, Qpad, qaux, labels, m, ld, out64, out32);                   \
    } while (0)
    switch (space) {
        case kSpaceL2: MLVDB_LAUNCH_PAIRS(kSpaceL2); break;
        case kSpaceCosine: MLVDB_LAUNCH_PAIRS(kSpaceCosine); break;
        default: MLVDB_LAUNCH_PAIRS(kSpaceIp); break;
    }
#undef MLVDB_LAUNCH_PAIRS
    return e != hipSuccess ? e : hipGetLastError();
}

ExactPlan plan_exact(int64_t nrows, int32_t ld, int32_t nq_sel, int32_t k, const Tuning& tn) {
    ExactPlan p;
    int qt = nq_sel >= 8 ? 8 : nq_sel >= 4 ? 4 : nq_sel >= 2 ? 2 : 1;
    while (qt > 1 && (size_t)qt * ld * sizeof(double) > 64 * 1024) qt >>= 1;
    p.qt = qt;
    const int nw = qt <= 2 ? 16 : 8;
    const int pw = qt == 4 ? 4 : 2;
    p.threads = nw * 64;
    p.nqtiles = (nq_sel + qt - 1) / qt;
    const int64_t npanels = (nrows + 15) / 16;
    const int64_t ntasks = (npanels + pw - 1) / pw;
    int64_t nblk = (ntasks + nw - 1) / nw;
    // one block per CU is the sweet spot for the streaming scan (tools/exact_ab.py); with several query
    // tiles the corpus is split over fewer blocks each
    int64_t cap = std::min<int64_t>(256, 1024 / p.nqtiles);
    if (tn.exact_nblk > 0) cap = tn.exact_nblk;  // tuning experiments only
    if (cap < 8) cap = 8;
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    p.nblk = (int)nblk;
    const size_t q_bytes = (size_t)qt * ld * sizeof(double);
    const size_t m_bytes = (size_t)nw * qt * 64 * (sizeof(double) + sizeof(int32_t));
    p.lds_bytes = q_bytes > m_bytes ? q_bytes : m_bytes;
    (void)k;
    return p;
}

template <int SPACE, int QT, int PW, int NW, bool NT = false>
static hipError_t launch_one(const ExactArgs& a, const ExactPlan& p, hipStream_t s) {
    auto kern = exact_scan_kernel<SPACE, QT, PW, NW, NT>;
    if (p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    kern<<<dim3(p.nblk, p.nqtiles), p.threads, p.lds_bytes, s>>>(a, p.nblk);
    return hipGetLastError();
}

template <int SPACE>
static hipError_t launch_space(const ExactArgs& a, const ExactPlan& p, hipStream_t s) {
    // non-temporal corpus loads (every row is read once per launch); MLVDB_EXACT_NT=0 selects plain loads for the
    // batch-1 shape only (A/B: tools/exact_ab.py)
    const bool plain = a.tn && a.tn->exact_nt == 0;
    if (p.qt == 1 && plain) return launch_one<SPACE, 1, 2, 16, false>(a, p, s);
    switch (p.qt) {
        case 1: return launch_one<SPACE, 1, 2, 16, true>(a, p, s);
        case 2: return launch_one<SPACE, 2, 2, 16, true>(a, p, s);
        case 4: return launch_one<SPACE, 4, 4, 8, true>(a, p, s);
        default: return launch_one<SPACE, 8, 2, 8, true>(a, p, s);
    }
}

hipError_t launch_exact_scan(const ExactArgs& a, const ExactPlan& p, hipStream_t s) {
    if (a.nq_sel <= 0) return hipSuccess;
    switch (a.space) {
        case kSpaceL2: return launch_space<kSpaceL2>(a, p, s);
        case kSpaceCosine: return launch_space<kSpaceCosine>(a, p, s);
        default: return launch_space<kSpaceIp>(a, p, s);
    }
}

hipError_t launch_exact_merge(const TopEntry* partial, int32_t nq_sel, const int32_t* nq_sel_dev, const int32_t* qsel,
                              int32_t nblk, int32_t k, int64_t* out_labels, float* out_dist, int32_t* out_counts,
                              double* out_d64, hipStream_t s) {
    if (nq_sel <= 0) return hipSuccess;
    exact_merge_kernel<<<nq_sel, 256, 0, s>>>(partial, nq_sel_dev, qsel, nblk, k, out_labels, out_dist, out_counts,
                                              out_d64);
    return hipGetLastError();
}

}  // namespace mlvdb
