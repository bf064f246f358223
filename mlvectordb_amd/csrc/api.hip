// C ABI of libmlvdb_hip.so (declared in include/mlvdb_hip.h): index handle, HBM management,
// and the orchestration of the scan kernels.  Nothing here touches the CPU for arithmetic:
// if no HIP device is usable every entry point fails with MLVDB_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>

#include "internal.h"

using namespace mlvdb;

namespace {

thread_local std::string g_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = need;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

// Pinned host staging (grow only): a hipMemcpyAsync from / to pageable memory goes through the runtime's own staging with
// a synchronisation per copy -- four small D2H copies and one 786 KB H2D copy cost a 256-query wave 0.33 ms on top of its
// 2.0 ms of kernels (round 2: p50_ms_per_wave_host_io 2.349 vs 2.016 device-resident).
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = need + need / 4;
        hipError_t e = hipHostMalloc(&p, want, 0);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct mlvdb_index {
    int device = 0;
    int32_t dim = 0, ld = 0, space = 0;
    int32_t ld8 = 0;      // width of the int8 shadow: round_up(ld, 256), zero padded; 0 = this index keeps none
    int32_t strategy = MLVDB_STRATEGY_AUTO;
    Tuning tn;            // tuning state: from the environment at creation, mlvdb_index_set_tuning afterwards; never getenv later
    bool profiling = false;
    float* X = nullptr;   // panels, capacity * ld floats
    void* Xb = nullptr;   // bf16 shadow of X for the filter scan (capacity * ld bf16), or nullptr
    bool shadow = false;  // keep the bf16 shadow (decided at creation: dim % 64 == 0 and not disabled)
    bool i8_only = false; // ld8 > 0 (default; SHADOW_BF16 at creation keeps both): no bf16 shadow, the
                          // int8 one serves every pass (1.25x instead of 1.75x the corpus in HBM); only an index whose rows
                          // quantise too badly for int8 bounds (l2 / ip, rmax8 > 0.03) converts the fp32 rows in registers
    float* rn = nullptr;  // row norms, NaN = tombstoned / not a row
    int64_t capacity = 0, total = 0, deleted = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // mlvdb_index_get_rows_at: a hit-enrichment gather must not queue behind the next scan
    DevBuf gather_out, gather_lab;     // its private buffers (a search may be running on `stream` from another host thread)
    std::mutex aux_mutex;              // ... shared by the two entry points that use it (get_rows_at, pair_distances): a consumer
                                       // thread enriching wave i and a caller scoring pairs must not swap the buffers under each other
    // workspaces (grow only)
    DevBuf stage, qpad, qaux, partial, qsel, seed_lab, seed_dist, seed_cnt, seed_d64;
    DevBuf cand_range, rhits, rhit_cnt;  // range passes: larger candidate lists, exact hits, hit counts
    DevBuf row_mask, rn_masked;  // filtered search
    DevBuf qerr, rowerr;         // rounding errors of the bf16 images: per query / maximum over the rows (device scalar)
    // experimental int8 shadow (MLVDB_I8=1, cosine, ld % 256 == 0): built lazily at search time, rebuilt after any mutation
    DevBuf x8, rp8, rowerr8, qimg8, sq8;
    DevBuf l2tag;  // l2: which pass scale the offsets plane behind rp8 ([0..3]) / rp8_masked ([4..7]) was computed for: filter_l2_offsets_kernel
    int64_t i8_rows = 0;      // rows [0, i8_rows) of the int8 shadow are current (0 after compact / reset / regrowth)
    float i8_err = 0.f;       // host copy of rowerr8 (read back whenever rows were converted)
    uint32_t i8_odd_groups = 0;  // ... and of its third word: 8-row scale groups whose largest row error exceeds 0.03
    bool sqmin_fresh = false;  // fmisc was just (re)allocated: FilterArgs::sqmin[] not initialised yet
    bool mask_active = false;  // h->rn is a masked copy (mlvdb_search_batch_filtered)
    bool mask_pairs_ready = false;  // ... and rp8_masked holds the masked copy of the int8 shadow's row pairs
    DevBuf rp8_masked;
    // fp16 row-major shadow for the mid bounds (kernels_refine.hip): built lazily by the first range query / top_k > 64 search
    DevBuf x16, s16, rowerr16, picks, npicks;
    int64_t l2_rows = 0;      // rows [0, l2_rows) of the fp16 shadow are current (0 after compact / reset / regrowth)
    bool l2_failed = false;   // its allocation failed once (HBM full): the callers fall back, nobody retries per call
    DevBuf qimg, fmisc, cand, rescr, wgbuf, wgcnt, io_q, io_lab, io_dist, io_cnt, io_d64, counters, labels_in;
    DevBuf page_lab, page_dist, page_cnt, page_d64, cur_d, cur_l;  // top_k > MLVDB_MAX_TOPK paging
    PinBuf pin_in, pin_out;          // pinned staging of the host-pointer entries (queries in; labels / distances / counts out)
    DevBuf io_out;                   // one device buffer for all outputs of a host-pointer search: one D2H copy
    bool flags_in_out = false;       // search_host: the deferred overflow flags travel behind the outputs (flags_out, device)
    uint32_t* flags_out = nullptr;
    uint32_t* host_flags = nullptr;  // pinned, kFilterQueries words
    bool deferred = false;           // search_host: the pass left its overflow flags in host_flags instead of launching the
    FilterArgs deferred_fa{};        //   exact fallback; its arguments, for the (rare) fallback after the sync
    int64_t host_fallbacks = 0;      // fallback queries decided on the host (added to the device-side count in the stats)
    bool host_overflow[256] = {};    // flags of the last collect_overflow
    std::string err;
    // statistics / profiling
    mlvdb_stats stats{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> scan_events;
    size_t scan_events_used = 0;
    hipEvent_t total_events[2] = {nullptr, nullptr};
    bool stats_pending = false;
    hipStream_t counters_stream = nullptr;  // stream of the last call that wrote the device counters
    bool counters_pending = false;
};

namespace {

int fail(mlvdb_index* h, int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (h)
        h->err = buf;
    else
        g_error = buf;
    return code;
}

#define HIP_TRY(h, call)                                                                    \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail(h, e__ == hipErrorOutOfMemory ? MLVDB_ERR_OUT_OF_MEMORY : MLVDB_ERR_HIP, #call, e__); \
    } while (0)

// ---- tuning state: name table, the one environment read, KEY=VAL parsing
struct TuningField {
    const char* name;
    int Tuning::*field;
};
const TuningField kTuningFields[] = {
#define X(field, name, dflt) {name, &Tuning::field},
    MLVDB_TUNING_FIELDS(X)
#undef X
};

// mlvdb_index_create only: MLVDB_<NAME> for every field of the table (internal.h); MLVDB_SHADOW=bf16 is the historical
// spelling of MLVDB_SHADOW_BF16=1.  Nothing else in the library reads the environment.
Tuning tuning_from_env() {
    Tuning t;
    char key[80];
    for (const TuningField& f : kTuningFields) {
        snprintf(key, sizeof key, "MLVDB_%s", f.name);
        if (const char* v = getenv(key)) t.*(f.field) = atoi(v);
        if (!strcmp(f.name, "SHADOW_BF16"))
            if (const char* v = getenv("MLVDB_SHADOW")) t.shadow_bf16 = !strcmp(v, "bf16") ? 1 : t.shadow_bf16;
    }
    return t;
}

const TuningField* find_tuning_field(const char* key, size_t len) {
    if (len > 6 && !strncmp(key, "MLVDB_", 6)) {
        key += 6;
        len -= 6;
    }
    for (const TuningField& f : kTuningFields)
        if (strlen(f.name) == len && !strncmp(f.name, key, len)) return &f;
    return nullptr;
}

int reserve_rows(mlvdb_index* h, int64_t rows) {
    if (!h->rowerr.p) {  // largest relative bf16 rounding error of any row appended so far (0 = no rows)
        HIP_TRY(h, h->rowerr.ensure(sizeof(unsigned int)));
        HIP_TRY(h, hipMemsetAsync(h->rowerr.p, 0, sizeof(unsigned int), h->stream));
    }
    if (rows <= h->capacity) return MLVDB_OK;
    int64_t want = rows;
    if (h->capacity > 0 && want < h->capacity + h->capacity / 2) want = h->capacity + h->capacity / 2;
    const int64_t cap = round_up_rows(want);
    float* nX = nullptr;
    float* nrn = nullptr;
    void* nXb = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&nX), (size_t)cap * h->ld * sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nrn), (size_t)cap * sizeof(float));
    if (e == hipSuccess && h->shadow) e = hipMalloc(&nXb, (size_t)cap * h->ld * 2);
    if (e != hipSuccess) {
        (void)hipFree(nX);
        if (nrn) (void)hipFree(nrn);
        return fail(h, MLVDB_ERR_OUT_OF_MEMORY, "hipMalloc(row norms / bf16 shadow)", e);
    }
    const size_t used_floats = (size_t)((h->total + 15) / 16) * 16 * h->ld;
    HIP_TRY(h, hipMemsetAsync(nX + used_floats, 0, ((size_t)cap * h->ld - used_floats) * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(nrn, 0xFF, (size_t)cap * sizeof(float), h->stream));  // 0xFFFFFFFF = NaN
    if (nXb) {
        HIP_TRY(h, hipMemsetAsync(static_cast<char*>(nXb) + used_floats * 2, 0, ((size_t)cap * h->ld - used_floats) * 2,
                                  h->stream));
        if (h->total > 0)
            HIP_TRY(h, hipMemcpyAsync(nXb, h->Xb, used_floats * 2, hipMemcpyDeviceToDevice, h->stream));
    }
    if (h->total > 0) {
        HIP_TRY(h, hipMemcpyAsync(nX, h->X, used_floats * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(nrn, h->rn, (size_t)h->total * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->X) (void)hipFree(h->X);
    if (h->rn) (void)hipFree(h->rn);
    if (h->Xb) (void)hipFree(h->Xb);
    h->X = nX;
    h->rn = nrn;
    h->Xb = nXb;
    h->capacity = cap;
    return MLVDB_OK;
}

// ---- profiling helpers
// Statistics accumulate over calls until mlvdb_index_last_stats reads (and resets) them, so a
// caller can enqueue many query waves without synchronising and still get per-kernel times.
// Timing-only events (profiling): created without the system-scope fence an event record otherwise carries -- the cache
// write-back / invalidate it costs sits between the kernels of the pass being measured (hip_runtime_api.h,
// hipEventDisableSystemFence: "events that are only being used to measure timing").  Nothing synchronises-with these events:
// results are read after a stream synchronisation.  Tuning EVENT_FENCE=1: default events (A/B).
static hipError_t create_timing_event(hipEvent_t* ev, bool fence) {
    return fence ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableSystemFence);
}

int begin_call(mlvdb_index* h, hipStream_t s) {
    if (h->scan_events_used > 8192) {  // nobody is reading them: start over rather than grow forever
        h->scan_events_used = 0;
        h->stats = mlvdb_stats{};
        h->stats_pending = false;
    }
    if (h->profiling && !h->stats_pending) {
        for (auto& ev : h->total_events)
            if (!ev) HIP_TRY(h, create_timing_event(&ev, h->tn.event_fence != 0));
        HIP_TRY(h, hipEventRecord(h->total_events[0], s));
    }
    return MLVDB_OK;
}

int end_call(mlvdb_index* h, hipStream_t s) {
    if (h->profiling) {
        HIP_TRY(h, hipEventRecord(h->total_events[1], s));
        h->stats_pending = true;
    }
    return MLVDB_OK;
}

int scan_event(mlvdb_index* h, hipStream_t s, bool start) {
    if (!h->profiling) return MLVDB_OK;
    if (start) {
        if (h->scan_events_used == h->scan_events.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(h, create_timing_event(&a, h->tn.event_fence != 0));
            HIP_TRY(h, create_timing_event(&b, h->tn.event_fence != 0));
            h->scan_events.emplace_back(a, b);
        }
        HIP_TRY(h, hipEventRecord(h->scan_events[h->scan_events_used].first, s));
    } else {
        HIP_TRY(h, hipEventRecord(h->scan_events[h->scan_events_used].second, s));
        ++h->scan_events_used;
    }
    return MLVDB_OK;
}

// ---- exact scan of rows [row_begin,row_end) for a set of queries of the prepared batch.
// Qpad/qaux point at query 0 of the set's index space; qsel (device) lists the members or is null.
int run_exact(mlvdb_index* h, hipStream_t s, const float* Qpad, const double* qaux, int32_t nq_sel,
              const int32_t* qsel, int64_t row_begin, int64_t row_end, int32_t k, int64_t* out_labels,
              float* out_dist, int32_t* out_counts, double* out_d64, bool is_main_scan,
              const int32_t* nq_sel_dev = nullptr, const double* cursor_d = nullptr,
              const int32_t* cursor_l = nullptr) {
    if (nq_sel <= 0) return MLVDB_OK;
    ExactPlan plan = plan_exact(row_end - row_begin, h->ld, nq_sel, k, h->tn);
    if (nq_sel_dev) {
        // device-decided fallback: usually zero or a few queries are selected, so spread each query
        // tile over many blocks; blocks of unselected tiles exit at once
        const int64_t by_mem = (int64_t)(64u << 20) / ((int64_t)nq_sel * k * (int64_t)sizeof(TopEntry));
        plan.nblk = (int)std::max<int64_t>(8, std::min<int64_t>(256, by_mem));
    }
    HIP_TRY(h, h->partial.ensure((size_t)nq_sel * plan.nblk * k * sizeof(TopEntry)));
    ExactArgs a{};
    a.X = h->X;
    a.rn = h->rn;
    a.row_begin = row_begin;
    a.row_end = row_end;
    a.ld = h->ld;
    a.space = h->space;
    a.Qpad = Qpad;
    a.qaux = qaux;
    a.qsel = qsel;
    a.nq_sel = nq_sel;
    a.nq_sel_dev = nq_sel_dev;
    a.k = k;
    a.cursor_d = cursor_d;
    a.cursor_l = cursor_l;
    a.partial = h->partial.as<TopEntry>();
    a.tn = &h->tn;
    if (is_main_scan) {
        int rc = scan_event(h, s, true);
        if (rc) return rc;
    }
    HIP_TRY(h, launch_exact_scan(a, plan, s));
    if (is_main_scan) {
        int rc = scan_event(h, s, false);
        if (rc) return rc;
        h->stats.scan_launches += 1;
        h->stats.rows_scanned += (row_end - row_begin) * plan.nqtiles;
    }
    HIP_TRY(h, launch_exact_merge(a.partial, nq_sel, nq_sel_dev, qsel, plan.nblk, k, out_labels, out_dist, out_counts,
                                  out_d64, s));
    return MLVDB_OK;
}

__global__ void fill_empty_kernel(int64_t* labels, float* dist, int32_t* counts, double* d64, int64_t nq, int32_t k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq * k) {
        labels[i] = -1;
        dist[i] = __builtin_inff();
        if (d64) d64[i] = __builtin_inf();
    }
    if (i < nq) counts[i] = 0;
}

__global__ void copy_words_kernel(const uint32_t* src, uint32_t* dst) { dst[threadIdx.x] = src[threadIdx.x]; }

int setup_filter_ws(mlvdb_index* h, FilterArgs& fa, const float* Qpad, const double* qaux, const float* qerr, int32_t nq) {
    HIP_TRY(h, h->qimg.ensure(filter_qimg_bytes((h->ld + 63) / 64 * 64)));
    {
        const void* before = h->fmisc.p;
        HIP_TRY(h, h->fmisc.ensure(9 * kFilterQueries * sizeof(uint32_t)));
        if (h->fmisc.p != before) h->sqmin_fresh = true;  // the two scalars the fused prep's atomics start from: see run_filter_pass
    }
    HIP_TRY(h, h->cand.ensure((size_t)kFilterQueries * kCandCap * sizeof(CandEntry)));
    HIP_TRY(h, h->rescr.ensure((size_t)kFilterQueries * kCandCap * sizeof(RangeHit)));  // exact scores of the rescored candidates
    if (h->Xb || h->i8_only) {  // the assembly scan appends through workgroup-private buffers
        HIP_TRY(h, h->wgbuf.ensure((size_t)kScanMaxGrid * kWgCap * sizeof(WgEntry)));
        HIP_TRY(h, h->wgcnt.ensure((size_t)kScanMaxGrid * 8 * sizeof(uint32_t)));
    }
    if (!h->host_flags)
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->host_flags), kFilterQueries * sizeof(uint32_t), 0));
    fa.tn = &h->tn;
    fa.X = h->X;
    fa.Xb = h->Xb;
    fa.rn = h->rn;
    fa.total = h->total;
    fa.ld = h->ld;
    fa.ld8 = h->ld8;
    fa.space = h->space;
    fa.Qpad = Qpad;
    fa.qaux = qaux;
    fa.qerr = qerr;
    fa.row_err = h->rowerr.as<float>();
    fa.nq = nq;
    fa.qimg = h->qimg.p;
    fa.qscale = h->fmisc.as<float>();
    fa.thr = h->fmisc.as<float>() + kFilterQueries;
    fa.cnt = h->fmisc.as<uint32_t>() + 2 * kFilterQueries;
    fa.overflow = h->fmisc.as<uint32_t>() + 3 * kFilterQueries;
    fa.ke = h->fmisc.as<float>() + 4 * kFilterQueries;
    fa.keb = h->fmisc.as<float>() + 5 * kFilterQueries;
    fa.ke8 = h->fmisc.as<float>() + 6 * kFilterQueries;  // 257 floats
    fa.sqmin = h->fmisc.as<uint32_t>() + 7 * kFilterQueries + 128;  // [0..1] smallest scale / largest error (cosine); [2..3] l2c QMAX by parity
    fa.l2c_out = h->fmisc.as<float>() + 7 * kFilterQueries + 132;
    fa.rmaxq = h->fmisc.as<float>() + 8 * kFilterQueries;
    fa.rs = h->rescr.as<RangeHit>();
    fa.cand = h->cand.as<CandEntry>();
    fa.cand_cap = kCandCap;
    fa.wgbuf = h->wgbuf.as<WgEntry>();
    fa.wgcnt = h->wgcnt.as<uint32_t>();
    return MLVDB_OK;
}

// Collect the overflowed queries of a pass on the host; returns their count (device list in h->qsel).
int collect_overflow(mlvdb_index* h, hipStream_t s, const FilterArgs& fa, int32_t* n_flagged) {
    HIP_TRY(h, hipStreamSynchronize(s));  // (a D2H copy enqueued behind a long kernel parks in the copy queue: search_host)
    HIP_TRY(h, hipMemcpyAsync(h->host_flags, fa.overflow, kFilterQueries * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    int32_t sel[kFilterQueries];
    int32_t n = 0;
    for (int q = 0; q < kFilterQueries; ++q) h->host_overflow[q] = q < fa.nq && h->host_flags[q] != 0;
    for (int q = 0; q < fa.nq; ++q)
        if (h->host_flags[q]) sel[n++] = q;
    *n_flagged = n;
    if (n) {
        HIP_TRY(h, h->qsel.ensure(kFilterQueries * sizeof(int32_t)));
        HIP_TRY(h, hipMemcpyAsync(h->qsel.p, sel, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipStreamSynchronize(s));  // `sel` is on this stack frame
    }
    return MLVDB_OK;
}

// int8 shadow of a cosine index (kernels_filter.hip, "int8 shadow"): kept current lazily -- rows appended since the
// last pass are converted here, tombstones are patched in by mlvdb_index_tombstone, compaction / reset / regrowth
// start it over.  MLVDB_I8=0 keeps the pass on the bf16 shadow.
// (l2: only the |x| slot -- the x-slot of an l2 pair holds its group's scale or error and must survive the row: shadow8_rows_kernel)
__global__ void tombstone_rp8_kernel(const int64_t* labels, int64_t n, float* rp8, int64_t rows, int l2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && labels[i] >= 0 && labels[i] < rows) {
        rp8[2 * labels[i] + 1] = __builtin_nanf("");
        if (!l2) rp8[2 * labels[i]] = __builtin_nanf("");
    }
}

bool i8_bounds_usable(const mlvdb_index* h);

bool i8_eligible(const mlvdb_index* h) {
    return h->tn.i8 != 0 && (h->Xb || h->i8_only) && h->ld8 > 0;  // (tools/scan_ab.py switches I8 inside one process: set_tuning)
}

// The l2 offsets plane of a pair buffer is valid for the pairs it was computed from: whatever rewrites pairs forgets the tag
// (slot 0: rp8, slot 1: rp8_masked); the next pass's filter_l2_offsets_kernel then recomputes the plane.
int forget_l2_offsets(mlvdb_index* h, int slot, hipStream_t s) {
    if (h->space != kSpaceL2) return MLVDB_OK;
    if (!h->l2tag.p) {
        HIP_TRY(h, h->l2tag.ensure(8 * sizeof(uint32_t)));
        HIP_TRY(h, hipMemsetAsync(h->l2tag.p, 0, 8 * sizeof(uint32_t), s));
        return MLVDB_OK;
    }
    HIP_TRY(h, hipMemsetAsync(h->l2tag.as<uint32_t>() + 4 * slot, 0, 4 * sizeof(uint32_t), s));
    return MLVDB_OK;
}

// Bring the int8 shadow up to date (rows appended since the last pass).  Must run with the index's own norms in h->rn
// (a row-mask search swaps them for a masked copy afterwards).
int update_i8_shadow(mlvdb_index* h, hipStream_t s) {
    // (l2: a third float-sized word per row behind the pairs: the integer offsets of the folded admission test, per pass)
    const size_t need_x8 = (size_t)h->capacity * h->ld8, need_rp = (size_t)h->capacity * (h->space == kSpaceL2 ? 3 : 2) * sizeof(float);
    if (h->x8.bytes < need_x8 || h->rp8.bytes < need_rp || !h->rowerr8.p) {
        HIP_TRY(h, h->x8.ensure(need_x8));
        HIP_TRY(h, h->rp8.ensure(need_rp));
        HIP_TRY(h, h->rowerr8.ensure(4 * sizeof(float)));  // {largest relative row error, smallest row norm, odd groups (u32), -}
        h->i8_rows = 0;
    }
    if (h->i8_rows == 0) {
        HIP_TRY(h, hipMemsetAsync(h->x8.p, 0, need_x8, s));
        HIP_TRY(h, hipMemsetAsync(h->rp8.p, 0xff, need_rp, s));  // NaN: not a row
        if (int rc = forget_l2_offsets(h, 0, s)) return rc;
        HIP_TRY(h, hipMemsetAsync(h->rowerr8.p, 0, sizeof(float), s));
        HIP_TRY(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->rowerr8.as<float>() + 1), 0x7f800000, 1, s));  // +inf
        HIP_TRY(h, hipMemsetAsync(h->rowerr8.as<float>() + 2, 0, sizeof(uint32_t), s));
    }
    if (h->i8_rows < h->total) {
        if (int rc = forget_l2_offsets(h, 0, s)) return rc;
        HIP_TRY(h, launch_shadow8_rows(h->X, h->rn, h->x8.p, h->rp8.as<float>(), h->rowerr8.as<float>(), h->i8_rows, h->total,
                                       h->ld, h->ld8, h->space, s));
        h->i8_rows = h->total;
        uint32_t words[3] = {0u, 0u, 0u};
        HIP_TRY(h, hipMemcpyAsync(words, h->rowerr8.p, sizeof words, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        std::memcpy(&h->i8_err, &words[0], sizeof(float));
        h->i8_odd_groups = words[2];
    }
    return MLVDB_OK;
}

int attach_i8(mlvdb_index* h, hipStream_t s, FilterArgs& fa) {
    if (!i8_eligible(h)) return MLVDB_OK;
    if (h->mask_active) {
        if (!h->mask_pairs_ready) return MLVDB_OK;  // the masked copy of the row pairs was not built: bf16 / fp32 bodies
    } else {
        int rc = update_i8_shadow(h, s);
        if (rc) return rc;
    }
    // One scale per row: a row with an outlier component quantises badly.  Cosine bounds carry every row's own error;
    // l2 / ip still use the index-wide maximum, which would then admit everything: beyond 0.03 (typical data sits at
    // 0.008-0.015) they keep to the bf16 shadow, whose error is relative per component.
    if (!i8_bounds_usable(h)) return MLVDB_OK;
    HIP_TRY(h, h->qimg8.ensure((size_t)kFilterQueries * h->ld8));
    HIP_TRY(h, h->sq8.ensure(kFilterQueries * sizeof(float)));
    fa.X8 = h->x8.p;
    fa.rp8 = h->mask_active ? h->rp8_masked.as<float>() : h->rp8.as<float>();  // a masked-out row is a NaN pair: "not a row"
    // l2: pairs + offsets through one buffer descriptor (32-bit offsets): 12 bytes per row must stay below 4 GB
    fa.rp8_cap = h->space == kSpaceL2 ? h->capacity : 0;  // (l2_int8_ok held: i8_bounds_usable)
    if (fa.rp8_cap > 0) {
        fa.l2c = 1;
        if (!h->l2tag.p) {
            int rc = forget_l2_offsets(h, 0, s);  // (allocates, zeroed)
            if (rc) return rc;
        }
        fa.l2tag = h->tn.l2_offset_cache != 0 ? h->l2tag.as<uint32_t>() + (h->mask_active ? 4 : 0) : nullptr;
    }
    fa.row_err8 = h->rowerr8.as<float>();
    fa.qimg8 = h->qimg8.p;
    fa.sq8 = h->sq8.as<float>();
    return MLVDB_OK;
}

// One pass of <= 256 queries through the filter path; outputs at query index q0.. of the batch.
int finish_filter_pass(mlvdb_index* h, hipStream_t s, FilterArgs& fa, int32_t q0, int32_t nq, int32_t k, int64_t* out_labels,
                       float* out_dist, int32_t* out_counts, double* out_d64, bool defer_fallback, bool ranked = false);

// Everything a filter pass needs of its queries, in one fused launch (+ the one-block fin for int8 passes of several queries).
int prep_pass(mlvdb_index* h, hipStream_t s, const FilterArgs& fa, const float* queries_raw, float* Qpad, double* qaux, float* qerr) {
    if (h->sqmin_fresh) {  // what the fused prep's atomicMin / atomicMax start from; afterwards every fin kernel restores it
        HIP_TRY(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fa.sqmin), 0x7f7f7f7f, 1, s));
        HIP_TRY(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fa.sqmin + 1), 0, 1, s));
        h->sqmin_fresh = false;
    }
    HIP_TRY(h, launch_filter_prep_fused(fa, queries_raw, h->dim, Qpad, qaux, qerr, s));
    if (fa.X8 && fa.rp8_cap > 0) {
        // l2: this pass's integer offsets (they depend on its largest query scale and on which rows are live): 29 us for 10M rows.
        // (Tried: on a side stream beside the seeding pass and the first refine, joined by an event before the first assembly scan
        // -- no gain, 1.971 vs 1.954 ms per wave in same-process A/Bs: profiles/r04/scan_ab_l2*_10m.txt; it stays in the pass's stream.)
        const int64_t rows = std::min<int64_t>(h->capacity, (h->total + kFilterTile - 1) / kFilterTile * kFilterTile);
        HIP_TRY(h, launch_filter_l2_offsets(fa, rows, s));
    }
    return MLVDB_OK;
}

// `queries_raw`: the pass's queries [nq][dim] as the caller gave them (device); this pass prepares them itself (padded copy,
// norms, images: one fused launch) into Qpad / qaux / qerr at q0.
int run_filter_pass(mlvdb_index* h, hipStream_t s, const float* queries_raw, float* Qpad, double* qaux, int32_t q0, int32_t nq,
                    int32_t k, int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_d64,
                    bool defer_fallback = false) {
    FilterArgs fa{};
    int rc = setup_filter_ws(h, fa, Qpad + (size_t)q0 * h->ld, qaux + q0, h->qerr.as<float>() + q0, nq);
    if (rc) return rc;
    rc = attach_i8(h, s, fa);
    if (rc) return rc;
    rc = prep_pass(h, s, fa, queries_raw + (size_t)q0 * h->dim, Qpad + (size_t)q0 * h->ld, qaux + q0, h->qerr.as<float>() + q0);
    if (rc) return rc;
    h->stats.bound_dtype = fa.X8 ? 2 : 1;
    // (Round 3 had tried a single scan round for <= 8 queries behind the generic exact scan of a 12k-row prefix: slower, 0.273 vs
    // 0.248 ms at batch 1 -- profiles/r03/small_batch_single_round_tried_1m.txt.  Round 4's version for single queries follows.)
    // ---- one round for single queries on small corpora (round 4; SMALL_BATCH=0: the rounds below): the exact k-th best of an
    // 11,520-row prefix (prefix_exact_kernel all over the chip + the one-block selection) puts the threshold at quantile
    // k / 11,520; ONE scan launch over every row then appends ~total (k / 11,520) x band entries per query -- 1M x 768, k = 10:
    // 870 x band (~4 on N(0,1) rows) of the list's 8,192 -- and the fused finish prunes, rescores and ranks: five launches instead
    // of seven (no 65k-row first round, no refine after it): 0.197-0.199 vs 0.206-0.212 ms at batch 1
    // (profiles/r04/small_batch_one_round_vs_rounds_1m.txt).  Taken only while the estimate with band = 6 stays inside the list;
    // a list that overflows all the same sends its query to the exact scan, as everywhere.
    {
        const int small_nq1 = std::max(0, std::min(1, h->tn.small_nq));  // (two queries: 0.214-0.220 vs 0.211-0.219 ms -- no gain)
        const int64_t m = 3 * kSeedRows;
        const bool one_round = h->tn.small_batch != 0 && fa.X8 && nq <= small_nq1 && k <= 64 && !h->mask_active &&
                               filter_refine_can_fuse(fa) && h->tn.small_finish != 0 && h->tn.small_seed != 0 && h->total > 4 * m &&
                               (double)h->total * k * 6.0 <= 6000.0 * (double)m;
        if (one_round) {
            HIP_TRY(h, h->seed_d64.ensure(((size_t)kFilterQueries * 64 + (size_t)8 * kSeedRows) * sizeof(double)));
            double* d64 = h->seed_d64.as<double>();
            HIP_TRY(h, launch_prefix_exact(h->X, fa.rn, fa.Qpad, fa.qaux, nq, (int32_t)m, h->ld, h->space, d64, h->tn, s));
            HIP_TRY(h, launch_filter_prefix_thr(fa, d64, (int32_t)m, k, s));
            rc = scan_event(h, s, true);
            if (rc) return rc;
            ScanInfo info;
            HIP_TRY(h, launch_filter_scan(fa, 0, h->total, s, &info));
            rc = scan_event(h, s, false);
            if (rc) return rc;
            h->stats.scan_launches += 1;
            h->stats.rows_scanned += h->total;
            HIP_TRY(h, h->qsel.ensure(kFilterQueries * sizeof(int32_t)));
            unsigned long long* stats = h->counters.as<unsigned long long>();
            HIP_TRY(h, launch_filter_finish_small(fa, k, q0, out_labels, out_dist, out_counts, out_d64, stats,
                                                  defer_fallback ? nullptr : h->qsel.as<int32_t>(),
                                                  reinterpret_cast<int32_t*>(stats + 2), s));
            return finish_filter_pass(h, s, fa, q0, nq, k, out_labels, out_dist, out_counts, out_d64, defer_fallback, true);
        }
    }
    // seed: a dense pass of the filter kernel over the first rows puts every bound into the lists,
    // the update kernel turns them into thresholds; the remaining rows follow in rounds of growing
    // size so that thresholds tighten early
    // rows of the dense seeding pass: a multiple of kFilterTile (the first scan round starts there), at most kSeedRows
    // (MLVDB_SEED_ROWS, in units of kFilterTile = 768 rows: tuning, read per call)
    const int64_t seed_rows = std::min<int64_t>(kSeedRows, std::max<int64_t>(1, h->tn.seed_rows) * kFilterTile);
    const int64_t n_seed = std::min<int64_t>(h->total, seed_rows);
    int64_t first_row = seed_rows;
    {
        // Tried in round 2 (MLVDB_SEED_EXACT=1): thresholds seeded from the EXACT k-th best score among the first
        // kSeedRows rows (exact fp64 scan of that prefix + merge + one tiny kernel) instead of the dense bf16 pass that
        // puts all 3840 bounds of every query into the lists + the exact-threshold refine over them; the prefix rows
        // then belong to the first scan round.  Slower: 2.16 vs 2.04 ms per 256-query wave, 0.303 vs 0.288 ms at batch 1
        // (3840 x 768 x 256 fp64 multiply-adds are not free); the dense pass stays the default.
        // (l2 index with a few badly quantising rows: the dense int8 pass would bound every seed row with the index-wide error,
        // and those inflated bounds then crowd the refines' picks -- the threshold stalls at the seed's quantile)
        const bool seed_exact = h->tn.seed_exact == 1 || (fa.X8 && h->space == kSpaceL2 && h->i8_err > 0.03f);
        // Batches of 1-2 queries (round 3): the exact k-th best of the prefix by a kernel made for it (one 16-row group per
        // wave all over the chip + a one-block selection of the k-th: 8 + 10 us) instead of the dense int8 pass + exact-threshold refine (7 + 15.5 us
        // of latency chains at batch 1); the prefix rows then belong to the first scan round.  MLVDB_SMALL_SEED=0: the dense pass.
        // (4-8 queries: no gain from either step; profiles/r03/small_batch_fused_finish_and_prefix_seed_1m.txt)
        const int small_nq = std::max(0, std::min(8, h->tn.small_nq));
        if (fa.X8 && nq <= small_nq && k <= 64 && !h->mask_active && !seed_exact && h->tn.small_seed != 0) {
            HIP_TRY(h, h->seed_d64.ensure(((size_t)kFilterQueries * 64 + (size_t)8 * kSeedRows) * sizeof(double)));
            double* d64 = h->seed_d64.as<double>();
            HIP_TRY(h, launch_prefix_exact(h->X, fa.rn, fa.Qpad, fa.qaux, nq, (int32_t)n_seed, h->ld, h->space, d64, h->tn, s));
            HIP_TRY(h, launch_filter_prefix_thr(fa, d64, (int32_t)n_seed, k, s));
            first_row = 0;
        } else if (seed_exact) {
            HIP_TRY(h, h->seed_lab.ensure((size_t)kFilterQueries * k * sizeof(int64_t)));
            HIP_TRY(h, h->seed_dist.ensure((size_t)kFilterQueries * k * sizeof(float)));
            HIP_TRY(h, h->seed_cnt.ensure(kFilterQueries * sizeof(int32_t)));
            HIP_TRY(h, h->seed_d64.ensure((size_t)kFilterQueries * k * sizeof(double)));
            rc = run_exact(h, s, fa.Qpad, fa.qaux, nq, nullptr, 0, n_seed, k, h->seed_lab.as<int64_t>(),
                           h->seed_dist.as<float>(), h->seed_cnt.as<int32_t>(), h->seed_d64.as<double>(), false);
            if (rc) return rc;
            HIP_TRY(h, launch_filter_seed_thr(fa, h->seed_d64.as<double>(), k, s));
            first_row = 0;
        } else {
            HIP_TRY(h, launch_filter_seed_scan(fa, n_seed, k, s));  // includes the first threshold update
        }
    }
    // scan rounds: [seed, r1) [r1, r2) [r2, total), each followed by an exact-threshold refine; in units of kFilterTile rows
    // (MLVDB_ROUND1 / MLVDB_ROUND2: tuning, read per call)
    // r1 = 85: 3840 + 240 tiles of 256 rows -- one tile for (nearly) every CU costs what 192 tiles did (64: +1.5 % per 10M-row
    // wave; 170: the same as 85; r2 = 1024 .. 2389: within noise, 4096: +2.5 %; profiles/r02/scan_ab_round_sizes_10m.txt)
    const int64_t r1 = std::max<int64_t>(6, h->tn.round1), r2 = std::max<int64_t>(r1, h->tn.round2);
    const int64_t bounds[] = {first_row, (int64_t)kFilterTile * r1, (int64_t)kFilterTile * r2, h->total};
    // Batches of 1-2 queries: the refine after the LAST round also rescores and ranks (one launch instead of three:
    // launch_filter_finish_small); MLVDB_SMALL_FINISH=0: the three kernels
    // (one block per query: at 4-8 queries the rescoring kernel's spread over the whole chip wins again -- 0.261 vs 0.247 ms at 4)
    const int small_nq = std::max(0, std::min(8, h->tn.small_nq));
    const bool small_finish = fa.X8 && nq <= small_nq && k <= 64 && filter_refine_can_fuse(fa) && h->tn.small_finish != 0;
    bool ranked = false;
    for (int r = 0; r < 3; ++r) {
        const int64_t b = std::min(bounds[r], h->total), e = std::min(bounds[r + 1], h->total);
        if (e <= b) continue;
        rc = scan_event(h, s, true);
        if (rc) return rc;
        ScanInfo info;
        HIP_TRY(h, launch_filter_scan(fa, b, e, s, &info));
        rc = scan_event(h, s, false);
        if (rc) return rc;
        if (h->tn.debug_entries) {  // tuning aid: entries appended by this scan launch (synchronises the stream)
            std::vector<uint32_t> wc((size_t)kScanMaxGrid * 8, 0u);
            HIP_TRY(h, hipMemcpyAsync(wc.data(), fa.wgcnt, wc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            uint64_t sum = 0, mx = 0;
            const int waves = (int)std::min<int64_t>(256, (e - b + 255) / 256) * 8;  // 8-wave workgroups of 256-row tiles
            for (int i = 0; i < waves; ++i) {
                sum += wc[(size_t)i];
                mx = std::max<uint64_t>(mx, wc[(size_t)i]);
            }
            fprintf(stderr, "[mlvdb] scan rows [%lld, %lld): %llu entries appended (%.1f per query), max per wave %llu\n",
                    (long long)b, (long long)e, (unsigned long long)sum, (double)sum / fa.nq, (unsigned long long)mx);
        }
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1, MLVDB_SCAN_DIAG=234: where a scan launch's time goes (its waves stamp their phases)
        if (fa.wgbuf && h->tn.scan_diag == 234) {
            const int nwg = (int)std::min<int64_t>(256, (e - b + 255) / 256);
            std::vector<unsigned long long> st((size_t)nwg * 8 * 8, 0ull);
            HIP_TRY(h, hipMemcpyAsync(st.data(), fa.wgbuf + (size_t)256 * kWgCap, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            unsigned long long t0 = ~0ull, t5 = 0;
            double ph[3] = {0, 0, 0};
            const double tiles = (double)((e - b + 255) / 256);
            int n = 0;
            for (int w = 0; w < nwg * 8; ++w) {
                const unsigned long long* p = &st[(size_t)w * 8];
                if (!p[0] || !p[5]) continue;
                ++n;
                t0 = std::min(t0, p[0]);
                t5 = std::max(t5, p[5]);
                ph[0] += (double)(p[1] - p[0]) * 0.01;  // C++ preamble
                ph[1] += (double)(p[4] - p[1]) * 0.01;  // the assembly: prologue + tiles
                ph[2] += (double)(p[5] - p[4]) * 0.01;  // scatter tail
            }
            fprintf(stderr, "[mlvdb] scan rows [%lld, %lld): %d waves, first start -> last end %.1f us; mean per wave: preamble %.1f, "
                    "assembly %.1f (%.1f tiles per workgroup), tail %.1f us\n", (long long)b, (long long)e, n,
                    (double)(t5 - t0) * 0.01, ph[0] / n, ph[1] / n, tiles / nwg, ph[2] / n);
        }
#endif
        h->stats.scan_launches += 1;
        h->stats.rows_scanned += e - b;
        // int8 bounds are loose: thresholds from exact scores of the k best bounds; the same kernel prunes the lists
        // (the update kernel's bound-derived threshold could only be lower) unless the query does not fit beside them
        const bool fuse = fa.X8 && filter_refine_can_fuse(fa);
        // (Tried in round 3 for batches of <= 8 queries: no refine after the LAST round -- the rescoring takes the unpruned
        // lists.  Slower: 0.257 vs 0.226 ms at batch 1 on 1M x 768, the ranking kernel pays more for the ~800-entry list
        // than the refine's launch costs; profiles/r03/small_batch_last_refine_1m.txt.)
        if (small_finish && e == h->total) {  // the last round of a small batch
            HIP_TRY(h, h->qsel.ensure(kFilterQueries * sizeof(int32_t)));
            unsigned long long* stats = h->counters.as<unsigned long long>();
            HIP_TRY(h, launch_filter_finish_small(fa, k, q0, out_labels, out_dist, out_counts, out_d64, stats,
                                                  defer_fallback ? nullptr : h->qsel.as<int32_t>(),
                                                  reinterpret_cast<int32_t*>(stats + 2), s));
            ranked = true;
            break;
        }
        if (fa.X8) HIP_TRY(h, launch_filter_refine_thr(fa, k, -1, fuse, s));
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: where the refine kernel's time goes (its blocks stamp their phases)
        if (fa.X8 && fa.wgbuf && h->tn.debug_refine) {
            std::vector<unsigned long long> st((size_t)fa.nq * 8, 0ull);
            HIP_TRY(h, hipMemcpyAsync(st.data(), fa.wgbuf, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            unsigned long long t0 = ~0ull, t5 = 0;
            double ph[5] = {0, 0, 0, 0, 0};
            for (int q = 0; q < fa.nq; ++q) {
                t0 = std::min(t0, st[(size_t)q * 8]);
                t5 = std::max(t5, st[(size_t)q * 8 + 5]);
                for (int i = 0; i < 5; ++i) ph[i] += (double)(st[(size_t)q * 8 + i + 1] - st[(size_t)q * 8 + i]) * 0.01 / fa.nq;
            }
            fprintf(stderr, "[mlvdb] refine rows [%lld, %lld): first start -> last end %.1f us; mean per block: load+keys %.1f, "
                    "select+compact %.1f, gather %.1f, rank+thr %.1f, prune %.1f us\n", (long long)b, (long long)e,
                    (double)(t5 - t0) * 0.01, ph[0], ph[1], ph[2], ph[3], ph[4]);
        }
#endif
        if (!fuse) HIP_TRY(h, launch_filter_update(fa, k, s));
    }
    return finish_filter_pass(h, s, fa, q0, nq, k, out_labels, out_dist, out_counts, out_d64, defer_fallback, ranked);
}

// The end of a kNN pass: exact fp64 rescoring of the candidate lists (unless the last refine did it: ranked), then the exact
// fallback for overflowed queries.
int finish_filter_pass(mlvdb_index* h, hipStream_t s, FilterArgs& fa, int32_t q0, int32_t nq, int32_t k, int64_t* out_labels,
                       float* out_dist, int32_t* out_counts, double* out_d64, bool defer_fallback, bool ranked) {
    int rc = MLVDB_OK;
    // counters: [0] rescored pairs, [1] fallback queries (accumulated over the passes of a call), [2] flag count
    unsigned long long* stats = h->counters.as<unsigned long long>();
#ifdef MLVDB_SCAN_DIAGNOSTICS
    if (fa.wgbuf && h->tn.debug_refine) HIP_TRY(h, hipMemsetAsync(fa.wgbuf, 0, (16384 + 1024) * 8, s));
#endif
    // (its ranking kernel also compacts the overflowed queries for the device-decided fallback below: qsel, nflag)
    HIP_TRY(h, h->qsel.ensure(kFilterQueries * sizeof(int32_t)));
    int32_t* nflag = reinterpret_cast<int32_t*>(stats + 2);
    if (!ranked)
        HIP_TRY(h, launch_filter_rescore(fa, k, q0, out_labels, out_dist, out_counts, out_d64, stats,
                                         defer_fallback ? nullptr : h->qsel.as<int32_t>(), nflag, s));
#ifdef MLVDB_SCAN_DIAGNOSTICS  // make DIAG=1: where the ranking kernel's time goes (its blocks stamp their phases)
    if (!ranked && fa.wgbuf && h->tn.debug_refine) {
        std::vector<unsigned long long> st(16384 + 1024, 0ull);
        HIP_TRY(h, hipMemcpyAsync(st.data(), fa.wgbuf, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        {
            unsigned long long t0 = ~0ull, tl = 0;
            double ph[3] = {0, 0, 0};
            int nw = 0, idle = 0;
            for (int w = 0; w < 4096; ++w) {
                const unsigned long long* p = &st[(size_t)w * 4];
                if (!p[0]) continue;
                t0 = std::min(t0, p[0]);
                if (!p[3]) { ++idle; tl = std::max(tl, p[1]); continue; }
                tl = std::max(tl, p[3]);
                ++nw;
                for (int i = 0; i < 3; ++i) ph[i] += (double)(p[i + 1] - p[i]) * 0.01;
            }
            fprintf(stderr, "[mlvdb] rescoring, score kernel: %d waves with rows, %d without; first start -> last end %.1f us; mean per "
                    "working wave (its last 16-row group): prefix %.1f, query %.1f, gather + score %.1f us\n", nw, idle,
                    (double)(tl - t0) * 0.01, ph[0] / std::max(nw, 1), ph[1] / std::max(nw, 1), ph[2] / std::max(nw, 1));
        }
        unsigned long long r0 = ~0ull, rl = 0, rs1 = 0, rmax = 0;
        double rp[3] = {0, 0, 0};
        for (int q = 0; q < fa.nq; ++q) {
            const unsigned long long* p = &st[16384 + (size_t)q * 4];
            if (!p[0] || !p[3]) continue;
            r0 = std::min(r0, p[0]);
            rl = std::max(rl, p[3]);
            rs1 = std::max(rs1, p[0]);
            rmax = std::max(rmax, p[3] - p[0]);
            for (int i = 0; i < 3; ++i) rp[i] += (double)(p[i + 1] - p[i]) * 0.01 / fa.nq;
        }
        fprintf(stderr, "[mlvdb] rescoring, rank kernel: first start -> last end %.1f us (last start %.1f us after the first, longest "
                "block %.1f); mean per block: load %.1f, rank %.1f, output %.1f us\n",
                (double)(rl - r0) * 0.01, (double)(rs1 - r0) * 0.01, (double)rmax * 0.01, rp[0], rp[1], rp[2]);
    }
#endif
    // overflowed queries (adversarial near-ties) are re-run on the exact scan.  The decision stays on
    // the device: the list is compacted there and the scan's blocks exit at once when it is empty,
    // so the call never waits for the host.
    if (defer_fallback) {
        // host-pointer entry, single pass: the caller synchronises anyway to copy the results out, so the overflow
        // flags ride along to pinned memory and the exact fallback is only launched if a query needs it (search_host)
        // -- three launches fewer on every ordinary call, which is 5 % of a batch-1 call
        if (h->flags_in_out) {  // a kernel on the pass's stream, not a copy-engine job
            copy_words_kernel<<<1, kFilterQueries, 0, s>>>(fa.overflow, h->flags_out);
            HIP_TRY(h, hipGetLastError());
        } else
            HIP_TRY(h, hipMemcpyAsync(h->host_flags, fa.overflow, kFilterQueries * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        h->deferred_fa = fa;
        h->deferred = true;
        return MLVDB_OK;
    }
    rc = run_exact(h, s, fa.Qpad, fa.qaux, nq, h->qsel.as<int32_t>(), 0, h->total, k, out_labels + (size_t)q0 * k,
                   out_dist + (size_t)q0 * k, out_counts + q0, out_d64 ? out_d64 + (size_t)q0 * k : nullptr, false,
                   nflag);
    if (rc) return rc;
    return MLVDB_OK;
}

// ---- the fp16 row-major shadow of the mid bounds (kernels_refine.hip): kept current lazily, like the int8 shadow.
// *m gets X16 == nullptr when the index keeps none (Tuning L2_SHADOW=0, or HBM was full when it was first wanted): the
// callers then do without the second level.
int attach_mid(mlvdb_index* h, hipStream_t s, MidArgs* m) {
    *m = MidArgs{};
    if (!h->tn.l2_shadow || h->l2_failed || h->total == 0) return MLVDB_OK;
    const int32_t ld16 = (h->dim + 63) / 64 * 64;
    const size_t need = (size_t)h->capacity * ld16 * sizeof(uint16_t), need_s = (size_t)h->capacity * sizeof(float);
    if (h->x16.bytes < need || h->s16.bytes < need_s || !h->rowerr16.p) {
        hipError_t e = h->x16.ensure(need);
        if (e == hipSuccess) e = h->s16.ensure(need_s);
        if (e == hipSuccess) e = h->rowerr16.ensure(sizeof(float));
        if (e != hipSuccess) {  // not an error of the call: the index works without it
            (void)hipGetLastError();
            h->x16.release();
            h->s16.release();
            h->l2_failed = true;
            return MLVDB_OK;
        }
        h->l2_rows = 0;
    }
    if (h->l2_rows == 0) {
        HIP_TRY(h, hipMemsetAsync(h->x16.p, 0, need, s));  // (the columns dim..ld16 of every row stay zero)
        HIP_TRY(h, hipMemsetAsync(h->rowerr16.p, 0, sizeof(float), s));
    }
    if (h->l2_rows < h->total) {
        HIP_TRY(h, launch_shadow16_rows(h->X, h->x16.p, h->s16.as<float>(), h->rowerr16.as<float>(), h->l2_rows, h->total, h->ld,
                                        ld16, s));
        h->l2_rows = h->total;
    }
    m->X16 = h->x16.as<_Float16>();
    m->s16 = h->s16.as<float>();
    m->row_err16 = h->rowerr16.as<float>();
    m->ld16 = ld16;
    return MLVDB_OK;
}

int run_paged_exact(mlvdb_index* h, hipStream_t s, const float* Qpad, const double* qaux, int64_t nq_space,
                    const int32_t* qsel, int32_t nsel, int32_t k, int64_t* out_labels, float* out_dist,
                    int32_t* out_counts, double* out_d64);

// ---- top_k in (64, 1024]: one pass of <= 256 queries on the filter path (round 4; DESIGN "big-k passes").
// Structure: dense seeding pass over the first <= 65,280 rows (every bound into the query's 65,536-slot list) -> [select
// the best-bounded entries -> mid (fp16) bounds for those not refined yet -> threshold = k-th largest mid LOWER bound, prune]
// -> scan rounds of geometrically growing size, each followed by the same three kernels -> mid bounds for whatever is left
// unrefined -> exact fp64 rescoring of the survivors (~1.02 k rows per query) -> ranking by (distance, label).
// Round sizes: a launch may append ~BIGK_BUDGET entries per 256 queries x k before its waves' append buffers (2,048 entries
// each) run over; with n rows seen the k-th best is at quantile k / n, so the next m rows yield ~nq m (k / n) band entries:
// m = n BUDGET / (nq k).  A query whose list or wave buffer overflows all the same is served by the paged exact scan.
// *handled = false: the index has no mid shadow (the caller takes the paged exact scan).
int run_bigk_pass(mlvdb_index* h, hipStream_t s, const float* queries_raw, float* Qpad, double* qaux, int32_t q0, int32_t nq,
                  int32_t k, int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_d64, bool* handled) {
    *handled = false;
    MidArgs m{};
    int rc = attach_mid(h, s, &m);
    if (rc) return rc;
    if (!m.X16) return MLVDB_OK;
    FilterArgs fa{};
    rc = setup_filter_ws(h, fa, Qpad + (size_t)q0 * h->ld, qaux + q0, h->qerr.as<float>() + q0, nq);
    if (rc) return rc;
    HIP_TRY(h, h->cand_range.ensure((size_t)kFilterQueries * kRangeCandCap * sizeof(CandEntry)));
    HIP_TRY(h, h->rhits.ensure((size_t)kFilterQueries * kCandCap * sizeof(RangeHit)));
    HIP_TRY(h, h->rhit_cnt.ensure(kFilterQueries * sizeof(uint32_t)));
    HIP_TRY(h, h->picks.ensure((size_t)kFilterQueries * kPicksCap * sizeof(uint32_t)));
    HIP_TRY(h, h->npicks.ensure(kFilterQueries * sizeof(uint32_t)));
    fa.cand = h->cand_range.as<CandEntry>();
    fa.cand_cap = kRangeCandCap;
    fa.rhits = h->rhits.as<RangeHit>();
    fa.rhit_cnt = h->rhit_cnt.as<uint32_t>();
    rc = attach_i8(h, s, fa);
    if (rc) return rc;
    rc = prep_pass(h, s, fa, queries_raw + (size_t)q0 * h->dim, Qpad + (size_t)q0 * h->ld, qaux + q0, h->qerr.as<float>() + q0);
    if (rc) return rc;
    h->stats.bound_dtype = fa.X8 ? 2 : 1;
    uint32_t* picks = h->picks.as<uint32_t>();
    uint32_t* npicks = h->npicks.as<uint32_t>();
    MidArgs mp = m;  // refine the picks / (m) everything still unrefined
    mp.picks = picks;
    mp.npicks = npicks;
    mp.picks_cap = kPicksCap;
    const int32_t want = k + std::max(32, k / 4);
    auto refine = [&](int32_t forced_cnt) -> int {
        HIP_TRY(h, launch_bigk_select(fa, want, forced_cnt, picks, npicks, kPicksCap, s));
        HIP_TRY(h, launch_mid_score(fa, mp, s));
        HIP_TRY(h, launch_bigk_thr_prune(fa, m, k, forced_cnt, s));
        return MLVDB_OK;
    };
    // seed: dense over the first rows (a multiple of 128; a whole number of 768-row units when rounds follow)
    // How many: the first scan round appends ~32 rows x nq x (k / n_seed) x 8 entries per wave and tile, which must stay
    // well inside a wave's buffer: n_seed >= ~66 k nq / 256 (k = 100: 6,912 rows, k = 1000: the 65,280 a list takes)
    const int64_t want_seed = std::min<int64_t>(kBigSeedRows, std::max<int64_t>(kSeedRows, ((int64_t)66 * k * std::max(nq, 16) / 256 + kFilterTile - 1) / kFilterTile * kFilterTile));
    const int64_t n_seed = h->total > want_seed ? want_seed : (h->total + 127) / 128 * 128;
    HIP_TRY(h, launch_filter_dense_scan(fa, n_seed, s));
    rc = refine((int32_t)n_seed);
    if (rc) return rc;
    // growth of the rows seen per round: bounded by the waves' append buffers (BIGK_BUDGET entries per launch and 256 queries:
    // 256 workgroups x 8 waves x ~1,000 slots, less the band factor ~8) and by the query's own list (half of its 65,536 slots
    // for one round's band: m (k / n) 8 <= 32,768)
    const double growth = std::min(std::min(24.0, 1.0 + 4096.0 / k),
                                   std::max(1.5, 1.0 + (double)std::max(1, h->tn.bigk_budget) / ((double)std::max(nq, 16) * k)));
    int64_t b = std::min<int64_t>(n_seed, h->total);
    while (b < h->total) {
        int64_t e = (int64_t)((double)b * growth) / kFilterTile * kFilterTile;
        if (e <= b) e = b + kFilterTile;
        if (e > h->total || (double)e * 1.25 > (double)h->total) e = h->total;  // (no sliver of a last round)
        rc = scan_event(h, s, true);
        if (rc) return rc;
        ScanInfo info;
        HIP_TRY(h, launch_filter_scan(fa, b, e, s, &info));
        rc = scan_event(h, s, false);
        if (rc) return rc;
        h->stats.scan_launches += 1;
        h->stats.rows_scanned += e - b;
        rc = refine(-1);
        if (rc) return rc;
        b = e;
    }
    // whatever survived without a mid bound (the band of the last rounds) gets one; the threshold then rests on all of them
    HIP_TRY(h, launch_mid_score(fa, m, s));
    HIP_TRY(h, launch_bigk_thr_prune(fa, m, k, -1, s));
    unsigned long long* stats = h->counters.as<unsigned long long>();
    HIP_TRY(h, launch_knn_rescore_rank(fa, k, q0, out_labels, out_dist, out_counts, out_d64, stats, s));
    // overflowed queries (a list or a wave's append buffer ran over, or more than 8,192 rows tie into the top k): the
    // paged exact scan serves them; the decision needs the host here (one synchronisation per pass, top_k > 64 only)
    int32_t n_flagged = 0;
    rc = collect_overflow(h, s, fa, &n_flagged);
    if (rc) return rc;
    if (n_flagged) {
        h->host_fallbacks += n_flagged;
        rc = run_paged_exact(h, s, fa.Qpad, fa.qaux, nq, h->qsel.as<int32_t>(), n_flagged, k, out_labels + (size_t)q0 * k,
                             out_dist + (size_t)q0 * k, out_counts + q0, out_d64 ? out_d64 + (size_t)q0 * k : nullptr);
        if (rc) return rc;
    }
    *handled = true;
    return MLVDB_OK;
}

// ---- top_k above MLVDB_MAX_TOPK: rank-ordered pages of the exact scan.  Page p returns the next
// entries strictly after the cursor (fp64 distance, label) of page p-1, so pages never overlap.
// Query ids: blockIdx / thread index i addresses qsel[i] when a selection is given.
__global__ void page_init_kernel(double* cur_d, int32_t* cur_l, int32_t* out_counts, const int32_t* qsel, int64_t nsel) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsel) {
        const int64_t q = qsel ? qsel[i] : i;
        cur_d[q] = -__builtin_inf();
        cur_l[q] = -1;
        out_counts[q] = 0;
    }
}

__global__ void page_commit_kernel(const int64_t* page_lab, const float* page_dist, const int32_t* page_cnt,
                                   const double* page_d64, const int32_t* qsel, int32_t kp, int32_t offset, int32_t k,
                                   int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_d64,
                                   double* cur_d, int32_t* cur_l) {
    const int q = qsel ? qsel[blockIdx.x] : blockIdx.x;
    const int i = threadIdx.x;  // kp <= 64 threads
    const int n = page_cnt[q];
    if (i < kp) {
        const int64_t o = (int64_t)q * k + offset + i;
        const bool ok = i < n;
        out_labels[o] = ok ? page_lab[(int64_t)q * kp + i] : -1;
        out_dist[o] = ok ? page_dist[(int64_t)q * kp + i] : __builtin_inff();
        if (out_d64) out_d64[o] = ok ? page_d64[(int64_t)q * kp + i] : __builtin_inf();
    }
    if (i == 0) {
        out_counts[q] += n;
        if (n > 0) {
            cur_d[q] = page_d64[(int64_t)q * kp + n - 1];
            cur_l[q] = (int32_t)page_lab[(int64_t)q * kp + n - 1];
        }
        if (n < kp) {  // exhausted: later pages must find nothing
            cur_d[q] = __builtin_inf();
            cur_l[q] = 0x7fffffff;
        }
    }
}

// nq_space = size of the query index space of Qpad/qaux and of the outputs; (qsel, nsel) = the queries to serve
int run_paged_exact(mlvdb_index* h, hipStream_t s, const float* Qpad, const double* qaux, int64_t nq_space,
                    const int32_t* qsel, int32_t nsel, int32_t k, int64_t* out_labels, float* out_dist,
                    int32_t* out_counts, double* out_d64) {
    const int32_t page = MLVDB_MAX_TOPK;
    HIP_TRY(h, h->page_lab.ensure((size_t)nq_space * page * sizeof(int64_t)));
    HIP_TRY(h, h->page_dist.ensure((size_t)nq_space * page * sizeof(float)));
    HIP_TRY(h, h->page_cnt.ensure((size_t)nq_space * sizeof(int32_t)));
    HIP_TRY(h, h->page_d64.ensure((size_t)nq_space * page * sizeof(double)));
    HIP_TRY(h, h->cur_d.ensure((size_t)nq_space * sizeof(double)));
    HIP_TRY(h, h->cur_l.ensure((size_t)nq_space * sizeof(int32_t)));
    page_init_kernel<<<(unsigned)((nsel + 255) / 256), 256, 0, s>>>(h->cur_d.as<double>(), h->cur_l.as<int32_t>(),
                                                                    out_counts, qsel, nsel);
    HIP_TRY(h, hipGetLastError());
    for (int32_t offset = 0; offset < k; offset += page) {
        const int32_t kp = std::min(page, k - offset);
        int rc = run_exact(h, s, Qpad, qaux, nsel, qsel, 0, h->total, kp, h->page_lab.as<int64_t>(),
                           h->page_dist.as<float>(), h->page_cnt.as<int32_t>(), h->page_d64.as<double>(), true, nullptr,
                           h->cur_d.as<double>(), h->cur_l.as<int32_t>());
        if (rc) return rc;
        page_commit_kernel<<<(unsigned)nsel, 64, 0, s>>>(h->page_lab.as<int64_t>(), h->page_dist.as<float>(),
                                                         h->page_cnt.as<int32_t>(), h->page_d64.as<double>(), qsel, kp,
                                                         offset, k, out_labels, out_dist, out_counts, out_d64,
                                                         h->cur_d.as<double>(), h->cur_l.as<int32_t>());
        HIP_TRY(h, hipGetLastError());
    }
    return MLVDB_OK;
}

// Range fallback for queries whose candidate list overflowed: copy the nearest hits found by the paged
// exact kNN into the range outputs and publish the exact counts.
__global__ void range_fallback_commit_kernel(const int32_t* qsel, const uint32_t* exact_cnt, const int64_t* knn_lab,
                                             const float* knn_dist, int32_t kmax, int32_t q0, int64_t cap_eff,
                                             int64_t* out_labels, float* out_dist, int64_t* out_counts) {
    const int q = qsel[blockIdx.x];
    const int64_t n = exact_cnt[q];
    const int64_t emit = n < cap_eff ? n : cap_eff;
    for (int64_t i = threadIdx.x; i < emit; i += blockDim.x) {
        out_labels[(int64_t)(q0 + q) * cap_eff + i] = knn_lab[(int64_t)q * kmax + i];
        out_dist[(int64_t)(q0 + q) * cap_eff + i] = knn_dist[(int64_t)q * kmax + i];
    }
    if (threadIdx.x == 0) out_counts[q0 + q] = n;
}

// valid prefixes of the dense per-query result rows -> one packed array (offsets from the host)
__global__ void range_pack_kernel(const int64_t* lab, const float* dist, const int64_t* offsets, int64_t cap_eff, int64_t* plab,
                                  float* pdist) {
    const int64_t q = blockIdx.x;
    const int64_t o = offsets[q], n = offsets[q + 1] - o;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        plab[o + i] = lab[q * cap_eff + i];
        pdist[o + i] = dist[q * cap_eff + i];
    }
}

__global__ void range_reset_kernel(uint32_t* cnt, const int32_t* qsel) { cnt[qsel[threadIdx.x]] = 0; }
__global__ void range_resolve_kernel(uint32_t* overflow, const uint32_t* cnt, const int32_t* qsel, uint32_t cap) {
    const int q = qsel[threadIdx.x];
    overflow[q] = cnt[q] > cap ? 1u : 0u;
}

// (l2: the int8 bodies are the l2c ones -- folded test, per-row integer offsets read through the row pairs' descriptor, which needs
// pairs + offsets, 12 bytes per row, below 4 GB; SCAN_L2C=0 / SCAN_L2E=0 take l2 off the int8 shadow: bf16 / fp32 / exact paths)
bool l2_int8_ok(const mlvdb_index* h) {
    return h->space != kSpaceL2 || (h->tn.scan_l2e && h->tn.scan_l2c && (uint64_t)h->capacity * 12ull < 0xfff00000ull);
}
// One scale per row (l2 / ip: per group of 8 rows): a row with an outlier component quantises badly.  Cosine bounds carry every
// row's own error (any index-wide maximum up to 0.5 will do).  ip bounds use the index-wide maximum: beyond 0.03 (typical data:
// 0.008-0.015) they would admit everything (I8_ERR_IP, thousandths).  l2 bounds carry per-group errors (round 4), so a FEW odd
// rows cost only their own groups: the index stays on the int8 shadow while at most 64 + 0.05 % of its rows sit in groups above
// 0.03 (shadow8_rows_kernel counts them) and the worst stays under I8_ERR_L2 thousandths (1100: any finite row -- a group
// quantised to zeros has error 1 and Cauchy-Schwarz's bound; 30: round 3's index-wide rule); such a pass seeds its thresholds
// exactly (run_filter_pass) because the dense int8 seeding pass does use the index-wide maximum.
// profiles/r04/outlier_row_ab_l2_ip_4m_before.txt: 5 rows with a 40-sigma component among 4M x 768 made every l2 / ip wave
// 3.3 x slower (fp32 rows converted in registers), 54 x on d = 300 (exact scan); outlier_row_ab_l2_4m.txt: l2 now 1.1-1.2 x.
bool i8_bounds_usable(const mlvdb_index* h) {
    if (!l2_int8_ok(h)) return false;
    if (h->space == kSpaceCosine) return h->i8_err <= 0.5f;
    if (h->i8_err <= 0.03f) return true;
    if (h->space == kSpaceIp) return h->i8_err <= 0.001f * (float)h->tn.i8_err_ip;
    return h->i8_err <= 0.001f * (float)h->tn.i8_err_l2 && (int64_t)h->i8_odd_groups * 8 <= 64 + h->total / 2000;
}

// Is a filter body available for this index right now?  ld % 64 == 0: always (bf16 shadow, int8 shadow, or the fp32 rows
// converted in registers).  Any other ld has only the int8 body: the (zero-padded) int8 shadow is brought up to date here
// and its bounds must be usable (attach_i8's criterion: l2 / ip rows that quantise too badly go to the exact scan).
bool use_filter(const mlvdb_index* h, int64_t nq, bool ready);

int filter_ready(mlvdb_index* h, hipStream_t s, int64_t nq, bool* ready) {
    *ready = false;
    if (h->strategy == MLVDB_STRATEGY_EXACT || h->total == 0) return MLVDB_OK;
    if (!use_filter(h, nq, true)) return MLVDB_OK;  // AUTO would take the exact scan anyway: no shadow is built for this call
    if (filter_supported(h->ld)) {
        *ready = true;
        return MLVDB_OK;
    }
    if (!i8_eligible(h)) return MLVDB_OK;
    if (h->mask_active) {
        *ready = h->mask_pairs_ready && i8_bounds_usable(h);
        return MLVDB_OK;
    }
    int rc = update_i8_shadow(h, s);
    if (rc) return rc;
    *ready = i8_bounds_usable(h);
    return MLVDB_OK;
}

bool use_filter(const mlvdb_index* h, int64_t nq, bool ready) {
    if (h->strategy == MLVDB_STRATEGY_EXACT || !ready) return false;
    if (h->strategy == MLVDB_STRATEGY_FILTER) return true;
    const bool shadowed = h->Xb != nullptr || h->i8_only;
    // dim < 64: the padded int8 shadow (256 B per row) is wider than the fp32 rows (4 ld B), and the exact scan of such short
    // rows is bound by its per-row work, not by HBM (4M rows, d = 16 / 32 / 48: 0.16-0.21 ms for one query, 0.30-0.36 for 4,
    // 0.45-0.58 ms per 8-query pass; the int8 chain: 0.28 ms for 5-64 queries, 0.37 ms for 256; 500k rows x 32: 0.12-0.16
    // against 0.17 / 0.20 / 0.35 / 1.67 ms for 5 / 8 / 16 / 256 queries -- profiles/r04/dim_ab_small_dims_{4m,500k}.txt).
    // Up to 4 queries the exact scan wins at any size; beyond, the filter wins once the exact passes (one per 8 queries)
    // cover some 400k rows between them.
    if (h->i8_only && h->ld8 >= 4 * h->ld) return nq > 4 && h->total >= 32768 && (nq + 7) / 8 * h->total >= 400000;
    if (nq >= 12 || (nq >= 8 && shadowed)) return h->total >= 32768;
    // small batches: the narrow filter kernel streams the bf16 shadow, half the bytes of the exact fp32 scan; below
    // ~125k rows of 768 columns the exact scan's two launches win (profiles/r01/small_batch_ab_crossover.txt)
    return shadowed && h->total * (int64_t)h->ld >= (int64_t)96 << 20;
}

int check_handle(mlvdb_index* h) {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(h, MLVDB_ERR_HIP, "hipSetDevice", e);
    return MLVDB_OK;
}

// The header promises "never throws, returns an int status": every extern "C" body runs inside this guard, so that a
// host-side allocation failure (std::vector / std::string growth) or any other C++ exception becomes a status code
// instead of unwinding through the caller's C / ctypes frames (std::terminate -> SIGABRT).
template <class F>
int guarded(mlvdb_index* h, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { return fail(h, MLVDB_ERR_OUT_OF_MEMORY, "host allocation failed (std::bad_alloc)"); } catch (...) { return MLVDB_ERR_OUT_OF_MEMORY; }
    } catch (const std::exception& e) {
        try { return fail(h, MLVDB_ERR_INTERNAL, e.what()); } catch (...) { return MLVDB_ERR_INTERNAL; }
    } catch (...) {
        try { return fail(h, MLVDB_ERR_INTERNAL, "unknown C++ exception"); } catch (...) { return MLVDB_ERR_INTERNAL; }
    }
}

}  // namespace

// ============================================================================== C ABI
extern "C" {

int mlvdb_abi_version(void) { return MLVDB_ABI_VERSION; }

const char* mlvdb_last_global_error(void) { return g_error.c_str(); }

const char* mlvdb_last_error(const mlvdb_index* h) { return h ? h->err.c_str() : g_error.c_str(); }

int mlvdb_device_count(int* count) {
    return guarded(nullptr, [&]() -> int {
    if (!count) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return fail(nullptr, MLVDB_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", e);
    }
    *count = n;
    return MLVDB_OK;
    });
}

int64_t mlvdb_layout_offset(int64_t row, int32_t col, int32_t ld) { return layout_offset(row, col, ld); }
int32_t mlvdb_layout_ld(int32_t dim) { return layout_ld(dim); }

int mlvdb_index_create(int device, int32_t dim, int32_t space, int64_t capacity_hint, mlvdb_index** out) {
    return guarded(nullptr, [&]() -> int {
    if (!out) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (dim <= 0 || dim > 8192) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "dim must be in 1..8192");
    if (space < 0 || space > 2) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "space must be MLVDB_SPACE_L2/COSINE/IP");
    if (capacity_hint < 0) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "capacity_hint < 0");
    int n = 0;
    int rc = mlvdb_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, MLVDB_ERR_HIP, "hipSetDevice", e);
    mlvdb_index* h = new (std::nothrow) mlvdb_index();
    if (!h) return fail(nullptr, MLVDB_ERR_OUT_OF_MEMORY, "host allocation failed");
    h->device = device;
    h->dim = dim;
    h->ld = layout_ld(dim);
    h->space = space;
    h->tn = tuning_from_env();  // the one moment the environment is consulted
    {
        // Which shadow the index keeps (decided here, once).  The int8 shadow has its own width ld8 = round_up(ld, 256), zero
        // padded -- zero columns change neither a dot product nor a norm -- so every dim gets the int8 body (round 4; dim < 64
        // too: there the shadow is wider than the fp32 rows, it is built by the first batch large enough to want it, see
        // use_filter; I8_PAD=0: only dim % 256 == 0, round 3).  It is the only shadow wherever it streams fewer bytes per row than the
        // bf16 one would (ld8 < 2 ld) or no bf16 body exists (ld % 64 != 0): seeding pass, small batches, scans, range and
        // row-mask searches all run on it.  ld = 64 / 128 keep the bf16 shadow (same bytes, tighter bounds).
        // SHADOW_BF16=1 (MLVDB_SHADOW=bf16) keeps the bf16 shadow as well (the bf16 bodies for A/B, I8=0).
        const Tuning& tn = h->tn;
        h->ld8 = 0;
        if (!tn.no_shadow) {
            const int32_t cand = (h->ld + 255) / 256 * 256;
            if (cand == h->ld || (tn.i8_pad && (cand < 2 * h->ld || !filter_supported(h->ld)))) h->ld8 = cand;
        }
        h->i8_only = h->ld8 > 0 && !tn.shadow_bf16;
        h->shadow = filter_supported(h->ld) && !tn.no_shadow && !h->i8_only;
    }
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(nullptr, MLVDB_ERR_HIP, "hipStreamCreate", e);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = h->counters.ensure(64);
    if (e != hipSuccess) {
        mlvdb_index_destroy(h);
        return fail(nullptr, MLVDB_ERR_HIP, "hipMalloc(counters)", e);
    }
    if (capacity_hint > 0) {
        rc = reserve_rows(h, capacity_hint);
        if (rc) {
            g_error = h->err;
            mlvdb_index_destroy(h);
            return rc;
        }
    }
    *out = h;
    return MLVDB_OK;
    });
}

int mlvdb_index_destroy(mlvdb_index* h) {
    return guarded(h, [&]() -> int {
    if (!h) return MLVDB_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipDeviceSynchronize();
    if (h->X) (void)hipFree(h->X);
    if (h->rn) (void)hipFree(h->rn);
    if (h->Xb) (void)hipFree(h->Xb);
    for (DevBuf* b : {&h->stage, &h->qpad, &h->qaux, &h->partial, &h->qsel, &h->seed_lab, &h->seed_dist, &h->seed_cnt,
                      &h->seed_d64, &h->qimg, &h->fmisc, &h->cand, &h->rescr, &h->wgbuf, &h->wgcnt, &h->row_mask, &h->rn_masked, &h->qerr, &h->rowerr, &h->io_q, &h->io_lab, &h->io_dist, &h->io_cnt, &h->io_d64, &h->gather_out, &h->gather_lab, &h->cand_range, &h->rhits, &h->rhit_cnt, &h->rp8_masked, &h->x16, &h->s16, &h->rowerr16, &h->picks, &h->npicks, &h->x8, &h->rp8, &h->rowerr8, &h->qimg8, &h->sq8,
                      &h->counters, &h->labels_in, &h->page_lab, &h->page_dist, &h->page_cnt, &h->page_d64, &h->cur_d,
                      &h->cur_l})
        b->release();
    if (h->host_flags) (void)hipHostFree(h->host_flags);
    h->pin_in.release();
    h->pin_out.release();
    h->io_out.release();
    for (auto& p : h->scan_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto& ev : h->total_events)
        if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    delete h;
    return MLVDB_OK;
    });
}

int mlvdb_index_append_device(mlvdb_index* h, const float* rows_device, int64_t n, int64_t* first_label) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !rows_device)) return fail(h, MLVDB_ERR_INVALID_ARG, "bad rows / n");
    if (h->total + n > 0x7fffff00ll) return fail(h, MLVDB_ERR_UNSUPPORTED, "more than 2^31 rows per index");
    if (first_label) *first_label = h->total;
    if (n == 0) return MLVDB_OK;
    rc = reserve_rows(h, h->total + n);
    if (rc) return rc;
    HIP_TRY(h, launch_scatter_rows(rows_device, h->X, h->total, n, h->dim, h->ld, h->stream));
    HIP_TRY(h, launch_row_norms(h->X, h->rn, h->total, n, h->ld, h->rowerr.as<unsigned int>(), h->stream));
    if (h->Xb) HIP_TRY(h, launch_shadow_rows(h->X, h->Xb, h->total, n, h->ld, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->total += n;
    return MLVDB_OK;
    });
}

int mlvdb_index_append(mlvdb_index* h, const float* rows, int64_t n, int64_t* first_label) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !rows)) return fail(h, MLVDB_ERR_INVALID_ARG, "bad rows / n");
    if (h->total + n > 0x7fffff00ll) return fail(h, MLVDB_ERR_UNSUPPORTED, "more than 2^31 rows per index");
    if (first_label) *first_label = h->total;
    if (n == 0) return MLVDB_OK;
    rc = reserve_rows(h, h->total + n);
    if (rc) return rc;
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(256u << 20) / ((int64_t)h->dim * 4));
    for (int64_t done = 0; done < n; done += chunk_rows) {
        const int64_t m = std::min(chunk_rows, n - done);
        HIP_TRY(h, h->stage.ensure((size_t)m * h->dim * sizeof(float)));
        HIP_TRY(h, hipMemcpyAsync(h->stage.p, rows + (size_t)done * h->dim, (size_t)m * h->dim * sizeof(float),
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, launch_scatter_rows(h->stage.as<float>(), h->X, h->total + done, m, h->dim, h->ld, h->stream));
        HIP_TRY(h, launch_row_norms(h->X, h->rn, h->total + done, m, h->ld, h->rowerr.as<unsigned int>(), h->stream));
        if (h->Xb) HIP_TRY(h, launch_shadow_rows(h->X, h->Xb, h->total + done, m, h->ld, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->total += n;
    return MLVDB_OK;
    });
}

int mlvdb_index_tombstone(mlvdb_index* h, const int64_t* labels, int64_t n, int64_t* newly_deleted) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !labels)) return fail(h, MLVDB_ERR_INVALID_ARG, "bad labels / n");
    if (newly_deleted) *newly_deleted = 0;
    if (n == 0 || h->total == 0) return MLVDB_OK;
    HIP_TRY(h, h->labels_in.ensure((size_t)n * sizeof(int64_t)));
    HIP_TRY(h, hipMemcpyAsync(h->labels_in.p, labels, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, sizeof(unsigned long long), h->stream));
    HIP_TRY(h, launch_tombstone(h->rn, h->labels_in.as<int64_t>(), n, h->total, h->counters.as<unsigned long long>(),
                                h->stream));
    if (h->rp8.p && h->i8_rows > 0) {  // the int8 shadow's row constants carry the tombstones too (before the sync below:
                                       // a search on another stream may follow this call at once)
        tombstone_rp8_kernel<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(h->labels_in.as<int64_t>(), n, h->rp8.as<float>(), h->i8_rows, h->space == kSpaceL2 ? 1 : 0);
        HIP_TRY(h, hipGetLastError());
        if (int rc2 = forget_l2_offsets(h, 0, h->stream)) return rc2;  // (a dead row changes its lane group's P0)
    }
    unsigned long long changed = 0;
    HIP_TRY(h, hipMemcpyAsync(&changed, h->counters.p, sizeof changed, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->deleted += (int64_t)changed;
    if (newly_deleted) *newly_deleted = (int64_t)changed;
    return MLVDB_OK;
    });
}

int mlvdb_index_compact(mlvdb_index* h, int64_t* old_labels, int64_t capacity, int64_t* live_out) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    const int64_t want = h->total - h->deleted;
    if (capacity < want || (want > 0 && !old_labels)) return fail(h, MLVDB_ERR_INVALID_ARG, "old_labels too small");
    if (live_out) *live_out = want;
    if (h->deleted == 0) {  // nothing to drop: identity
        for (int64_t i = 0; i < want; ++i) old_labels[i] = i;
        return MLVDB_OK;
    }
    hipStream_t s = h->stream;
    const int64_t nblocks = (h->total + 1023) / 1024;
    HIP_TRY(h, h->partial.ensure((size_t)nblocks * sizeof(uint32_t) + 64));
    HIP_TRY(h, h->labels_in.ensure((size_t)std::max<int64_t>(want, 1) * sizeof(int32_t)));
    uint32_t* scratch = h->partial.as<uint32_t>();
    int32_t* old_of_new = h->labels_in.as<int32_t>();
    unsigned long long* live_d = h->counters.as<unsigned long long>();
    HIP_TRY(h, launch_compact_map(h->rn, h->total, scratch, live_d, old_of_new, s));
    unsigned long long live = 0;
    HIP_TRY(h, hipMemcpyAsync(&live, live_d, sizeof live, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if ((int64_t)live != want) return fail(h, MLVDB_ERR_INTERNAL, "live-row count disagrees with the tombstone accounting");
    // new buffers sized for the live rows (zero / NaN filled), gather, swap
    const int64_t cap = round_up_rows(std::max<int64_t>(want, 1));
    float* nX = nullptr;
    float* nrn = nullptr;
    void* nXb = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nX), (size_t)cap * h->ld * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&nrn), (size_t)cap * sizeof(float));
    if (e == hipSuccess && h->Xb) e = hipMalloc(&nXb, (size_t)cap * h->ld * 2);
    if (e != hipSuccess) {
        if (nX) (void)hipFree(nX);
        if (nrn) (void)hipFree(nrn);
        return fail(h, MLVDB_ERR_OUT_OF_MEMORY, "hipMalloc(compaction buffers)", e);
    }
    HIP_TRY(h, hipMemsetAsync(nX, 0, (size_t)cap * h->ld * sizeof(float), s));
    HIP_TRY(h, hipMemsetAsync(nrn, 0xFF, (size_t)cap * sizeof(float), s));  // NaN = not a row
    if (nXb) HIP_TRY(h, hipMemsetAsync(nXb, 0, (size_t)cap * h->ld * 2, s));
    HIP_TRY(h, launch_compact_rows(h->X, nX, h->Xb, nXb, h->rn, nrn, old_of_new, want, h->ld, s));
    std::vector<int32_t> host_map((size_t)want);
    if (want > 0)
        HIP_TRY(h, hipMemcpyAsync(host_map.data(), old_of_new, (size_t)want * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int64_t i = 0; i < want; ++i) old_labels[i] = host_map[(size_t)i];
    (void)hipFree(h->X);
    (void)hipFree(h->rn);
    if (h->Xb) (void)hipFree(h->Xb);
    h->X = nX;
    h->rn = nrn;
    h->Xb = nXb;
    h->capacity = cap;
    h->total = want;
    h->deleted = 0;
    h->i8_rows = 0;
    h->l2_rows = 0;
    return MLVDB_OK;
    });
}

int mlvdb_index_counts(const mlvdb_index* h, int64_t* total, int64_t* deleted) {
    return guarded(const_cast<mlvdb_index*>(h), [&]() -> int {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    if (total) *total = h->total;
    if (deleted) *deleted = h->deleted;
    return MLVDB_OK;
    });
}

int mlvdb_index_reset(mlvdb_index* h, int32_t space) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (space > 2) return fail(h, MLVDB_ERR_INVALID_ARG, "space must be < 0 (keep) or a MLVDB_SPACE_* value");
    if (h->capacity > 0) {
        HIP_TRY(h, hipMemsetAsync(h->X, 0, (size_t)h->capacity * h->ld * sizeof(float), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->rn, 0xFF, (size_t)h->capacity * sizeof(float), h->stream));
        if (h->Xb) HIP_TRY(h, hipMemsetAsync(h->Xb, 0, (size_t)h->capacity * h->ld * 2, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (h->rowerr.p) HIP_TRY(h, hipMemsetAsync(h->rowerr.p, 0, sizeof(unsigned int), h->stream));
    h->total = 0;
    h->deleted = 0;
    h->i8_rows = 0;
    h->l2_rows = 0;
    if (space >= 0) h->space = space;
    return MLVDB_OK;
    });
}

int mlvdb_index_get_rows(mlvdb_index* h, int64_t first, int64_t n, float* out_rows) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (first < 0 || n < 0 || first + n > h->total || (n > 0 && !out_rows))
        return fail(h, MLVDB_ERR_INVALID_ARG, "row range out of bounds");
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(256u << 20) / ((int64_t)h->dim * 4));
    for (int64_t done = 0; done < n; done += chunk_rows) {
        const int64_t m = std::min(chunk_rows, n - done);
        HIP_TRY(h, h->stage.ensure((size_t)m * h->dim * sizeof(float)));
        HIP_TRY(h, launch_gather_rows(h->X, h->stage.as<float>(), first + done, m, h->dim, h->ld, h->stream));
        HIP_TRY(h, hipMemcpyAsync(out_rows + (size_t)done * h->dim, h->stage.p, (size_t)m * h->dim * sizeof(float),
                                  hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return MLVDB_OK;
    });
}

int mlvdb_index_get_rows_at(mlvdb_index* h, const int64_t* labels, int64_t n, float* out_rows) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!labels || !out_rows))) return fail(h, MLVDB_ERR_INVALID_ARG, "bad labels / n / out_rows");
    for (int64_t i = 0; i < n; ++i)
        if (labels[i] < 0 || labels[i] >= h->total) return fail(h, MLVDB_ERR_INVALID_ARG, "label out of range");
    // Own stream and buffers: rows are immutable once appended, so this may overlap a search that another host thread
    // has in flight on h->stream (QueryProcessor.find_similar_stream enriches wave i while wave i+1 is scanned).
    hipStream_t s = h->aux_stream;
    std::lock_guard<std::mutex> aux_lock(h->aux_mutex);
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)h->dim * 4));
    for (int64_t done = 0; done < n; done += chunk_rows) {
        const int64_t m = std::min(chunk_rows, n - done);
        HIP_TRY(h, h->gather_out.ensure((size_t)m * h->dim * sizeof(float)));
        HIP_TRY(h, h->gather_lab.ensure((size_t)m * sizeof(int64_t)));
        HIP_TRY(h, hipMemcpyAsync(h->gather_lab.p, labels + done, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIP_TRY(h, launch_gather_rows_at(h->X, h->gather_out.as<float>(), h->gather_lab.as<int64_t>(), m, h->dim, h->ld, s));
        HIP_TRY(h, hipMemcpyAsync(out_rows + (size_t)done * h->dim, h->gather_out.p, (size_t)m * h->dim * sizeof(float),
                                  hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return MLVDB_OK;
    });
}

static int search_device_impl(mlvdb_index* h, const float* queries_device, int64_t nq, int32_t k,
                              int64_t* out_labels_device, float* out_dist_device, int32_t* out_counts_device,
                              double* out_dist64_device, void* stream, bool defer_fallback);

int mlvdb_search_batch_device(mlvdb_index* h, const float* queries_device, int64_t nq, int32_t k,
                              int64_t* out_labels_device, float* out_dist_device, int32_t* out_counts_device,
                              double* out_dist64_device, void* stream) {
    return guarded(h, [&]() -> int {
    return search_device_impl(h, queries_device, nq, k, out_labels_device, out_dist_device, out_counts_device,
                              out_dist64_device, stream, false);
    });
}

static int search_device_impl(mlvdb_index* h, const float* queries_device, int64_t nq, int32_t k,
                              int64_t* out_labels_device, float* out_dist_device, int32_t* out_counts_device,
                              double* out_dist64_device, void* stream, bool defer_fallback) {
    int rc = check_handle(h);
    if (rc) return rc;
    h->deferred = false;
    if (nq < 0 || nq > (1 << 24)) return fail(h, MLVDB_ERR_INVALID_ARG, "nq out of range");
    if (k < 1) return fail(h, MLVDB_ERR_INVALID_ARG, "k must be >= 1");
    if (k > MLVDB_MAX_TOPK_PAGED) return fail(h, MLVDB_ERR_UNSUPPORTED, "k above MLVDB_MAX_TOPK_PAGED");
    if (nq == 0) return MLVDB_OK;
    if (!queries_device || !out_labels_device || !out_dist_device || !out_counts_device)
        return fail(h, MLVDB_ERR_INVALID_ARG, "null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = the caller's default (null) stream
    rc = begin_call(h, s);
    if (rc) return rc;
    if (h->total == 0 || h->total == h->deleted) {
        fill_empty_kernel<<<(unsigned)((nq * k + 255) / 256), 256, 0, s>>>(out_labels_device, out_dist_device,
                                                                            out_counts_device, out_dist64_device, nq, k);
        HIP_TRY(h, hipGetLastError());
        return end_call(h, s);
    }
    HIP_TRY(h, h->qpad.ensure((size_t)nq * h->ld * sizeof(float)));
    HIP_TRY(h, h->qaux.ensure((size_t)nq * sizeof(double)));
    if (!h->counters_pending) HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, 32, s));
    h->counters_stream = s;
    HIP_TRY(h, h->qerr.ensure((size_t)nq * sizeof(float)));
    bool ready = false;
    rc = filter_ready(h, s, nq, &ready);
    if (rc) return rc;
    const bool filt = k <= MLVDB_MAX_TOPK && use_filter(h, nq, ready);
    if (!filt && k <= MLVDB_MAX_TOPK)  // (the filter passes prepare their own queries: one fused launch each)
        HIP_TRY(h, launch_query_prep(queries_device, (int32_t)nq, h->dim, h->ld, h->space, h->qpad.as<float>(),
                                     h->qaux.as<double>(), h->qerr.as<float>(), s));
    bool bigk_done = false;
    if (k > MLVDB_MAX_TOPK && k <= kBigKMax && h->tn.bigk && use_filter(h, nq, ready) && h->total - h->deleted > (int64_t)k) {
        // top_k 65..1024 stays on the filter path (round 4): passes of 256 queries with 65,536-slot lists, mid bounds, exact
        // rescoring of the survivors.  (The first pass tells whether the index has a mid shadow at all.)
        h->counters_pending = true;
        for (int64_t q0 = 0; q0 < nq; q0 += kFilterQueries) {
            const int32_t n = (int32_t)std::min<int64_t>(kFilterQueries, nq - q0);
            bool handled = false;
            rc = run_bigk_pass(h, s, queries_device, h->qpad.as<float>(), h->qaux.as<double>(), (int32_t)q0, n, k, out_labels_device,
                               out_dist_device, out_counts_device, out_dist64_device, &handled);
            if (rc) return rc;
            if (!handled) break;  // (only ever the first pass)
            bigk_done = true;
        }
        if (bigk_done) h->stats.strategy_used = MLVDB_STRATEGY_FILTER;
    }
    if (bigk_done) {
    } else if (k > MLVDB_MAX_TOPK) {
        HIP_TRY(h, launch_query_prep(queries_device, (int32_t)nq, h->dim, h->ld, h->space, h->qpad.as<float>(),
                                     h->qaux.as<double>(), h->qerr.as<float>(), s));
        h->stats.strategy_used = MLVDB_STRATEGY_EXACT;
        rc = run_paged_exact(h, s, h->qpad.as<float>(), h->qaux.as<double>(), nq, nullptr, (int32_t)nq, k, out_labels_device,
                             out_dist_device, out_counts_device, out_dist64_device);
        if (rc) return rc;
    } else if (filt) {
        h->stats.strategy_used = MLVDB_STRATEGY_FILTER;
        h->counters_pending = true;
        for (int64_t q0 = 0; q0 < nq; q0 += kFilterQueries) {
            const int32_t n = (int32_t)std::min<int64_t>(kFilterQueries, nq - q0);
            rc = run_filter_pass(h, s, queries_device, h->qpad.as<float>(), h->qaux.as<double>(), (int32_t)q0, n, k, out_labels_device,
                                 out_dist_device, out_counts_device, out_dist64_device,
                                 defer_fallback && nq <= kFilterQueries);
            if (rc) return rc;
        }
    } else {
        h->stats.strategy_used = MLVDB_STRATEGY_EXACT;
        rc = run_exact(h, s, h->qpad.as<float>(), h->qaux.as<double>(), (int32_t)nq, nullptr, 0, h->total, k,
                       out_labels_device, out_dist_device, out_counts_device, out_dist64_device, true);
        if (rc) return rc;
    }
    return end_call(h, s);
}

namespace {
int search_host(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, int64_t* out_labels, float* out_dist,
                int32_t* out_counts, double* out_dist64) {
    if (nq < 0 || nq > (1 << 24)) return fail(h, MLVDB_ERR_INVALID_ARG, "nq out of range");
    if (k < 1) return fail(h, MLVDB_ERR_INVALID_ARG, "k must be >= 1");
    if (k > MLVDB_MAX_TOPK_PAGED) return fail(h, MLVDB_ERR_UNSUPPORTED, "k above MLVDB_MAX_TOPK_PAGED");
    if (nq == 0) return MLVDB_OK;
    if (!queries || !out_labels || !out_dist || !out_counts) return fail(h, MLVDB_ERR_INVALID_ARG, "null buffer");
    // queries: user memory -> pinned staging -> device (one DMA); outputs: ONE device buffer [d64 | labels | dist | counts]
    // -> one DMA into pinned memory -> user arrays
    const size_t qbytes = (size_t)nq * h->dim * sizeof(float);
    const size_t b64 = out_dist64 ? (size_t)nq * k * sizeof(double) : 0, blab = (size_t)nq * k * sizeof(int64_t);
    const size_t bdist = (size_t)nq * k * sizeof(float), bcnt = (size_t)nq * sizeof(int32_t);
    const size_t obytes = b64 + blab + bdist + bcnt;
    HIP_TRY(h, h->io_q.ensure(qbytes));
    const size_t fbytes = kFilterQueries * sizeof(uint32_t), obytes_all = ((obytes + 15) & ~(size_t)15) + fbytes;
    HIP_TRY(h, h->io_out.ensure(obytes_all));
    // (Tried in round 3: outputs of small batches written by the kernels straight into host-mapped pinned memory -- no D2H
    // copy, one host wait instead of two.  No gain: 0.239 vs 0.238 ms at batch 1 on 1M x 768; removed.)
    char* dout = h->io_out.as<char>();
    double* d_d64 = out_dist64 ? reinterpret_cast<double*>(dout) : nullptr;
    int64_t* d_lab = reinterpret_cast<int64_t*>(dout + b64);
    float* d_dist = reinterpret_cast<float*>(dout + b64 + blab);
    int32_t* d_cnt = reinterpret_cast<int32_t*>(dout + b64 + blab + bdist);
    // MLVDB_PINNED_IO=0: round 2's pageable copies (A/B).  The pinned D2H copy is enqueued only AFTER the kernels have
    // finished (one more host wait, ~10 us): enqueued behind them it parks at the head of the copy engine's queue for the
    // whole scan and every copy of the index's other stream -- the hit-enrichment gather of find_similar_stream -- waits
    // with it; measured on 4M rows: protocol stream 1.92 ms per wave parked vs 1.20 unparked (engine alone 1.07;
    // profiles/r03/protocol_stream_pinned_io_modes_4m.txt).  A blocking event wait instead of hipStreamSynchronize: no change.
    // (pinned staging only while it stays small: a paged search with nq = 1024, k = 16384 would pin ~400 MB per handle -- per
    // shard under MultiDeviceEngine -- for the handle's lifetime, and every growth goes through hipHostFree / hipHostMalloc,
    // which synchronise the device; above 16 MB the pageable copies of round 2 are used and nothing stays pinned)
    constexpr size_t kPinnedMax = (size_t)16 << 20;
    const bool pinned = h->tn.pinned_io != 0 && qbytes <= kPinnedMax && obytes_all <= kPinnedMax;
    h->flags_in_out = pinned;  // the pass copies its overflow flags device-to-device behind the outputs (no parked D2H either)
    h->flags_out = reinterpret_cast<uint32_t*>(dout + ((obytes + 15) & ~(size_t)15));
    if (pinned) {
        HIP_TRY(h, h->pin_in.ensure(qbytes));
        HIP_TRY(h, h->pin_out.ensure(obytes_all));
        std::memcpy(h->pin_in.p, queries, qbytes);
        HIP_TRY(h, hipMemcpyAsync(h->io_q.p, h->pin_in.p, qbytes, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->io_q.p, queries, qbytes, hipMemcpyHostToDevice, h->stream));
    }
    int rc = search_device_impl(h, h->io_q.as<float>(), nq, k, d_lab, d_dist, d_cnt, d_d64, h->stream, true);
    if (rc) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (pinned) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));  // the copy must not park behind the kernels (see above)
            HIP_TRY(h, hipMemcpyAsync(h->pin_out.p, dout, obytes_all, hipMemcpyDeviceToHost, h->stream));
        } else {  // round 2's form: pageable copies straight into the caller's arrays
            if (out_dist64) HIP_TRY(h, hipMemcpyAsync(out_dist64, d_d64, b64, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(out_labels, d_lab, blab, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(out_dist, d_dist, bdist, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(out_counts, d_cnt, bcnt, hipMemcpyDeviceToHost, h->stream));
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->deferred) {
            // the pass left the overflow decision to us (run_filter_pass, defer_fallback): usually nothing is flagged
            h->deferred = false;
            int32_t sel[kFilterQueries];
            int32_t n = 0;
            const uint32_t* flags = pinned ? reinterpret_cast<const uint32_t*>(static_cast<const char*>(h->pin_out.p) + ((obytes + 15) & ~(size_t)15))
                                           : h->host_flags;
            for (int q = 0; q < (int)nq; ++q)
                if (flags[q]) sel[n++] = q;
            if (n > 0) {
                h->host_fallbacks += n;
                const FilterArgs& fa = h->deferred_fa;
                HIP_TRY(h, hipMemcpyAsync(h->qsel.p, sel, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));  // `sel` is on this stack frame
                rc = run_exact(h, h->stream, fa.Qpad, fa.qaux, n, h->qsel.as<int32_t>(), 0, h->total, k, d_lab, d_dist, d_cnt, d_d64,
                               false);
                if (rc) return rc;
                continue;  // copy the corrected outputs
            }
        }
        if (!pinned) break;
        const char* po = static_cast<const char*>(h->pin_out.p);
        if (out_dist64) std::memcpy(out_dist64, po, b64);
        std::memcpy(out_labels, po + b64, blab);
        std::memcpy(out_dist, po + b64 + blab, bdist);
        std::memcpy(out_counts, po + b64 + blab + bdist, bcnt);
        break;
    }
    return MLVDB_OK;
}
}  // namespace

int mlvdb_search_batch_ex(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, const uint8_t* row_mask,
                          int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_dist64) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!row_mask || h->total == 0) return search_host(h, queries, nq, k, out_labels, out_dist, out_counts, out_dist64);
    HIP_TRY(h, h->row_mask.ensure((size_t)h->total));
    HIP_TRY(h, h->rn_masked.ensure((size_t)h->capacity * sizeof(float)));
    HIP_TRY(h, hipMemcpyAsync(h->row_mask.p, row_mask, (size_t)h->total, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, launch_mask_norms(h->rn, h->row_mask.as<uint8_t>(), h->rn_masked.as<float>(), h->total, h->capacity, h->stream));
    // the int8 shadow serves masked searches too: bring it up to date with the index's own norms, then mask a copy of its
    // row pairs (8 bytes per row) exactly like the norms
    h->mask_pairs_ready = false;
    if (i8_eligible(h) && h->strategy != MLVDB_STRATEGY_EXACT && use_filter(h, nq, true)) {  // (not for a call the exact scan takes)
        rc = update_i8_shadow(h, h->stream);
        if (rc) return rc;
        HIP_TRY(h, h->rp8_masked.ensure((size_t)h->capacity * (h->space == kSpaceL2 ? 3 : 2) * sizeof(float)));
        HIP_TRY(h, launch_mask_pairs(h->rp8.as<float>(), h->row_mask.as<uint8_t>(), h->rp8_masked.as<float>(), h->total,
                                     h->capacity, h->space == kSpaceL2 ? 1 : 0, h->stream));
        rc = forget_l2_offsets(h, 1, h->stream);
        if (rc) return rc;
        h->mask_pairs_ready = true;
    }
    float* const all_rows = h->rn;  // every kernel of the call reads the masked norms instead
    h->rn = h->rn_masked.as<float>();
    h->mask_active = true;
    rc = search_host(h, queries, nq, k, out_labels, out_dist, out_counts, out_dist64);
    h->rn = all_rows;
    h->mask_active = false;
    h->mask_pairs_ready = false;
    return rc;
    });
}

int mlvdb_search_batch(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, int64_t* out_labels,
                       float* out_dist, int32_t* out_counts) {
    return guarded(h, [&]() -> int {
    return mlvdb_search_batch_ex(h, queries, nq, k, nullptr, out_labels, out_dist, out_counts, nullptr);
    });
}

int mlvdb_search_batch_filtered(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, const uint8_t* row_mask,
                                int64_t* out_labels, float* out_dist, int32_t* out_counts) {
    return guarded(h, [&]() -> int {
    return mlvdb_search_batch_ex(h, queries, nq, k, row_mask, out_labels, out_dist, out_counts, nullptr);
    });
}

// Both range entries.  out_offsets == nullptr: the dense form (out_labels / out_dist are [nq, capacity]); otherwise the packed
// form: the hits of query i are entries out_offsets[i] .. out_offsets[i + 1] of out_labels / out_dist (total_capacity entries).
static int range_batch_impl(mlvdb_index* h, const float* queries, int64_t nq, float radius, int64_t capacity, int64_t* out_labels,
                            float* out_dist, int64_t* out_counts, int64_t* out_offsets, int64_t total_capacity, bool packed) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (packed && !out_offsets) return fail(h, MLVDB_ERR_INVALID_ARG, "null buffer");
    if (nq < 0 || nq > (1 << 24)) return fail(h, MLVDB_ERR_INVALID_ARG, "nq out of range");
    if (capacity < 1) return fail(h, MLVDB_ERR_INVALID_ARG, "capacity must be >= 1");
    if (out_offsets && total_capacity < 0) return fail(h, MLVDB_ERR_INVALID_ARG, "total_capacity must be >= 0");
    if (nq == 0) {
        if (out_offsets) out_offsets[0] = 0;
        return MLVDB_OK;
    }
    if (!queries || !out_counts || ((!out_labels || !out_dist) && (!out_offsets || total_capacity > 0)))
        return fail(h, MLVDB_ERR_INVALID_ARG, "null buffer");
    hipStream_t s = h->stream;
    rc = begin_call(h, s);
    if (rc) return rc;
    if (h->total == 0 || h->total == h->deleted) {
        for (int64_t i = 0; i < nq; ++i) out_counts[i] = 0;
        if (out_offsets)
            for (int64_t i = 0; i <= nq; ++i) out_offsets[i] = 0;
        return end_call(h, s);
    }
    HIP_TRY(h, h->io_q.ensure((size_t)nq * h->dim * sizeof(float)));
    HIP_TRY(h, h->qpad.ensure((size_t)nq * h->ld * sizeof(float)));
    HIP_TRY(h, h->qaux.ensure((size_t)nq * sizeof(double)));
    const int64_t cap_eff = std::min<int64_t>(capacity, MLVDB_MAX_TOPK_PAGED);  // most hits returned per query
    HIP_TRY(h, h->io_lab.ensure((size_t)nq * cap_eff * sizeof(int64_t)));
    HIP_TRY(h, h->io_dist.ensure((size_t)nq * cap_eff * sizeof(float)));
    HIP_TRY(h, h->io_cnt.ensure((size_t)nq * sizeof(int64_t)));
    HIP_TRY(h, h->pin_in.ensure(std::max((size_t)nq * h->dim * sizeof(float), ((size_t)nq + 1) * sizeof(int64_t))));
    std::memcpy(h->pin_in.p, queries, (size_t)nq * h->dim * sizeof(float));  // pinned staging: one DMA (search_host)
    HIP_TRY(h, hipMemcpyAsync(h->io_q.p, h->pin_in.p, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(h, h->qerr.ensure((size_t)nq * sizeof(float)));
    bool ready = false;
    rc = filter_ready(h, s, nq, &ready);
    if (rc) return rc;
    const bool filt = use_filter(h, nq, ready);
    if (!filt)  // (the filter passes prepare their own queries: one fused launch each)
        HIP_TRY(h, launch_query_prep(h->io_q.as<float>(), (int32_t)nq, h->dim, h->ld, h->space, h->qpad.as<float>(),
                                     h->qaux.as<double>(), h->qerr.as<float>(), s));
    h->stats.strategy_used = filt ? MLVDB_STRATEGY_FILTER : MLVDB_STRATEGY_EXACT;
    for (int64_t q0 = 0; q0 < nq; q0 += kFilterQueries) {
        const int32_t n = (int32_t)std::min<int64_t>(kFilterQueries, nq - q0);
        FilterArgs fa{};
        rc = setup_filter_ws(h, fa, h->qpad.as<float>() + (size_t)q0 * h->ld, h->qaux.as<double>() + q0,
                             h->qerr.as<float>() + q0, n);
        if (rc) return rc;
        // a range pass keeps its own, larger candidate lists: true hits + the band of the bound, per query anything
        // from none to tens of thousands
        HIP_TRY(h, h->cand_range.ensure((size_t)kFilterQueries * kRangeCandCap * sizeof(CandEntry)));
        HIP_TRY(h, h->rhits.ensure((size_t)kFilterQueries * kCandCap * sizeof(RangeHit)));
        HIP_TRY(h, h->rhit_cnt.ensure(kFilterQueries * sizeof(uint32_t)));
        fa.cand = h->cand_range.as<CandEntry>();
        fa.cand_cap = kRangeCandCap;
        fa.rhits = h->rhits.as<RangeHit>();
        fa.rhit_cnt = h->rhit_cnt.as<uint32_t>();
        // int8 bounds (half the scan time of the bf16 body; MLVDB_RANGE_I8=0 keeps the bf16 one): their band admits
        // ~8x more candidates than there are hits, which the chunked rescoring below absorbs
        if (filt && h->tn.range_i8 != 0) {
            rc = attach_i8(h, s, fa);
            if (rc) return rc;
        }
        if (filt) {  // (also clears the candidate counters)
            rc = prep_pass(h, s, fa, h->io_q.as<float>() + (size_t)q0 * h->dim, h->qpad.as<float>() + (size_t)q0 * h->ld,
                           h->qaux.as<double>() + q0, h->qerr.as<float>() + q0);
            if (rc) return rc;
        } else {
            HIP_TRY(h, launch_filter_prep(fa, s));  // per-query state only (no image: the exact range scan reads Qpad)
        }
        h->stats.bound_dtype = fa.X8 ? 2 : (filt ? 1 : 0);
        rc = scan_event(h, s, true);
        if (rc) return rc;
        if (filt) {
            HIP_TRY(h, launch_filter_range_thr(fa, radius, s));
            ScanInfo info;
            HIP_TRY(h, launch_filter_scan(fa, 0, h->total, s, &info));
            } else {
            HIP_TRY(h, launch_exact_range_scan(fa, radius, nullptr, 0, s));
        }
        rc = scan_event(h, s, false);
        if (rc) return rc;
        h->stats.scan_launches += 1;
        h->stats.rows_scanned += h->total;
        int32_t n_flagged = 0;
        rc = collect_overflow(h, s, fa, &n_flagged);
        if (rc) return rc;
        if (n_flagged) {
            // more candidates than list slots (or a wave buffer overflowed): the exact range scan restricted to those
            // queries stores exactly their hits; a list that now fits is complete, larger ones stay flagged
            h->stats.fallback_queries += n_flagged;
            const int32_t* qsel = h->qsel.as<int32_t>();
            range_reset_kernel<<<1, n_flagged, 0, s>>>(fa.cnt, qsel);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, launch_exact_range_scan(fa, radius, qsel, n_flagged, s));
            range_resolve_kernel<<<1, n_flagged, 0, s>>>(fa.overflow, fa.cnt, qsel, (uint32_t)fa.cand_cap);
            HIP_TRY(h, hipGetLastError());
        }
        // mid bounds first (round 4): the int8 band admits ~8x the hits; a pass over the candidates' fp16 rows (whole 128-byte
        // lines, half the bytes of the fp32 rows) prunes it to ~1.01x, so the exact gather below touches (nearly) only hits
        if (filt && h->tn.range_l2) {
            MidArgs m{};
            rc = attach_mid(h, s, &m);
            if (rc) return rc;
            if (m.X16) {
                HIP_TRY(h, launch_mid_score(fa, m, s));
                HIP_TRY(h, launch_bigk_thr_prune(fa, m, 0, -1, s));
            }
        }
        // exact fp64 distance of every candidate (blocks over query x candidate chunk), then sort + emit per query
        HIP_TRY(h, launch_range_rescore(fa, radius, (int32_t)q0, cap_eff, h->io_lab.as<int64_t>(),
                                        h->io_dist.as<float>(), h->io_cnt.as<int64_t>(), s));
        // queries with more hits than one block sorts (> kCandCap), or more than a list holds: their nearest
        // min(count, capacity) hits come from the paged exact kNN; cnt[q] holds the exact count in both cases
        int32_t n_still = 0;
        rc = collect_overflow(h, s, fa, &n_still);
        if (rc) return rc;
        if (n_still) {
            const int32_t* qsel = h->qsel.as<int32_t>();
            HIP_TRY(h, hipMemcpyAsync(h->host_flags, fa.cnt, kFilterQueries * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            int64_t kmax = 1;
            for (int q = 0; q < n; ++q)
                if (h->host_overflow[q]) kmax = std::max<int64_t>(kmax, std::min<int64_t>(h->host_flags[q], cap_eff));
            if (n_flagged == 0) h->stats.fallback_queries += n_still;
            HIP_TRY(h, h->seed_lab.ensure((size_t)kFilterQueries * kmax * sizeof(int64_t)));
            HIP_TRY(h, h->seed_dist.ensure((size_t)kFilterQueries * kmax * sizeof(float)));
            HIP_TRY(h, h->seed_cnt.ensure(kFilterQueries * sizeof(int32_t)));
            rc = run_paged_exact(h, s, fa.Qpad, fa.qaux, kFilterQueries, qsel, n_still, (int32_t)kmax,
                                 h->seed_lab.as<int64_t>(), h->seed_dist.as<float>(), h->seed_cnt.as<int32_t>(), nullptr);
            if (rc) return rc;
            range_fallback_commit_kernel<<<n_still, 256, 0, s>>>(qsel, fa.cnt, h->seed_lab.as<int64_t>(),
                                                                 h->seed_dist.as<float>(), (int32_t)kmax, (int32_t)q0, cap_eff,
                                                                 h->io_lab.as<int64_t>(), h->io_dist.as<float>(),
                                                                 h->io_cnt.as<int64_t>());
            HIP_TRY(h, hipGetLastError());
        }
    }
    // copy back.  Hit counts vary by orders of magnitude between queries, so the dense [nq, cap_eff] device
    // arrays are mostly padding: when the hits are a small part of them, pack the rows' valid prefixes on the
    // device and send only those (25 MB -> 0.4 MB at 256 queries x 8192 slots with 128 hits on average).
    HIP_TRY(h, h->pin_out.ensure((size_t)nq * sizeof(int64_t)));
    HIP_TRY(h, hipStreamSynchronize(s));
    HIP_TRY(h, hipMemcpyAsync(h->pin_out.p, h->io_cnt.p, (size_t)nq * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::vector<int64_t> counts(static_cast<const int64_t*>(h->pin_out.p), static_cast<const int64_t*>(h->pin_out.p) + nq);
    std::vector<int64_t> offsets((size_t)nq + 1, 0);
    for (int64_t i = 0; i < nq; ++i) offsets[(size_t)i + 1] = offsets[(size_t)i] + std::min<int64_t>(std::max<int64_t>(counts[i], 0), cap_eff);
    const int64_t total_hits = offsets[(size_t)nq];
    if (out_offsets) {
        // packed form: offsets and counts always; the hits when they fit (one packing kernel, one DMA, two copies out of pinned memory)
        std::memcpy(out_offsets, offsets.data(), ((size_t)nq + 1) * sizeof(int64_t));
        bool hard_p = false;
        for (int64_t i = 0; i < nq; ++i) {
            out_counts[i] = counts[i];
            hard_p |= counts[i] > cap_eff && capacity > cap_eff;
        }
        const bool fits = total_hits <= total_capacity;
        if (fits && total_hits > 0) {
            HIP_TRY(h, h->labels_in.ensure(((size_t)nq + 1) * sizeof(int64_t)));
            const size_t plab = (size_t)total_hits * sizeof(int64_t), pdst = (size_t)total_hits * sizeof(float);
            HIP_TRY(h, h->seed_lab.ensure(plab + pdst));
            const bool pinned = plab + pdst <= ((size_t)16 << 20);  // (pinned staging only while it stays small: search_host)
            if (pinned) HIP_TRY(h, h->pin_out.ensure(plab + pdst));
            int64_t* d_pl = h->seed_lab.as<int64_t>();
            float* d_pd = reinterpret_cast<float*>(h->seed_lab.as<char>() + plab);
            std::memcpy(h->pin_in.p, offsets.data(), ((size_t)nq + 1) * sizeof(int64_t));
            HIP_TRY(h, hipMemcpyAsync(h->labels_in.p, h->pin_in.p, ((size_t)nq + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
            range_pack_kernel<<<(unsigned)nq, 256, 0, s>>>(h->io_lab.as<int64_t>(), h->io_dist.as<float>(), h->labels_in.as<int64_t>(),
                                                           cap_eff, d_pl, d_pd);
            HIP_TRY(h, hipGetLastError());
            if (pinned) {
                HIP_TRY(h, hipMemcpyAsync(h->pin_out.p, d_pl, plab + pdst, hipMemcpyDeviceToHost, s));
                HIP_TRY(h, hipStreamSynchronize(s));
                std::memcpy(out_labels, h->pin_out.p, plab);
                std::memcpy(out_dist, static_cast<const char*>(h->pin_out.p) + plab, pdst);
            } else {
                HIP_TRY(h, hipMemcpyAsync(out_labels, d_pl, plab, hipMemcpyDeviceToHost, s));
                HIP_TRY(h, hipMemcpyAsync(out_dist, d_pd, pdst, hipMemcpyDeviceToHost, s));
                HIP_TRY(h, hipStreamSynchronize(s));
            }
        }
        rc = end_call(h, s);
        if (rc) return rc;
        if (!fits)
            return fail(h, MLVDB_ERR_OVERFLOW, "range query: more hits in all than total_capacity; out_counts / out_offsets hold "
                                               "the exact counts and the layout the hits need, out_labels / out_dist nothing");
        if (hard_p)
            return fail(h, MLVDB_ERR_UNSUPPORTED,
                        "range query: a query has more than MLVDB_MAX_TOPK_PAGED hits; out_counts holds the exact counts, the "
                        "outputs the nearest MLVDB_MAX_TOPK_PAGED");
        for (int64_t i = 0; i < nq; ++i)
            if (counts[i] > capacity)
                return fail(h, MLVDB_ERR_OVERFLOW, "some query has more hits than `capacity`; out_counts holds the exact counts");
        return MLVDB_OK;
    }
    if (total_hits * 4 < nq * cap_eff) {
        if (total_hits > 0) {
            HIP_TRY(h, h->labels_in.ensure(((size_t)nq + 1) * sizeof(int64_t)));
            // packed outputs: one device buffer [labels | distances], one DMA into pinned memory
            const size_t plab = (size_t)total_hits * sizeof(int64_t), pdst = (size_t)total_hits * sizeof(float);
            HIP_TRY(h, h->seed_lab.ensure(plab + pdst));
            HIP_TRY(h, h->pin_out.ensure(plab + pdst));
            int64_t* d_pl = h->seed_lab.as<int64_t>();
            float* d_pd = reinterpret_cast<float*>(h->seed_lab.as<char>() + plab);
            std::memcpy(h->pin_in.p, offsets.data(), ((size_t)nq + 1) * sizeof(int64_t));
            HIP_TRY(h, hipMemcpyAsync(h->labels_in.p, h->pin_in.p, ((size_t)nq + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
            range_pack_kernel<<<(unsigned)nq, 256, 0, s>>>(h->io_lab.as<int64_t>(), h->io_dist.as<float>(), h->labels_in.as<int64_t>(),
                                                           cap_eff, d_pl, d_pd);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipMemcpyAsync(h->pin_out.p, d_pl, plab + pdst, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            const int64_t* pl = static_cast<const int64_t*>(h->pin_out.p);
            const float* pd = reinterpret_cast<const float*>(static_cast<const char*>(h->pin_out.p) + plab);
            for (int64_t i = 0; i < nq; ++i) {
                const int64_t n = offsets[(size_t)i + 1] - offsets[(size_t)i];
                std::memcpy(out_labels + (size_t)i * capacity, pl + offsets[(size_t)i], (size_t)n * sizeof(int64_t));
                std::memcpy(out_dist + (size_t)i * capacity, pd + offsets[(size_t)i], (size_t)n * sizeof(float));
            }
        }
    } else {
        HIP_TRY(h, hipMemcpy2DAsync(out_labels, (size_t)capacity * sizeof(int64_t), h->io_lab.p, (size_t)cap_eff * sizeof(int64_t),
                                    (size_t)cap_eff * sizeof(int64_t), (size_t)nq, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpy2DAsync(out_dist, (size_t)capacity * sizeof(float), h->io_dist.p, (size_t)cap_eff * sizeof(float),
                                    (size_t)cap_eff * sizeof(float), (size_t)nq, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    bool over = false, hard = false;
    for (int64_t i = 0; i < nq; ++i) {
        out_counts[i] = counts[i];
        over |= counts[i] > capacity;
        hard |= counts[i] > cap_eff && capacity > cap_eff;
    }
    rc = end_call(h, s);
    if (rc) return rc;
    if (hard)
        return fail(h, MLVDB_ERR_UNSUPPORTED,
                    "range query: a query has more than MLVDB_MAX_TOPK_PAGED hits; out_counts holds the exact counts, the "
                    "outputs the nearest MLVDB_MAX_TOPK_PAGED");
    if (over) return fail(h, MLVDB_ERR_OVERFLOW, "some query has more hits than `capacity`; out_counts holds the exact counts");
    return MLVDB_OK;
}

int mlvdb_range_batch(mlvdb_index* h, const float* queries, int64_t nq, float radius, int64_t capacity,
                      int64_t* out_labels, float* out_dist, int64_t* out_counts) {
    return guarded(h, [&]() -> int {
    return range_batch_impl(h, queries, nq, radius, capacity, out_labels, out_dist, out_counts, nullptr, 0, false);
    });
}

int mlvdb_range_batch_packed(mlvdb_index* h, const float* queries, int64_t nq, float radius, int64_t capacity,
                             int64_t total_capacity, int64_t* out_labels, float* out_dist, int64_t* out_offsets,
                             int64_t* out_counts) {
    return guarded(h, [&]() -> int {
    return range_batch_impl(h, queries, nq, radius, capacity, out_labels, out_dist, out_counts, out_offsets, total_capacity, true);
    });
}

int mlvdb_pair_distances(mlvdb_index* h, const float* queries, int64_t nq, const int64_t* labels, int64_t m,
                         double* out_dist64, float* out_dist) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (nq < 0 || nq > (1 << 24) || m < 0 || m > (1 << 24)) return fail(h, MLVDB_ERR_INVALID_ARG, "nq / m out of range");
    if (nq == 0 || m == 0) return MLVDB_OK;
    if (!queries || !labels || !out_dist64) return fail(h, MLVDB_ERR_INVALID_ARG, "null buffer");
    for (int64_t i = 0; i < nq * m; ++i)
        if (labels[i] >= h->total) return fail(h, MLVDB_ERR_INVALID_ARG, "label out of range");
    hipStream_t s = h->aux_stream;  // rows are immutable once appended: may overlap a search in flight on h->stream
    std::lock_guard<std::mutex> aux_lock(h->aux_mutex);
    // private buffers of the auxiliary stream (gather_out / gather_lab) + query staging of its own
    const size_t qbytes = (size_t)nq * h->dim * sizeof(float), pbytes = (size_t)nq * h->ld * sizeof(float);
    const size_t abytes = (size_t)nq * sizeof(double), o64 = (size_t)nq * m * sizeof(double), o32 = (size_t)nq * m * sizeof(float);
    HIP_TRY(h, h->gather_lab.ensure((size_t)nq * m * sizeof(int64_t)));
    HIP_TRY(h, h->gather_out.ensure(qbytes + pbytes + abytes + o64 + o32 + 64));
    char* base = h->gather_out.as<char>();
    double* d_o64 = reinterpret_cast<double*>(base);  // 8-byte aligned pieces first
    double* d_aux = reinterpret_cast<double*>(base + o64);
    float* d_pad = reinterpret_cast<float*>(base + o64 + abytes);
    float* d_q = reinterpret_cast<float*>(base + o64 + abytes + pbytes);
    float* d_o32 = reinterpret_cast<float*>(base + o64 + abytes + pbytes + qbytes);
    HIP_TRY(h, hipMemcpyAsync(d_q, queries, qbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->gather_lab.p, labels, (size_t)nq * m * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(h, launch_query_prep(d_q, (int32_t)nq, h->dim, h->ld, h->space, d_pad, d_aux, nullptr, s));
    HIP_TRY(h, launch_pair_distances(h->X, d_pad, d_aux, h->gather_lab.as<int64_t>(), (int32_t)nq, (int32_t)m, h->ld, h->space,
                                     d_o64, out_dist ? d_o32 : nullptr, s));
    HIP_TRY(h, hipMemcpyAsync(out_dist64, d_o64, o64, hipMemcpyDeviceToHost, s));
    if (out_dist) HIP_TRY(h, hipMemcpyAsync(out_dist, d_o32, o32, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MLVDB_OK;
    });
}

int mlvdb_index_set_strategy(mlvdb_index* h, int32_t strategy) {
    return guarded(h, [&]() -> int {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    if (strategy < 0 || strategy > 2) return fail(h, MLVDB_ERR_INVALID_ARG, "unknown strategy");
    if (strategy == MLVDB_STRATEGY_FILTER && !filter_supported(h->ld) && h->ld8 == 0)
        return fail(h, MLVDB_ERR_UNSUPPORTED, "filter strategy needs an index created with a shadow (NO_SHADOW / I8_PAD=0 took it away)");
    h->strategy = strategy;
    return MLVDB_OK;
    });
}

int mlvdb_index_set_tuning(mlvdb_index* h, const char* assignment) {
    return guarded(h, [&]() -> int {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    const char* eq = assignment ? strchr(assignment, '=') : nullptr;
    if (!eq || eq == assignment) return fail(h, MLVDB_ERR_INVALID_ARG, "tuning assignment must be KEY=VALUE");
    const TuningField* f = find_tuning_field(assignment, (size_t)(eq - assignment));
    if (!f) return fail(h, MLVDB_ERR_INVALID_ARG, "unknown tuning key");
    if (!strcmp(f->name, "NO_SHADOW") || !strcmp(f->name, "SHADOW_BF16") || !strcmp(f->name, "I8_PAD"))
        return fail(h, MLVDB_ERR_UNSUPPORTED, "creation-time knob: set MLVDB_<KEY> in the environment before mlvdb_index_create");
    char* end = nullptr;
    const long v = strtol(eq + 1, &end, 10);
    if (end == eq + 1 || *end != '\0') return fail(h, MLVDB_ERR_INVALID_ARG, "tuning value must be an integer");
    h->tn.*(f->field) = (int)v;
    return MLVDB_OK;
    });
}

int mlvdb_index_get_tuning(const mlvdb_index* h, const char* key, int32_t* value) {
    return guarded(const_cast<mlvdb_index*>(h), [&]() -> int {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    if (!key || !value) return fail(const_cast<mlvdb_index*>(h), MLVDB_ERR_INVALID_ARG, "null key / value");
    const TuningField* f = find_tuning_field(key, strlen(key));
    if (!f) return fail(const_cast<mlvdb_index*>(h), MLVDB_ERR_INVALID_ARG, "unknown tuning key");
    *value = h->tn.*(f->field);
    return MLVDB_OK;
    });
}

int mlvdb_index_set_profiling(mlvdb_index* h, int32_t enabled) {
    return guarded(h, [&]() -> int {
    if (!h) return fail(nullptr, MLVDB_ERR_INVALID_ARG, "null index handle");
    h->profiling = enabled != 0;
    return MLVDB_OK;
    });
}

int mlvdb_index_last_stats(mlvdb_index* h, mlvdb_stats* out) {
    return guarded(h, [&]() -> int {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(h, MLVDB_ERR_INVALID_ARG, "out is null");
    if (h->counters_pending) {
        unsigned long long c[2] = {0, 0};
        HIP_TRY(h, hipMemcpyAsync(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost, h->counters_stream));
        HIP_TRY(h, hipStreamSynchronize(h->counters_stream));
        h->stats.candidates_rescored = (int64_t)c[0];
        h->stats.fallback_queries = (int64_t)c[1];
        h->counters_pending = false;
    }
    if (h->stats_pending) {
        HIP_TRY(h, hipEventSynchronize(h->total_events[1]));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->total_events[0], h->total_events[1]));
        h->stats.total_ms = ms;
        double scan = 0.0;
        for (size_t i = 0; i < h->scan_events_used; ++i) {
            HIP_TRY(h, hipEventElapsedTime(&ms, h->scan_events[i].first, h->scan_events[i].second));
            scan += ms;
        }
        h->stats.scan_ms = scan;
        h->stats_pending = false;
    }
    h->stats.fallback_queries += h->host_fallbacks;  // fallbacks decided on the host (search_host)
    h->host_fallbacks = 0;
    *out = h->stats;
    // reset: the next call starts a new accumulation window
    const int32_t strategy = h->stats.strategy_used;
    h->stats = mlvdb_stats{};
    h->stats.strategy_used = strategy;
    h->scan_events_used = 0;
    return MLVDB_OK;
    });
}

}  // extern "C"
