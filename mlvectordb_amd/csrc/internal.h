// Internal declarations shared by the translation units of libmlvdb_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/mlvdb_hip.h"
#include "layout.h"
#include "wave_topk.h"

namespace mlvdb {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is bound per device: one bit per device ordinal per call site, so a
// second index on another GPU of the same process (Index(devices=[...])) configures its own copy of the kernel, and
// host threads serving different handles do not race on a plain bool (setting the attribute twice is harmless).
inline hipError_t ensure_dynamic_lds(std::atomic<uint64_t>& done, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// ---------------------------------------------------------------- tuning state (per handle)
// Every knob the kernels' launchers and the pass orchestration consult.  Filled ONCE, by mlvdb_index_create, from the
// MLVDB_* environment variables of that moment (tuning_from_env, api.hip: the library's only getenv loop), and changed
// afterwards only through mlvdb_index_set_tuning(h, "KEY=VAL") -- nothing on the search path reads the environment, so
// host threads that serve different handles (MultiDeviceEngine: one per shard) never race a setenv in glibc.
// Variants marked [AB] exist only in `make AB=1` builds (tools/scan_ab.py, the `ab`-marked tests); the default
// library ignores them.
#define MLVDB_TUNING_FIELDS(X)                                                                                          \
    X(i8, "I8", 1)                     /* 0: bf16 bodies on an index that keeps a bf16 shadow */                       \
    X(no_shadow, "NO_SHADOW", 0)       /* creation: no shadow at all (the filter converts fp32 rows in registers) */   \
    X(shadow_bf16, "SHADOW_BF16", 0)   /* creation: keep the bf16 shadow beside the int8 one (MLVDB_SHADOW=bf16) */    \
    X(i8_pad, "I8_PAD", 1)             /* creation: int8 shadow zero-padded to a multiple of 256 columns for any dim */\
    X(i8_err_l2, "I8_ERR_L2", 1100)    /* l2: largest relative row error (thousandths) of the FEW odd groups an int8 index may hold (30: none; a row quantised to all zeros has 1.0) */ \
    X(i8_err_ip, "I8_ERR_IP", 30)      /* ip: largest index-wide relative row error (thousandths) the int8 bounds are used with */ \
    X(small_batch, "SMALL_BATCH", 1)   /* one query on a small corpus: ONE scan round after an 11,520-row exact prefix (0: the rounds) */ \
    X(small_seed, "SMALL_SEED", 1)     /* 1-2 queries: exact prefix seed */                                             \
    X(small_finish, "SMALL_FINISH", 1) /* 1-2 queries: fused last refine + rescoring + ranking */                       \
    X(small_nq, "SMALL_NQ", 2)                                                                                          \
    X(seed_rows, "SEED_ROWS", 5)       /* dense seeding pass, units of 768 rows */                                      \
    X(seed_exact, "SEED_EXACT", 0)                                                                                      \
    X(seed_i8, "SEED_I8", 1)                                                                                            \
    X(round1, "ROUND1", 85)            /* ends of the first / second scan round, units of 768 rows */                   \
    X(round2, "ROUND2", 2048)                                                                                           \
    X(refine_picks, "REFINE_PICKS", 0) /* 0 = 1.5 k, at least 16 */                                                     \
    X(pinned_io, "PINNED_IO", 1)                                                                                        \
    X(event_fence, "EVENT_FENCE", 0)                                                                                    \
    X(range_i8, "RANGE_I8", 1)                                                                                          \
    X(range_flat, "RANGE_FLAT", 1)                                                                                      \
    X(range_l2, "RANGE_L2", 1)         /* range passes: fp16 second-level bound before the exact gather */             \
    X(bigk, "BIGK", 1)                 /* top_k in (64, 1024] on the filter path (0: paged exact scan) */              \
    X(bigk_budget, "BIGK_BUDGET", 400000) /* entries one scan launch of a big-k pass may append (sizes its rounds) */ \
    X(l2_shadow, "L2_SHADOW", 1)       /* fp16 row-major shadow for second-level bounds (built lazily; 0: never) */    \
    X(debug_entries, "DEBUG_ENTRIES", 0)                                                                                \
    X(debug_refine, "DEBUG_REFINE", 0)                                                                                  \
    X(scan_narrow, "SCAN_NARROW", 1)                                                                                    \
    X(narrow_i8_max, "NARROW_I8_MAX", 0) /* largest batch the int8 NARROW kernel scans (round 4: none -- the 4-tile assembly body is faster at every batch size) */                                                                                \
    X(narrow_wgs, "NARROW_WGS", 0)     /* 0 = as many workgroups per CU as the image leaves LDS for */                  \
    X(narrow_balance, "NARROW_BALANCE", 1)                                                                              \
    X(scan_xcd, "SCAN_XCD", 0)                                                                                          \
    X(scan_asm, "SCAN_ASM", 1)                                                                                          \
    X(scan_l2c, "SCAN_L2C", 1)         /* l2: one query quantisation step per pass + one error coefficient: cosine's one-constant test */ \
    X(scan_l2e, "SCAN_L2E", 1)         /* l2: folded admission test with per-row integer offsets (0: serial test, round 3's body) */ \
    X(scan_nqt, "SCAN_NQT", 0)         /* query tiles of the int8 body: 0 = by batch size, else 4 / 8 / 16 */          \
    X(scan_nw, "SCAN_NW", 8)           /* [AB] */                                                                       \
    X(scan_mt, "SCAN_MT", 2)           /* [AB] */                                                                       \
    X(scan_va, "SCAN_VA", 1)           /* [AB] 0: AccVGPR accumulators, serial admission test (round 1) */             \
    X(scan_var, "SCAN_VAR", 0)         /* 237: round 2's body; the other codes [AB] */                                  \
    X(scan_prio, "SCAN_PRIO", -1)      /* [AB] -1 = the body's default */                                               \
    X(scan_nt, "SCAN_NT", 1)           /* [AB] */                                                                       \
    X(scan_dma, "SCAN_DMA", 1)         /* [AB] */                                                                       \
    X(scan_stag, "SCAN_STAG", 0)       /* [AB] */                                                                       \
    X(scan_diag, "SCAN_DIAG", 0)       /* make DIAG=1 builds */                                                         \
    X(l2_offset_cache, "L2_OFFSET_CACHE", 1) /* l2: keep the offsets plane across passes of the same scale (0: recompute per pass) */ \
    X(exact_nt, "EXACT_NT", 1)                                                                                          \
    X(exact_nblk, "EXACT_NBLK", 0)                                                                                      \
    X(prefix_pf, "PREFIX_PF", 24)                                                                                       \
    X(prefix_waves, "PREFIX_WAVES", 4)

struct Tuning {
#define X(field, name, dflt) int field = dflt;
    MLVDB_TUNING_FIELDS(X)
#undef X
};

// ---------------------------------------------------------------- layout kernels (kernels_layout.hip)
// stage: row-major [n, dim] on the device -> panels; rows first_row..first_row+n-1
hipError_t launch_scatter_rows(const float* stage, float* X, int64_t first_row, int64_t n, int32_t dim, int32_t ld,
                               hipStream_t s);
// Xb (bf16 shadow, layout_offset_b) for the same rows, from the fp32 panels
hipError_t launch_shadow_rows(const float* X, void* Xb, int64_t first_row, int64_t n, int32_t ld, hipStream_t s);
// rn[row] = (float)|x_row| for the same rows (fp64 sum of squares); *rel_err_max = max(*rel_err_max, |x - bf16(x)| / |x|)
// as float bits (rounded up)
hipError_t launch_row_norms(const float* X, float* rn, int64_t first_row, int64_t n, int32_t ld, unsigned int* rel_err_max,
                            hipStream_t s);
// panels -> row-major [n, dim]
hipError_t launch_gather_rows(const float* X, float* out, int64_t first_row, int64_t n, int32_t dim, int32_t ld,
                              hipStream_t s);
// labelled rows -> row-major [n, dim] (labels on the device, all within [0, total))
hipError_t launch_gather_rows_at(const float* X, float* out, const int64_t* labels, int64_t n, int32_t dim, int32_t ld,
                                 hipStream_t s);
// rn[label] = NaN for each valid, live label; *changed += number of rows that changed state
hipError_t launch_tombstone(float* rn, const int64_t* labels, int64_t n, int64_t total, unsigned long long* changed,
                            hipStream_t s);
// compaction: old_of_new[new label] = old label of the live rows (rn == rn) in order, *live = their count;
// block_scratch holds ceil(total / 1024) words
hipError_t launch_compact_map(const float* rn, int64_t total, uint32_t* block_scratch, unsigned long long* live,
                              int32_t* old_of_new, hipStream_t s);
// gather the live rows (fp32 panels, bf16 shadow when Xb != nullptr, norms) into freshly zeroed / NaN-filled buffers
hipError_t launch_compact_rows(const float* X, float* nX, const void* Xb, void* nXb, const float* rn, float* nrn,
                               const int32_t* old_of_new, int64_t live, int32_t ld, hipStream_t s);
// out[i] = mask[i] ? rn[i] : NaN (i < total), NaN up to capacity: a masked-out row looks tombstoned to every scan
hipError_t launch_mask_norms(const float* rn, const uint8_t* mask, float* out, int64_t total, int64_t capacity, hipStream_t s);
// the same for the int8 shadow's row pairs [rows][2]
hipError_t launch_mask_pairs(const float* rp8, const uint8_t* mask, float* out, int64_t total, int64_t capacity, int l2,
                             hipStream_t s);
// Qpad[q][0..ld) = queries[q][0..dim) zero padded; qaux[q] = 1/(|q|+1e-30) (cosine) or |q| (l2, ip)
// qerr (optional): |q^ - bf16 image of q^| per query, rounded up
hipError_t launch_query_prep(const float* queries, int32_t nq, int32_t dim, int32_t ld, int32_t space, float* Qpad,
                             double* qaux, float* qerr, hipStream_t s);

// ---------------------------------------------------------------- exact scan (kernels_exact.hip)
struct ExactPlan {
    int qt;        // queries per block tile (1, 2, 4, 8)
    int nblk;      // blocks along the corpus
    int nqtiles;   // blocks along the query batch
    int threads;   // block size
    size_t lds_bytes;
};
ExactPlan plan_exact(int64_t nrows, int32_t ld, int32_t nq_sel, int32_t k, const Tuning& tn);

struct ExactArgs {
    const float* X;
    const float* rn;
    int64_t row_begin;   // first row scanned (multiple of 16)
    int64_t row_end;     // one past the last row scanned (<= total)
    int32_t ld;
    int32_t space;
    const float* Qpad;   // [nq][ld]
    const double* qaux;  // [nq]
    const int32_t* qsel; // [nq_sel] query indices to process, or nullptr for 0..nq_sel-1
    int32_t nq_sel;
    const int32_t* nq_sel_dev;  // optional: the actual count lives on the device (<= nq_sel); blocks beyond it exit
    int32_t k;
    const double* cursor_d;   // optional paging cursor per query (nullptr = none):
    const int32_t* cursor_l;  //   only rows strictly after (cursor_d, cursor_l) in rank order are admitted
    TopEntry* partial;        // [nq_sel][nblk][k]
    const Tuning* tn;         // host only
};
hipError_t launch_exact_scan(const ExactArgs& a, const ExactPlan& p, hipStream_t s);
// merge partial lists -> final outputs at the original query index
hipError_t launch_exact_merge(const TopEntry* partial, int32_t nq_sel, const int32_t* nq_sel_dev, const int32_t* qsel,
                              int32_t nblk, int32_t k,
                              int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_d64,
                              hipStream_t s);

// exact fp64 distances of rows 0..m-1 to every query (tombstoned rows: +inf): d64 = [nq][m]
hipError_t launch_prefix_exact(const float* X, const float* rn, const float* Qpad, const double* qaux, int32_t nq, int32_t m,
                               int32_t ld, int32_t space, double* d64, const Tuning& tn, hipStream_t s);
// exact fp64 distances of given pairs: out[q][j] = d(query q, row labels[q][j]) (labels on the device, each < total or < 0 = +inf)
hipError_t launch_pair_distances(const float* X, const float* Qpad, const double* qaux, const int64_t* labels, int32_t nq,
                                 int32_t m, int32_t ld, int32_t space, double* out64, float* out32, hipStream_t s);

// ---------------------------------------------------------------- filter path (kernels_filter.hip)
constexpr int kFilterQueries = 256;   // queries per filter pass
constexpr int kFilterChunkK = 64;     // columns per Q chunk staged in LDS
constexpr int kCandCap = 8192;        // candidate slots per query (kNN passes; also the most range hits sorted in LDS)
constexpr int kRangeCandCap = 65536;  // candidate slots per query of a range pass: true hits + the bound's band
constexpr int kRangeChunk = 256;      // candidates one block of the range rescoring scores (4 gather steps: the gathers are
                                      // latency-bound, so a long list is spread over many resident blocks rather than looped over)
constexpr int kFilterTile = 768;      // scan ranges start on multiples of it (common multiple of the kernels' 192/128/256-row tiles)
constexpr int kScanMaxGrid = 512;     // most workgroups any scan launch uses
constexpr int kWgCap = 16384;         // append slots per workgroup and launch (assembly scan: split evenly over its waves)

struct CandEntry {
    float u;      // upper bound of the row's score (higher = nearer)
    int32_t row;
};

struct RangeHit {
    double d;
    int32_t l;
    int32_t pad;
};

// What a workgroup of the assembly scan appends to its private buffer; the scatter kernel then moves
// the entries into the per-query lists (the scan itself never waits for a device-scope atomic).
struct WgEntry {
    float u;
    int32_t row;
    uint32_t q;
    uint32_t pad;
};

bool filter_supported(int32_t ld);
size_t filter_qimg_bytes(int32_t ld);   // bf16 query image for one pass of kFilterQueries

struct FilterArgs {
    const Tuning* tn;       // host only: the handle's tuning state (never dereferenced on the device)
    const float* X;
    const void* Xb;         // bf16 shadow of X (layout_offset_b) or nullptr: the scan then converts fp32 in registers
    const float* rn;
    int64_t total;
    int32_t ld;
    int32_t ld8;            // columns of the int8 shadow and of the int8 query image: round_up(ld, 256), zero padded (0: no int8 shadow)
    int32_t space;
    const float* Qpad;      // [nq][ld] raw queries of this pass (q0..q0+nq)
    const double* qaux;
    const float* qerr;      // [nq] rounding error of the bf16 query image, |q^ - q^_b| (rounded up)
    const float* row_err;   // device scalar: max over rows of |x - bf16(x)| / |x| (rounded up)
    int32_t nq;             // <= kFilterQueries
    // workspace (per pass)
    void* qimg;             // bf16 image, filter_qimg_bytes
    float* qscale;          // [256] per-query multiplier of the dot product in score units
    float* thr;             // [256] admission threshold (lower bound of the k-th best score)
    float* ke;              // [256] per-query error term of the bound: cosine E1q + 2 slack, ip / l2 E1q + slack (x |x|)
    uint32_t* cnt;          // [256] candidates appended
    uint32_t* overflow;     // [256] nonzero = list overflowed, query must be re-run exactly
    CandEntry* cand;        // [256][cand_cap]
    int32_t cand_cap;       // kCandCap (kNN passes) or kRangeCandCap (range passes: their own, larger lists)
    struct RangeHit* rhits; // range passes: [256][kCandCap] exact hits (fp64 distance, row) found by range_score_kernel
    uint32_t* rhit_cnt;     // [256] exact hit count per query (may exceed kCandCap: the excess is counted, not stored)
    // int8 shadow (cosine, ld % 256 == 0; MLVDB_I8=0 disables): all null / unused otherwise
    const void* X8;         // int8 rows, per-row scale: panels of 16 rows, 64-column groups of 1 KiB (layout_offset_i8)
    const float* rp8;       // [rows][2] cosine {scale/(|x|+1e-30), row error}, l2 / ip {scale, |x|}; NaN = tombstoned
    int64_t rp8_cap;        // l2: rows the rp8 array was allocated for; behind its pairs (float index 2 * rp8_cap) lies the plane of
                            // per-row int32 offsets of the folded l2 admission test (filter_l2_offsets_kernel).  0 = none
    int32_t l2c;            // l2, common query scale for the pass: 0 = off
    float* l2c_out;         // l2: {SQ, KEq, KEr} of the pass, written by filter_l2_offsets_kernel, read by the l2c scan bodies
    uint32_t* l2tag;        // l2: {pass scale the offsets plane behind rp8 holds (float bits), its rows, block counter, -} or nullptr
    float* rmaxq;           // l2: [256] largest raw component of each query of the pass (fused prep -> filter_prep8_l2c_kernel)
    const float* row_err8;  // device scalar: max over rows of |x - scale * x8| / |x| (rounded up)
    void* qimg8;            // int8 query image
    float* sq8;             // [256] scale of the query image
    float* ke8;             // [257] cosine: the query's own int8 error term (row errors are per row); [256] = K = (1 + max eq8)/min sq8
    float* keb;             // [256] the bf16 error term: what the (bf16) seeding pass adds, while `ke` covers both kinds of entry
    unsigned int* sqmin;    // two device scalars: bits of the smallest sq8 / of the largest query error of the pass's queries
    struct RangeHit* rs;    // kNN passes: [256][kCandCap] exact (distance, label) of every rescored candidate (filter_rescore_score_kernel -> _rank_kernel)
    WgEntry* wgbuf;         // [kScanMaxGrid][kWgCap] append buffers of one scan launch, one slice per wave
    uint32_t* wgcnt;        // [kScanMaxGrid * 8] entries appended per wave (may exceed the slice: the excess was flagged as overflow)
};
hipError_t launch_filter_prep(const FilterArgs& a, hipStream_t s);
// query_prep + filter_prep + filter_prep8 (when a.X8) of one pass in one launch (+ the one-block fin): `queries` = the pass's raw
// [a.nq][dim] rows on the device; writes Qpad / qaux / qerr (= a.Qpad / a.qaux / a.qerr) and everything launch_filter_prep(8) does.
// a.sqmin[] must hold {0x7f7f7f7f, 0} on entry (set when the workspace is allocated, restored by every fin kernel).
hipError_t launch_filter_prep_fused(const FilterArgs& a, const float* queries, int32_t dim, float* Qpad, double* qaux, float* qerr,
                                    hipStream_t s);
// seed thresholds from exact kNN distances of a prefix of the corpus: seed_d64[q][k]
// thr[q] from the k-th smallest of d64[q][0..m) (m <= 3 kSeedRows; fewer than k finite values: thr stays as it is)
hipError_t launch_filter_prefix_thr(const FilterArgs& a, const double* d64, int32_t m, int32_t k, hipStream_t s);
hipError_t launch_filter_seed_thr(const FilterArgs& a, const double* seed_d64, int32_t k, hipStream_t s);
// What a scan launch reports back (tuning aids).  The assembly scan stages its hits per wave in LDS and its own tail
// moves them into the per-query candidate lists (round 1 had a separate scatter launch for that).
struct ScanInfo {
    int nw = 0;
    int dbg = 0;
    int i8 = 0;  // int8 scan, entries in units of the query's scale: 1 cosine (u = w sq8 + ke), 2 ip (u = w sq8)
};
// int8 shadow: (re)build the panels covering rows [row_begin, row_end) (also rp8 and the index-wide error), the query image of a
// pass (after launch_filter_prep: overrides ke with the int8 error term), exact thresholds from the k best bounds
hipError_t launch_shadow8_rows(const float* X, const float* rn, void* X8, float* rp8, float* row_err8, int64_t row_begin,
                               int64_t row_end, int32_t ld, int32_t ld8, int32_t space, hipStream_t s);
hipError_t launch_filter_prep8(const FilterArgs& a, hipStream_t s);
// l2, once per pass (after the prep kernels): the integer offsets e_j of rows [0, rows) that fold the row term -|x|^2 of the l2
// score into the scan's accumulators (tools/gen_scan_asm.py, l2e): needs a.rp8_cap > 0
hipError_t launch_filter_l2_offsets(const FilterArgs& a, int64_t rows, hipStream_t s);
bool filter_refine_can_fuse(const FilterArgs& a);
bool filter_narrow_ok(const FilterArgs& a);  // the pass's scans run on the narrow kernel (<= 64 queries, image resident in LDS)
hipError_t launch_filter_refine_thr(const FilterArgs& a, int32_t k, int32_t forced_cnt, bool fuse, hipStream_t s);
// batches of <= 8 queries: the last refine + exact rescoring + ranking + output in one launch (needs filter_refine_can_fuse)
hipError_t launch_filter_finish_small(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                      int32_t* out_counts, double* out_d64, unsigned long long* rescored, int32_t* qsel,
                                      int32_t* nflag, hipStream_t s);
hipError_t launch_filter_scan(const FilterArgs& a, int64_t row_begin, int64_t row_end, hipStream_t s, ScanInfo* info);
// dense seeding pass over rows [0,row_end), row_end <= kSeedRows: all bounds -> candidate lists -> thresholds (update)
constexpr int kSeedRows = 3840;  // a multiple of every scan tile (128, 192) and <= kCandCap
hipError_t launch_filter_seed_scan(const FilterArgs& a, int64_t row_end, int32_t k, hipStream_t s);
hipError_t launch_filter_update(const FilterArgs& a, int32_t k, hipStream_t s);
// rescored[0] += rescored pairs; qsel != nullptr: also compacts the overflowed queries of the pass, qsel[0..*nflag) = their
// indices, rescored[1] += *nflag (the device-decided exact fallback that follows reads them)
hipError_t launch_filter_rescore(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                 int32_t* out_counts, double* out_d64, unsigned long long* rescored, int32_t* qsel,
                                 int32_t* nflag, hipStream_t s);
// range variant: fixed per-query threshold from the radius, then exact rescoring with emit
hipError_t launch_filter_range_thr(const FilterArgs& a, float radius, hipStream_t s);
// exact candidate generator for range queries (any dim): appends every live row with dist <= radius
// qsel/nsel: restrict to these queries of the pass (nullptr = all a.nq); cnt[q] ends as the exact hit count
hipError_t launch_exact_range_scan(const FilterArgs& a, float radius, const int32_t* qsel, int32_t nsel, hipStream_t s);
hipError_t launch_range_rescore(const FilterArgs& a, float radius, int32_t q0, int64_t capacity, int64_t* out_labels,
                                float* out_dist, int64_t* out_counts, hipStream_t s);
// The dense pass alone: every (query, row) bound of rows [0, rows) -> slot `row` of the query's list (rows a multiple of 128,
// <= a.cand_cap).  launch_filter_seed_scan = this + the first refine / update; big-k passes seed up to 65,280 rows with it.
hipError_t launch_filter_dense_scan(const FilterArgs& a, int64_t rows, hipStream_t s);
// kNN ending on the range kernels (big-k passes): exact fp64 distance of every list entry (range_score_flat_kernel with an
// infinite radius), then the ranking kernel in kNN mode: the k nearest by (distance, label), int32 counts, padded tails,
// optional fp64 distances; more than kCandCap live entries flag the query (overflow = 2) for the paged exact scan.
hipError_t launch_knn_rescore_rank(const FilterArgs& a, int32_t k, int32_t q0, int64_t* out_labels, float* out_dist,
                                   int32_t* out_counts, double* out_d64, unsigned long long* rescored, hipStream_t s);

// ---------------------------------------------------------------- mid bounds + big-k candidate lists (kernels_refine.hip)
constexpr uint32_t kRefinedBit = 0x80000000u;  // CandEntry::row bit 31: u is the entry's mid (fp16) upper bound, not the scan's
constexpr int kBigKMax = 1024;                 // largest top_k of a filter pass (beyond: paged exact scan)
constexpr int kBigSeedRows = 65280;            // dense seeding pass of a big-k pass: 85 x 768 rows (<= kRangeCandCap)
constexpr int kPicksCap = 2048;                // pick-list slots per query (>= kBigKMax + kBigKMax / 4 + 32)

struct MidArgs {
    const _Float16* X16;     // row-major fp16 shadow [rows][ld16]: x ~ s16[row] * h
    const float* s16;        // [rows] scale of the row
    const float* row_err16;  // device scalar: max over rows of |x - s16 h| / |x| (rounded up)
    int32_t ld16;            // round_up(dim, 64): 128 bytes per row and step
    const uint32_t* picks;   // [256][picks_cap] list indices to refine, or nullptr: every entry without a mid bound
    const uint32_t* npicks;  // [256]
    int32_t picks_cap;
};
hipError_t launch_shadow16_rows(const float* X, void* X16, float* s16, float* row_err16, int64_t row_begin, int64_t row_end,
                                int32_t ld, int32_t ld16, hipStream_t s);
hipError_t launch_mid_score(const FilterArgs& a, const MidArgs& m, hipStream_t s);
hipError_t launch_bigk_select(const FilterArgs& a, int32_t want, int32_t forced_cnt, uint32_t* picks, uint32_t* npicks,
                              int32_t picks_cap, hipStream_t s);
// k > 0: threshold from the k-th largest mid lower bound, then prune; k == 0: prune only (range passes)
hipError_t launch_bigk_thr_prune(const FilterArgs& a, const MidArgs& m, int32_t k, int32_t forced_cnt, hipStream_t s);

}  // namespace mlvdb
