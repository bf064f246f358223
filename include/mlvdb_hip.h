/*
 * mlvdb_hip.h -- C ABI of the MI355X (gfx950) exhaustive-scan index.
 *
 * This is the drop-in boundary for MLVectorDB's brute-force kNN hot path.  One
 * `mlvdb_index` is what the reference keeps per namespace as one `hnswlib.Index`
 * (reference: src/mlvectordb/implementations/index.py:19,32-48): a dense label space
 * 0..total-1 in insertion order, fp32 rows, tombstones, nearest-first kNN.  The
 * reference reaches hnswlib through pybind11; a maintainer swapping in this library
 * binds the functions below with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every function returns an int status (MLVDB_OK == 0), never throws, never calls
 *     back into the host language;
 *   - all pointers are plain host pointers unless the name ends in `_device`;
 *   - outputs are caller-allocated; row-major; labels are int64 (hnswlib: uint64 labels);
 *   - distances follow hnswlib 0.8.0's spaces (reference pin: pyproject.toml:12):
 *       l2     = sum_i (q_i - x_i)^2                  (squared, no sqrt)
 *       cosine = 1 - <q,x> / ((|q|+1e-30)(|x|+1e-30))
 *       ip     = 1 - <q,x>
 *     computed exhaustively, ranked on the fp64 value of the fp32 inputs, ties broken
 *     by ascending label, returned as fp32;
 *   - a handle is not thread-safe; callers serialise per handle (the reference is
 *     single-threaded: rest_api.py:164-184).
 */
#ifndef MLVDB_HIP_H
#define MLVDB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLVDB_ABI_VERSION 6

/* status codes */
#define MLVDB_OK 0
#define MLVDB_ERR_INVALID_ARG 1   /* bad pointer / size / enum */
#define MLVDB_ERR_DIM_MISMATCH 2  /* reference: hnswlib RuntimeError on wrong dim (index.py:54-55,65) */
#define MLVDB_ERR_NO_DEVICE 3     /* no usable HIP device: the product path fails loudly, no CPU fallback */
#define MLVDB_ERR_HIP 4           /* a HIP runtime call failed; see mlvdb_last_error */
#define MLVDB_ERR_OUT_OF_MEMORY 5
#define MLVDB_ERR_UNSUPPORTED 6   /* e.g. top_k above MLVDB_MAX_TOPK */
#define MLVDB_ERR_INTERNAL 8      /* an internal consistency check failed */
#define MLVDB_ERR_OVERFLOW 7      /* range query: some query had more hits than `capacity` (counts still exact) */

/* distance spaces; replaces hnswlib.Index(space=...) at index.py:36 */
#define MLVDB_SPACE_L2 0
#define MLVDB_SPACE_COSINE 1
#define MLVDB_SPACE_IP 2

/* largest top_k one scan selects (one list entry per wavefront lane); larger top_k is served in
 * rank-ordered pages of this size by the exact scan, up to MLVDB_MAX_TOPK_PAGED */
#define MLVDB_MAX_TOPK 64
#define MLVDB_MAX_TOPK_PAGED 16384

/* search strategy (mlvdb_index_set_strategy); AUTO picks per call */
#define MLVDB_STRATEGY_AUTO 0
#define MLVDB_STRATEGY_EXACT 1   /* fp64 streaming scan only */
#define MLVDB_STRATEGY_FILTER 2  /* int8- or bf16-MFMA bound filter + exact fp64 rescoring (falls back to EXACT per query) */

typedef struct mlvdb_index mlvdb_index;

/* Statistics of the search/range calls made since the previous mlvdb_index_last_stats on the handle. */
typedef struct mlvdb_stats {
    int32_t strategy_used;        /* MLVDB_STRATEGY_EXACT or MLVDB_STRATEGY_FILTER */
    int32_t scan_launches;        /* launches of the dominant scan kernel in the call */
    int64_t rows_scanned;         /* rows streamed from HBM by those launches (sum) */
    int64_t candidates_rescored;  /* (query,row) pairs that reached the exact rescoring kernel */
    int64_t fallback_queries;     /* queries re-run on the exact scan after a candidate-list overflow */
    double scan_ms;               /* summed HIP-event time of the scan launches (0 unless profiling is on) */
    double total_ms;              /* HIP-event time of the whole call on its stream (0 unless profiling is on) */
    int32_t bound_dtype;          /* filter strategy: 1 = bf16 shadow, 2 = int8 shadow (cosine) computed the bounds; else 0 */
    int32_t reserved;
} mlvdb_stats;

int mlvdb_abi_version(void);

/* Number of visible HIP devices (0 and MLVDB_ERR_NO_DEVICE when none). */
int mlvdb_device_count(int* count);

/* Message for the last failing call made without a handle (create). Thread-local. */
const char* mlvdb_last_global_error(void);

/*
 * Create an empty index for `dim`-dimensional fp32 rows on HIP device `device`.
 * Replaces hnswlib.Index(space, dim) + init_index (index.py:36-38); there is no
 * max_elements cap (the reference's 10,000 is not reproduced): capacity grows with
 * HBM.  `capacity_hint` rows are reserved up front (0 = grow on demand).
 */
int mlvdb_index_create(int device, int32_t dim, int32_t space, int64_t capacity_hint, mlvdb_index** out);
int mlvdb_index_destroy(mlvdb_index* h);

/* Human-readable message for the last failing call on this handle. */
const char* mlvdb_last_error(const mlvdb_index* h);

/*
 * Append n rows (host, row-major [n, dim], fp32).  The index copies the data
 * (hnswlib owns its storage: index.py:65).  Labels are total..total+n-1
 * (index.py:56-60); *first_label receives the first one.
 */
int mlvdb_index_append(mlvdb_index* h, const float* rows, int64_t n, int64_t* first_label);
/* Same, rows already resident on this index's device. */
int mlvdb_index_append_device(mlvdb_index* h, const float* rows_device, int64_t n, int64_t* first_label);

/*
 * Tombstone labels (hnswlib mark_deleted: index.py:80).  Unknown / already deleted
 * labels are ignored; *newly_deleted receives how many rows changed state.
 */
int mlvdb_index_tombstone(mlvdb_index* h, const int64_t* labels, int64_t n, int64_t* newly_deleted);

/*
 * Drop the tombstoned rows on the device: the live rows keep their order and become labels
 * 0..live-1, exactly the labels a rebuild from the surviving vectors in insertion order would
 * assign (Index.rebuild: index.py:145-162; QueryProcessor.delete's trigger: query_processor.py:58-61)
 * -- without sending the corpus over PCIe again.  old_labels[i] receives the former label of new
 * label i (capacity >= total - deleted entries), *live the new total.
 */
int mlvdb_index_compact(mlvdb_index* h, int64_t* old_labels, int64_t capacity, int64_t* live);

/* total rows ever appended, and how many of them are tombstoned (index.py:27-28,103). */
int mlvdb_index_counts(const mlvdb_index* h, int64_t* total, int64_t* deleted);

/* Drop every row, keep dim; `space` < 0 keeps the current space (Index.rebuild: index.py:131-162). */
int mlvdb_index_reset(mlvdb_index* h, int32_t space);

/* Copy rows [first, first+n) back to the host in row-major order (tombstoned rows included). */
int mlvdb_index_get_rows(mlvdb_index* h, int64_t first, int64_t n, float* out_rows);
/* Copy the rows of `labels` (host, n entries, any order, repeats allowed) back to the host: out_rows [n, dim].
 * What a caller without a second copy of the corpus uses to enrich the hits of a query wave with their values
 * (QueryProcessor.find_similar's "values" field: query_processor.py:42-47). */
int mlvdb_index_get_rows_at(mlvdb_index* h, const int64_t* labels, int64_t n, float* out_rows);

/*
 * Batched exact kNN.  Replaces hnswlib knn_query (index.py:111) for nq >= 1 queries.
 *   queries     [nq, dim] fp32
 *   k           1..MLVDB_MAX_TOPK_PAGED; the caller clamps to the live count (index.py:107);
 *               if fewer than k live rows exist the tail is padded (label -1, distance +inf)
 *   out_labels  [nq, k] int64, nearest first
 *   out_dist    [nq, k] fp32 distances in this index's space
 *   out_counts  [nq]    int32 number of valid entries per query (= min(k, live))
 */
int mlvdb_search_batch(mlvdb_index* h, const float* queries, int64_t nq, int32_t k,
                       int64_t* out_labels, float* out_dist, int32_t* out_counts);
/* Same with every buffer on the index's device; work is enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = the null stream) and is complete when the
 * stream is: the call itself only synchronises when a fallback decision needs it.
 * The handle's workspaces are reused by every call: use one stream at a time per handle.
 * out_dist64_device (optional, may be NULL) receives the unrounded fp64 distances [nq, k]:
 * a caller that merges per-shard results must rank on these, not on the fp32 roundings. */
int mlvdb_search_batch_device(mlvdb_index* h, const float* queries_device, int64_t nq, int32_t k,
                              int64_t* out_labels_device, float* out_dist_device,
                              int32_t* out_counts_device, double* out_dist64_device, void* stream);

/*
 * mlvdb_search_batch restricted to the rows with row_mask[label] != 0 (host, `total` bytes): the
 * pre-computed row mask of a metadata-filtered search (README.md:121,130,252,274: intent only, no
 * reference code).  A masked-out row is treated exactly like a tombstoned one for this call;
 * results are the exact k nearest allowed live rows.  row_mask == NULL: no restriction.
 */
int mlvdb_search_batch_filtered(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, const uint8_t* row_mask,
                                int64_t* out_labels, float* out_dist, int32_t* out_counts);

/*
 * The general host-pointer form: row_mask (optional, NULL = no restriction) as in mlvdb_search_batch_filtered, and
 * out_dist64 (optional, NULL = not wanted) [nq, k] the unrounded fp64 distances -- what a caller that merges the
 * answers of several indexes (row shards of one namespace: one mlvdb_index per GPU) must rank on.
 */
int mlvdb_search_batch_ex(mlvdb_index* h, const float* queries, int64_t nq, int32_t k, const uint8_t* row_mask,
                          int64_t* out_labels, float* out_dist, int32_t* out_counts, double* out_dist64);

/*
 * Batched range query: every live row with distance <= radius (distance in the
 * index's space, so squared radius for l2), nearest first, ties by ascending label.
 * No reference implementation exists (README-only); semantics defined in DESIGN.md.
 *   capacity    entries available per query in out_labels / out_dist
 *   out_counts  [nq] int64 exact number of hits per query, even when > capacity
 * Returns MLVDB_ERR_OVERFLOW (outputs hold the nearest `capacity` hits) when any
 * query had more hits than capacity, and MLVDB_ERR_UNSUPPORTED when a query has more than
 * MLVDB_MAX_TOPK_PAGED hits and more than that were asked for (counts are still exact).
 */
int mlvdb_range_batch(mlvdb_index* h, const float* queries, int64_t nq, float radius, int64_t capacity,
                      int64_t* out_labels, float* out_dist, int64_t* out_counts);

/*
 * The same query with packed outputs (ABI version 6): hit counts differ by orders of magnitude between queries, so the dense
 * [nq, capacity] arrays of mlvdb_range_batch are mostly padding.  Here the hits of query i are the entries
 * out_offsets[i] .. out_offsets[i + 1] of out_labels / out_dist (nearest first), at most `capacity` per query.
 *   total_capacity  entries available in out_labels / out_dist (both may be NULL when it is 0: a counting call)
 *   out_offsets     [nq + 1] int64, always written: out_offsets[nq] = the entries the hits need in all
 *   out_counts      [nq] int64 exact number of hits per query, even when > capacity
 * Returns MLVDB_ERR_OVERFLOW when the hits need more than total_capacity entries (nothing is written to out_labels /
 * out_dist; call again with out_offsets[nq] entries) or when a query had more hits than `capacity` (outputs hold the nearest
 * `capacity` of each), MLVDB_ERR_UNSUPPORTED as mlvdb_range_batch.
 */
int mlvdb_range_batch_packed(mlvdb_index* h, const float* queries, int64_t nq, float radius, int64_t capacity,
                             int64_t total_capacity, int64_t* out_labels, float* out_dist, int64_t* out_offsets,
                             int64_t* out_counts);

/*
 * Exact distances of given (query, row) pairs, in the index's space: out_dist64[q*m + j] = distance of queries[q] to the
 * row labels[q*m + j] (fp64; out_dist, optional, its fp32 rounding).  The vector-level face of the search arithmetic --
 * what the reference's README calls SimpleVector.distance() / similarity() (README.md:30-41,178-181; no reference code)
 * -- computed by the same summation as the scans, so a pair scored here equals, bit for bit, the distance a search
 * returns for it.  A label < 0 (search padding) gives +inf; labels >= total are an error; tombstoned rows still score.
 */
int mlvdb_pair_distances(mlvdb_index* h, const float* queries, int64_t nq, const int64_t* labels, int64_t m,
                         double* out_dist64, float* out_dist);

/* Strategy override (testing / benchmarking); default MLVDB_STRATEGY_AUTO. */
int mlvdb_index_set_strategy(mlvdb_index* h, int32_t strategy);

/*
 * Tuning state of the handle (A/B runs and tests; every setting returns the same ids).  The MLVDB_<KEY> environment
 * variables are read ONCE, by mlvdb_index_create; afterwards the library never consults the environment -- a caller
 * that wants another setting on a live handle says so here: `assignment` = "KEY=VALUE" (integer value; the
 * "MLVDB_" prefix is optional), keys as in DESIGN.md ("Tuning knobs").  Creation-time knobs (NO_SHADOW, SHADOW_BF16,
 * I8_PAD) are refused with MLVDB_ERR_UNSUPPORTED, unknown keys with MLVDB_ERR_INVALID_ARG.
 * No reference counterpart (hnswlib's only run-time knob is set_ef, index.py:38, which an exhaustive scan has no use for).
 */
int mlvdb_index_set_tuning(mlvdb_index* h, const char* assignment);
int mlvdb_index_get_tuning(const mlvdb_index* h, const char* key, int32_t* value);

/* Turn HIP-event timing of the scan kernels on/off (off by default: events add launch overhead). */
int mlvdb_index_set_profiling(mlvdb_index* h, int32_t enabled);

/* Statistics accumulated since the previous call of this function (waits for the pending events /
 * device counters of those calls), then starts a new accumulation window. */
int mlvdb_index_last_stats(mlvdb_index* h, mlvdb_stats* out);

/*
 * Layout introspection, used by the tests to pin the HBM panel layout against its
 * restatement: element (row, col) of the corpus lives at this float offset.
 */
int64_t mlvdb_layout_offset(int64_t row, int32_t col, int32_t ld);
/* Padded row length (floats) used for `dim`. */
int32_t mlvdb_layout_ld(int32_t dim);

#ifdef __cplusplus
}
#endif
#endif /* MLVDB_HIP_H */
