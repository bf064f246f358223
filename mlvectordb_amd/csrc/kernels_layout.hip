// Corpus maintenance kernels: row-major <-> panel layout, row norms, tombstones, query prep.
// Reference roles: hnswlib add_items / mark_deleted as called from
// src/mlvectordb/implementations/index.py:65,80,158 (the index copies and owns the rows).
#include <algorithm>

#include "internal.h"

namespace mlvdb {

// One thread per (row, 4-column group) of the padded row.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ stage, float* __restrict__ X,
                                                           int64_t first_row, int64_t n, int32_t dim, int32_t ld) {
    const int32_t ngrp = ld >> 2;
    const int64_t total = n * ngrp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / ngrp;
        const int32_t col = (int32_t)(idx - r * ngrp) << 2;
        const float* src = stage + r * dim + col;
        float4 v;
        if ((dim & 3) == 0) {
            v = col < dim ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v.x = col + 0 < dim ? src[0] : 0.f;
            v.y = col + 1 < dim ? src[1] : 0.f;
            v.z = col + 2 < dim ? src[2] : 0.f;
            v.w = col + 3 < dim ? src[3] : 0.f;
        }
        *reinterpret_cast<float4*>(X + layout_offset(first_row + r, col, ld)) = v;
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));

// One thread per (row, 8-column group): fp32 panels -> bf16 shadow, round-to-nearest-even
// (the same v_cvt_pk_bf16_f32 the scan kernel uses when it converts in registers).
__global__ __launch_bounds__(256) void shadow_rows_kernel(const float* __restrict__ X, __bf16* __restrict__ Xb,
                                                          int64_t first_row, int64_t n, int32_t ld) {
    const int32_t ngrp = ld >> 3;
    const int64_t total = n * ngrp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = first_row + idx / ngrp;
        const int32_t col = (int32_t)(idx % ngrp) << 3;
        const float4 lo = *reinterpret_cast<const float4*>(X + layout_offset(r, col, ld));
        const float4 hi = *reinterpret_cast<const float4*>(X + layout_offset(r, col + 4, ld));
        const f32x8_t v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        *reinterpret_cast<bf16x8_t*>(Xb + layout_offset_b(r, col, ld)) = __builtin_convertvector(v, bf16x8_t);
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, float* __restrict__ out,
                                                          int64_t first_row, int64_t n, int32_t dim, int32_t ld) {
    const int64_t total = n * dim;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / dim;
        const int32_t col = (int32_t)(idx - r * dim);
        out[idx] = X[layout_offset(first_row + r, col, ld)];
    }
}

// out[i][:] = row labels[i]: one block per output row, coalesced on the output side
__global__ __launch_bounds__(256) void gather_rows_at_kernel(const float* __restrict__ X, float* __restrict__ out,
                                                             const int64_t* __restrict__ labels, int32_t dim, int32_t ld) {
    const int64_t row = labels[blockIdx.x];
    for (int32_t col = threadIdx.x; col < dim; col += blockDim.x)
        out[(int64_t)blockIdx.x * dim + col] = X[layout_offset(row, col, ld)];
}

// One wave per panel; lane 16*g + r sums the squares of row r over its column slices.
// Also maintains *rel_err_max = max over rows of |x - bf16(x)| / |x| (RNE, the conversion the filter scan and the
// shadow use): the row half of the filter's rounding bound (kernels_filter.hip).  Non-negative floats order like
// their bit patterns, so the maximum is an atomicMax on the bits.
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ X, float* __restrict__ rn,
                                                        int64_t first_row, int64_t n, int32_t ld,
                                                        unsigned int* __restrict__ rel_err_max) {
    const int lane = threadIdx.x & 63;
    const int64_t first_panel = first_row >> 4;
    const int64_t last_panel = (first_row + n - 1) >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nkb = ld >> 4;
    for (int64_t panel = first_panel + wave; panel <= last_panel; panel += nwaves) {
        const float* base = X + panel * (int64_t)(kPanelRows * ld) + lane_group_offset(lane);
        double s = 0.0, err = 0.0;
        for (int kb = 0; kb < nkb; ++kb) {
            const float4 x = *reinterpret_cast<const float4*>(base + (int64_t)kb * kGroupFloats);
            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s = __builtin_fma((double)xs[i], (double)xs[i], s);
                const double e = (double)xs[i] - (double)(float)(__bf16)xs[i];
                err = __builtin_fma(e, e, err);
            }
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        err += __shfl_xor(err, 16);
        err += __shfl_xor(err, 32);
        const int64_t row = panel * kPanelRows + (lane & 15);
        if (lane < 16 && row >= first_row && row < first_row + n) {
            rn[row] = (float)__builtin_sqrt(s);
            if (s > 0.0 && err > 0.0) {
                float rel = (float)(__builtin_sqrt(err / s) * 1.000001);
                rel = __uint_as_float(__float_as_uint(rel) + 1u);  // (float) may have rounded down
                atomicMax(rel_err_max, __float_as_uint(rel));
            }
        }
    }
}

__global__ __launch_bounds__(256) void tombstone_kernel(float* __restrict__ rn, const int64_t* __restrict__ labels,
                                                        int64_t n, int64_t total, unsigned long long* changed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t label = labels[i];
    if (label < 0 || label >= total) return;
    // exchange so that a label listed twice in one call is counted once
    const unsigned int old = atomicExch(reinterpret_cast<unsigned int*>(rn) + label, 0x7fc00000u);
    const float oldf = __uint_as_float(old);
    if (oldf == oldf) atomicAdd(changed, 1ull);
}

// One block per query: zero-padded copy + fp64 norm.
// qerr[q] (optional) = |q^ - q^_b| rounded up, where q^ = q / (|q| + 1e-30) and q^_b is the bf16 query image exactly
// as filter_prep_kernel computes it (fp32 product with the fp32 inverse norm, then RNE to bf16): the query half of
// the filter's rounding bound.
__global__ __launch_bounds__(256) void query_prep_kernel(const float* __restrict__ queries, int32_t dim, int32_t ld,
                                                         int32_t space, float* __restrict__ Qpad,
                                                         double* __restrict__ qaux, float* __restrict__ qerr) {
    __shared__ double red[4];
    __shared__ double inv_s;
    const int q = blockIdx.x;
    const float* src = queries + (int64_t)q * dim;
    float* dst = Qpad + (int64_t)q * ld;
    double s = 0.0;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        const float v = c < dim ? src[c] : 0.f;
        dst[c] = v;
        s = __builtin_fma((double)v, (double)v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nrm = __builtin_sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const double aux = space == kSpaceCosine ? 1.0 / (nrm + 1e-30) : nrm;
        qaux[q] = aux;
        inv_s = space == kSpaceCosine ? aux : 1.0 / (aux + 1e-30);  // what filter_prep_kernel multiplies by
    }
    if (!qerr) return;
    __syncthreads();
    const double inv = inv_s;
    double e2 = 0.0;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        const float v = src[c];
        const double e = (double)v * inv - (double)(float)(__bf16)(v * (float)inv);
        e2 = __builtin_fma(e, e, e2);
    }
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e2;
    __syncthreads();
    if (threadIdx.x == 0) {
        float e = (float)(__builtin_sqrt((red[0] + red[1]) + (red[2] + red[3])) * 1.000001);
        qerr[q] = __uint_as_float(__float_as_uint(e) + 1u);  // (float) may have rounded down; never below the true error
    }
}

static inline int grid_for(int64_t work, int threads, int cap = 256 * 8) {
    int64_t g = (work + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_scatter_rows(const float* stage, float* X, int64_t first_row, int64_t n, int32_t dim, int32_t ld,
                               hipStream_t s) {
    if (n <= 0) return hipSuccess;
    scatter_rows_kernel<<<grid_for(n * (ld >> 2), 256), 256, 0, s>>>(stage, X, first_row, n, dim, ld);
    return hipGetLastError();
}

hipError_t launch_shadow_rows(const float* X, void* Xb, int64_t first_row, int64_t n, int32_t ld, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    shadow_rows_kernel<<<grid_for(n * (ld >> 3), 256), 256, 0, s>>>(X, static_cast<__bf16*>(Xb), first_row, n, ld);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* X, float* out, int64_t first_row, int64_t n, int32_t dim, int32_t ld,
                              hipStream_t s) {
    if (n <= 0) return hipSuccess;
    gather_rows_kernel<<<grid_for(n * dim, 256), 256, 0, s>>>(X, out, first_row, n, dim, ld);
    return hipGetLastError();
}

hipError_t launch_gather_rows_at(const float* X, float* out, const int64_t* labels, int64_t n, int32_t dim, int32_t ld,
                                 hipStream_t s) {
    if (n <= 0) return hipSuccess;
    gather_rows_at_kernel<<<(unsigned)n, 256, 0, s>>>(X, out, labels, dim, ld);
    return hipGetLastError();
}

hipError_t launch_row_norms(const float* X, float* rn, int64_t first_row, int64_t n, int32_t ld, unsigned int* rel_err_max,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t panels = ((first_row + n - 1) >> 4) - (first_row >> 4) + 1;
    row_norms_kernel<<<grid_for(panels, 4), 256, 0, s>>>(X, rn, first_row, n, ld, rel_err_max);
    return hipGetLastError();
}

hipError_t launch_tombstone(float* rn, const int64_t* labels, int64_t n, int64_t total, unsigned long long* changed,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    tombstone_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(rn, labels, n, total, changed);
    return hipGetLastError();
}

hipError_t launch_query_prep(const float* queries, int32_t nq, int32_t dim, int32_t ld, int32_t space, float* Qpad,
                             double* qaux, float* qerr, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    query_prep_kernel<<<nq, 256, 0, s>>>(queries, dim, ld, space, Qpad, qaux, qerr);
    return hipGetLastError();
}

// ------------------------------------------------------------------ compaction (drop tombstoned rows)
// Live rows (rn == rn) keep their order and move to labels 0..live-1.  Three small passes build
// old_of_new[new label] = old label (block counts -> one-block scan -> per-block ranks), then one
// wave per destination panel gathers the fp32 rows (whole 64-byte pieces, layout.h), the bf16 shadow
// pieces and the norms.
constexpr int kCompactRowsPerBlock = 1024;  // 256 threads x 4 rows

__global__ __launch_bounds__(256) void compact_count_kernel(const float* __restrict__ rn, int64_t total,
                                                            uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wsum[4];
    const int64_t r0 = (int64_t)blockIdx.x * kCompactRowsPerBlock + threadIdx.x * 4;
    uint32_t c = 0;
    for (int i = 0; i < 4; ++i)
        if (r0 + i < total) {
            const float v = rn[r0 + i];
            c += v == v ? 1u : 0u;
        }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one block: block_counts -> exclusive offsets in place; *live = total count
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ block_counts, int64_t nblocks,
                                                            unsigned long long* __restrict__ live) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nblocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_counts[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < nblocks) block_counts[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *live = carry_s;
}

__global__ __launch_bounds__(256) void compact_map_kernel(const float* __restrict__ rn, int64_t total,
                                                          const uint32_t* __restrict__ block_offsets,
                                                          int32_t* __restrict__ old_of_new) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kCompactRowsPerBlock + threadIdx.x * 4;
    bool alive[4];
    uint32_t c = 0;
    for (int i = 0; i < 4; ++i) {
        const float v = r0 + i < total ? rn[r0 + i] : __builtin_nanf("");
        alive[i] = v == v;
        c += alive[i] ? 1u : 0u;
    }
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pos = block_offsets[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    for (int i = 0; i < 4; ++i)
        if (alive[i]) old_of_new[pos++] = (int32_t)(r0 + i);
}

// one wave per destination panel
__global__ __launch_bounds__(256) void compact_rows_kernel(const float* __restrict__ X, float* __restrict__ nX,
                                                           const char* __restrict__ Xb, char* __restrict__ nXb,
                                                           const float* __restrict__ rn, float* __restrict__ nrn,
                                                           const int32_t* __restrict__ old_of_new, int64_t live, int32_t ld) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int64_t npanels = (live + 15) >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t panel_floats = (int64_t)kPanelRows * ld;
    for (int64_t panel = wave; panel < npanels; panel += nwaves) {
        const int64_t dst_row = panel * 16 + r;
        const bool have = dst_row < live;
        const int64_t src = have ? old_of_new[dst_row] : 0;
        const float* sp = X + (src >> 4) * panel_floats + (src & 15) * 16 + g * 4;
        float* dp = nX + panel * panel_floats + r * 16 + g * 4;
        for (int kb = 0; kb < (ld >> 4); ++kb)
            if (have) *reinterpret_cast<float4*>(dp + (int64_t)kb * kGroupFloats) = *reinterpret_cast<const float4*>(sp + (int64_t)kb * kGroupFloats);
        if (Xb) {  // shadow: 32-column groups of 1 KiB, [g][r][8 bf16]
            const char* sb = Xb + ((src >> 4) * panel_floats) * 2 + g * 256 + (src & 15) * 16;
            char* db = nXb + (panel * panel_floats) * 2 + g * 256 + r * 16;
            for (int kb = 0; kb < (ld >> 5); ++kb)
                if (have) *reinterpret_cast<uint4*>(db + (int64_t)kb * 1024) = *reinterpret_cast<const uint4*>(sb + (int64_t)kb * 1024);
        }
        if (have && g == 0) nrn[dst_row] = rn[src];
    }
}

hipError_t launch_compact_map(const float* rn, int64_t total, uint32_t* block_scratch, unsigned long long* live,
                              int32_t* old_of_new, hipStream_t s) {
    const int64_t nblocks = (total + kCompactRowsPerBlock - 1) / kCompactRowsPerBlock;
    if (nblocks == 0) return hipMemsetAsync(live, 0, sizeof(unsigned long long), s);
    compact_count_kernel<<<(unsigned)nblocks, 256, 0, s>>>(rn, total, block_scratch);
    compact_scan_kernel<<<1, 1024, 0, s>>>(block_scratch, nblocks, live);
    compact_map_kernel<<<(unsigned)nblocks, 256, 0, s>>>(rn, total, block_scratch, old_of_new);
    return hipGetLastError();
}

hipError_t launch_compact_rows(const float* X, float* nX, const void* Xb, void* nXb, const float* rn, float* nrn,
                               const int32_t* old_of_new, int64_t live, int32_t ld, hipStream_t s) {
    const int64_t npanels = (live + 15) >> 4;
    if (npanels == 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((npanels + 3) / 4, 256 * 8);
    compact_rows_kernel<<<(unsigned)blocks, 256, 0, s>>>(X, nX, static_cast<const char*>(Xb), static_cast<char*>(nXb), rn, nrn,
                                                          old_of_new, live, ld);
    return hipGetLastError();
}

// ------------------------------------------------------------------ per-call row mask (filtered search)
// out[i] = mask[i] ? rn[i] : NaN for i < total, NaN beyond: every scan kernel then treats a masked-out row
// exactly like a tombstoned one (seeding, thresholds, admission, rescoring), with no kernel knowing about masks.
__global__ __launch_bounds__(256) void mask_norms_kernel(const float* __restrict__ rn, const uint8_t* __restrict__ mask,
                                                         float* __restrict__ out, int64_t total, int64_t capacity) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (i < total && mask[i]) ? rn[i] : __builtin_nanf("");
}

// the same for the int8 shadow's row pairs (NaN pair = not a row)
// (l2 pairs keep their x-slot -- the group's scale / error, which the other rows of the group need -- and lose only |x|)
__global__ __launch_bounds__(256) void mask_pairs_kernel(const float2* __restrict__ rp, const uint8_t* __restrict__ mask,
                                                         float2* __restrict__ out, int64_t total, int64_t capacity, int l2) {
    const float nan = __builtin_nanf("");
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 pr = rp[i];
        out[i] = (i < total && mask[i]) ? pr : make_float2(l2 ? pr.x : nan, nan);
    }
}

hipError_t launch_mask_pairs(const float* rp8, const uint8_t* mask, float* out, int64_t total, int64_t capacity, int l2,
                             hipStream_t s) {
    if (capacity == 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((capacity + 255) / 256, 256 * 16);
    mask_pairs_kernel<<<(unsigned)blocks, 256, 0, s>>>(reinterpret_cast<const float2*>(rp8), mask, reinterpret_cast<float2*>(out),
                                                       total, capacity, l2);
    return hipGetLastError();
}

hipError_t launch_mask_norms(const float* rn, const uint8_t* mask, float* out, int64_t total, int64_t capacity, hipStream_t s) {
    if (capacity == 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((capacity + 255) / 256, 256 * 16);
    mask_norms_kernel<<<(unsigned)blocks, 256, 0, s>>>(rn, mask, out, total, capacity);
    return hipGetLastError();
}

}  // namespace mlvdb
