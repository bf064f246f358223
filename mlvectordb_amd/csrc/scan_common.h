// Device helpers shared by the exact-scan and rescoring kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"
#include "wave_topk.h"

namespace mlvdb {

// Exact fp64 accumulation of PW rows against QT queries over all column groups.
//
// Lane l = 16*g + r works on row r of each of its PW rows and on the column slice
// {16*kb + 4*g + j : j < 4} of every group kb.  `base[p]` already points at this lane's
// float4 of group 0 of row p (layout.h), successive groups are 256 floats apart.
// `qs` is the query tile in LDS as fp64, [QT][ld].  Products of two fp32 values are exact
// in fp64, so the only rounding is in the sums; the order (j ascending inside a group,
// groups ascending, then g via xor-16 / xor-32) is the same in every kernel that calls
// this, which makes a row's distance bit-identical wherever it is computed.
// One column group: x[p] holds this lane's float4 of row p.
template <int SPACE, int QT, int PW>
__device__ __forceinline__ void accumulate_group(const float4 (&x)[PW], const double* qs, int ld, int g, int kb,
                                                 double (&acc)[PW][QT], double (&nx)[PW]) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const double* qp = qs + t * ld + kb * 16 + g * 4;
        const double2 q01 = *reinterpret_cast<const double2*>(qp);
        const double2 q23 = *reinterpret_cast<const double2*>(qp + 2);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            if (SPACE == kSpaceL2) {
                double e;
                e = q01.x - (double)x[p].x; acc[p][t] = __builtin_fma(e, e, acc[p][t]);
                e = q01.y - (double)x[p].y; acc[p][t] = __builtin_fma(e, e, acc[p][t]);
                e = q23.x - (double)x[p].z; acc[p][t] = __builtin_fma(e, e, acc[p][t]);
                e = q23.y - (double)x[p].w; acc[p][t] = __builtin_fma(e, e, acc[p][t]);
            } else {
                acc[p][t] = __builtin_fma(q01.x, (double)x[p].x, acc[p][t]);
                acc[p][t] = __builtin_fma(q01.y, (double)x[p].y, acc[p][t]);
                acc[p][t] = __builtin_fma(q23.x, (double)x[p].z, acc[p][t]);
                acc[p][t] = __builtin_fma(q23.y, (double)x[p].w, acc[p][t]);
            }
        }
    }
    if (SPACE == kSpaceCosine) {
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            nx[p] = __builtin_fma((double)x[p].x, (double)x[p].x, nx[p]);
            nx[p] = __builtin_fma((double)x[p].y, (double)x[p].y, nx[p]);
            nx[p] = __builtin_fma((double)x[p].z, (double)x[p].z, nx[p]);
            nx[p] = __builtin_fma((double)x[p].w, (double)x[p].w, nx[p]);
        }
    }
}

// PF = 0: plain loop (high-occupancy callers: other waves hide the latency).
// PF > 0: the groups are fetched in blocks of PF, two register banks, the next block is in
//         flight while the current one is consumed (low-occupancy callers: seed scan, rescoring).
//         The summation order is the same for every PF.
// NT: non-temporal loads (the streaming scans read every row once: +8 % on the batch-1 scan of 1M x 768).
template <int SPACE, int QT, int PW, int PF = 0, bool NT = false>
__device__ __forceinline__ void accumulate_rows(const float* const (&base)[PW], const double* qs, int ld, int g,
                                                double (&acc)[PW][QT], double (&nx)[PW]) {
    const int nkb = ld >> 4;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        nx[p] = 0.0;
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[p][t] = 0.0;
    }
    if (PF == 0) {
#pragma unroll QT * PW <= 4 ? 2 : 1
        for (int kb = 0; kb < nkb; ++kb) {
            float4 x[PW];
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v* src = reinterpret_cast<const f4v*>(base[p] + (int64_t)kb * kGroupFloats);
                const f4v v = NT ? __builtin_nontemporal_load(src) : *src;
                x[p] = make_float4(v.x, v.y, v.z, v.w);
            }
            accumulate_group<SPACE, QT, PW>(x, qs, ld, g, kb, acc, nx);
        }
    } else {
        constexpr int B = PF > 0 ? PF : 1;
        const int nblk = nkb / B;
        float4 bank0[B][PW], bank1[B][PW];
        auto fetch = [&](float4(&bank)[B][PW], int blk) __attribute__((always_inline)) {
            const int bb = blk < nblk ? blk : (nblk > 0 ? nblk - 1 : 0);  // clamped: loads stay unconditional
#pragma unroll
            for (int i = 0; i < B; ++i)
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v* src = reinterpret_cast<const f4v*>(base[p] + (int64_t)(bb * B + i) * kGroupFloats);
                    const f4v v = NT ? __builtin_nontemporal_load(src) : *src;
                    bank[i][p] = make_float4(v.x, v.y, v.z, v.w);
                }
        };
        auto consume = [&](const float4(&bank)[B][PW], int blk) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < B; ++i) accumulate_group<SPACE, QT, PW>(bank[i], qs, ld, g, blk * B + i, acc, nx);
        };
        if (nblk > 0) {
            fetch(bank0, 0);
            for (int blk = 0; blk < nblk; blk += 2) {
                fetch(bank1, blk + 1);
                consume(bank0, blk);
                fetch(bank0, blk + 2);
                if (blk + 1 < nblk) consume(bank1, blk + 1);
            }
        }
        for (int kb = nblk * B; kb < nkb; ++kb) {  // remainder groups
            float4 x[PW];
#pragma unroll
            for (int p = 0; p < PW; ++p) x[p] = *reinterpret_cast<const float4*>(base[p] + (int64_t)kb * kGroupFloats);
            accumulate_group<SPACE, QT, PW>(x, qs, ld, g, kb, acc, nx);
        }
    }
    // combine the four column slices of each row: (g0+g1) + (g2+g3), identical in all 4 lanes
#pragma unroll
    for (int p = 0; p < PW; ++p) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            double v = acc[p][t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[p][t] = v;
        }
        if (SPACE == kSpaceCosine) {
            double v = nx[p];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            nx[p] = v;
        }
    }
}

// fp64 distance in the index's space from the accumulated sums (include/mlvdb_hip.h).
// qinv = 1/(|q|+1e-30) for cosine (unused otherwise).
template <int SPACE>
__device__ __forceinline__ double finish_distance(double acc, double nx, double qinv) {
    if (SPACE == kSpaceL2) return acc;
    if (SPACE == kSpaceIp) return 1.0 - acc;
    const double xinv = 1.0 / (__builtin_sqrt(nx) + 1e-30);
    return 1.0 - (acc * qinv) * xinv;
}

}  // namespace mlvdb
