// Wave-level sorted selection list: lane i of a 64-lane wavefront holds the i-th best
// (distance, label) entry seen so far, ordered by (distance ascending, label ascending) --
// the canonical ranking of this library (include/mlvdb_hip.h).  All operations must be
// executed by the full wavefront (EXEC all ones).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlvdb {

constexpr int kWave = 64;
constexpr int32_t kNoLabel = 0x7fffffff;  // sorts after every real label

__device__ __forceinline__ bool entry_less(double ad, int32_t al, double bd, int32_t bl) {
    return ad < bd || (ad == bd && al < bl);
}

struct WaveTopK {
    double d;     // this lane's entry
    int32_t l;
    double kth_d;  // wave-uniform copy of entry k-1 (admission threshold)
    int32_t kth_l;

    __device__ __forceinline__ void init() {
        d = __builtin_inf();
        l = kNoLabel;
        kth_d = __builtin_inf();
        kth_l = kNoLabel;
    }

    // Offer one candidate per lane (lanes with want == false offer nothing).  k in 1..64.
    __device__ __forceinline__ void offer(bool want, double cd, int32_t cl, int k, int lane) {
        unsigned long long m = __ballot(want && entry_less(cd, cl, kth_d, kth_l));
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const double vd = __shfl(cd, src);
            const int32_t vl = __shfl(cl, src);
            if (!entry_less(vd, vl, kth_d, kth_l)) continue;  // threshold moved since the ballot
            const int pos = __popcll(__ballot(entry_less(d, l, vd, vl)));
            const double up_d = __shfl_up(d, 1);
            const int32_t up_l = __shfl_up(l, 1);
            if (lane > pos) {
                d = up_d;
                l = up_l;
            } else if (lane == pos) {
                d = vd;
                l = vl;
            }
            kth_d = __shfl(d, k - 1);
            kth_l = __shfl(l, k - 1);
        }
    }
};

// 16-byte record used for per-block partial results
struct __attribute__((aligned(16))) TopEntry {
    double d;
    int32_t l;
    int32_t pad;
};

}  // namespace mlvdb
