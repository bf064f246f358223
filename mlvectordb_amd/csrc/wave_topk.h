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

// Cross-lane moves go through __shfl / __shfl_up (ds_bpermute).  Two faster-looking forms were tried
// and are wrong on gfx950 / ROCm 7.2 in this code: DPP wave_shr:1 and v_readlane-based reads both
// made list entries vanish (the exact-scan parity tests catch it), so they stay out.
__device__ __forceinline__ int32_t lane_read(int32_t v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double lane_read(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int32_t lane_shr1(int32_t v) { return __shfl_up(v, 1); }
__device__ __forceinline__ double lane_shr1(double v) { return __shfl_up(v, 1); }

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
            const double vd = lane_read(cd, src);
            const int32_t vl = lane_read(cl, src);
            if (!entry_less(vd, vl, kth_d, kth_l)) continue;  // threshold moved since the ballot
            const int pos = __popcll(__ballot(entry_less(d, l, vd, vl)));
            const double up_d = lane_shr1(d);
            const int32_t up_l = lane_shr1(l);
            if (lane > pos) {
                d = up_d;
                l = up_l;
            } else if (lane == pos) {
                d = vd;
                l = vl;
            }
            kth_d = lane_read(d, k - 1);
            kth_l = lane_read(l, k - 1);
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
