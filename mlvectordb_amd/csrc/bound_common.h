// Small device helpers shared by the bound-filter kernels (kernels_filter.hip, kernels_refine.hip): directed rounding between
// double and float, and the monotone float <-> uint order keys the radix selections work on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlvdb {

// round a double down to a float that is <= it (finite results for finite inputs beyond the float range: -max / +max)
__device__ __forceinline__ float float_below(double v) {
    float f = (float)v;
    if ((double)f > v) f = __uint_as_float(f > 0.f ? __float_as_uint(f) - 1 : (f < 0.f ? __float_as_uint(f) + 1 : 0x80000001u));
    return f;
}

// round a double up to a float that is >= it
__device__ __forceinline__ float float_above(double v) { return -float_below(-v); }

__device__ __forceinline__ uint32_t float_order_key(float f) {  // monotone float -> uint (larger float, larger key)
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

}  // namespace mlvdb
