// kernels_common.h -- device helpers shared by the kernel translation units (kernels.hip, predict.hip, categorical.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace gbrl {
namespace kern {
namespace {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t float_to_key(float x) {
    // order-preserving map float -> uint32; -0.0 and +0.0 map to the same key (they compare equal as floats)
    uint32_t u = __float_as_uint(x);
    if ((u << 1) == 0) u = 0;
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;   // NaN: below every threshold, like the float comparison `x > t` (false)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

inline int grid_for(size_t n, int bs, int cap) {
    size_t b = (n + bs - 1) / bs;
    if (b < 1) b = 1;
    return static_cast<int>(b > static_cast<size_t>(cap) ? cap : b);
}

}  // namespace
}  // namespace kern
}  // namespace gbrl
