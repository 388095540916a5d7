// lfa_device.h -- small device-side helpers (wave64 reductions, vector loads, gray conversion)
#ifndef LFA_DEVICE_H_
#define LFA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lfa.h"

namespace lfa {

// ---- wave64 reductions (every lane ends with the result) ------------------------------------------------------------
__device__ inline int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ inline int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ inline double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// ---- loads -----------------------------------------------------------------------------------------------------------
// 48 bytes (16 colour pixels) from a 16-byte aligned address
__device__ inline void load48(const uint8_t* p, uint32_t (&w)[12]) {
    const uint4* v = (const uint4*)p;
    const uint4 a = v[0], b = v[1], c = v[2];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}

// byte k of a little-endian dword array; k must be a compile-time constant after unrolling
template <int N>
__device__ inline uint32_t byte_of(const uint32_t (&w)[N], int k) {
    return (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
}

// ---- rgb2gray on uint8 (skimage/color/colorconv.py:876-877 after skimage/util/dtype.py:328) ------------------------
// x_k = fl(pixel_k * fl(1/255));  V selects the rounding order of the 3-term dot product (include/lfa.h LFA_GRAY_*).
template <int V>
__device__ inline double gray_u8(uint32_t p0, uint32_t p1, uint32_t p2) {
    const double inv = 1.0 / 255.0;
    const double c0 = 0.2125, c1 = 0.7154, c2 = 0.0721;
    const double x0 = (double)p0 * inv, x1 = (double)p1 * inv, x2 = (double)p2 * inv;
    if (V == LFA_GRAY_FMA_102) return fma(x2, c2, fma(x0, c0, x1 * c1));
    if (V == LFA_GRAY_SEQ) return (x0 * c0 + x1 * c1) + x2 * c2;
    if (V == LFA_GRAY_FMA_012) return fma(x2, c2, fma(x1, c1, x0 * c0));
    if (V == LFA_GRAY_FMA_210) return fma(x0, c0, fma(x1, c1, x2 * c2));
    if (V == LFA_GRAY_PAIR) return x0 * c0 + (x1 * c1 + x2 * c2);
    return fma(x1, c1, x0 * c0) + x2 * c2;
}

// |g - t| <= 2^-40 * t, t finite
__device__ inline bool near_rel(double g, double t) {
    return (t < 1.0e300) && (fabs(g - t) <= t * 9.094947017729282e-13);
}

}  // namespace lfa
#endif
