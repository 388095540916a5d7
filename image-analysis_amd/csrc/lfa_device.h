// lfa_device.h -- small device-side helpers (wave64 reductions, vector loads, gray conversion)
#ifndef LFA_DEVICE_H_
#define LFA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lfa.h"

namespace lfa {

// ---- wave64 reductions (every lane ends with the result) ------------------------------------------------------------
// 32-bit reductions run on the DPP data path (row_shr 1/2/4/8 inside each row of 16 lanes, row_bcast15 / row_bcast31
// across the rows, result in lane 63, v_readlane to everybody): ~15 instructions instead of six ds_bpermute round trips
// (~330 cycles).  ALL 64 lanes must be active at the call.
template <class Op>
__device__ __forceinline__ int wave_reduce_dpp(int v, int identity, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline int wave_min_i32(int v) { return wave_reduce_dpp(v, 0x7fffffff, [](int a, int b) { return min(a, b); }); }
__device__ inline int wave_max_i32(int v) { return wave_reduce_dpp(v, (int)0x80000000, [](int a, int b) { return max(a, b); }); }
__device__ inline uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)wave_reduce_dpp((int)v, 0, [](int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }); }
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

// ---- reductions inside groups of G = 8 / 16 / 32 / 64 consecutive lanes (every lane ends with its group's result) -----------
// On the DPP data path while the partner lies in the lane's own row of 16: quad_perm for lane ^ 1 and lane ^ 2, then
// row_half_mirror (lane i <-> 7 - i of its eight) and row_mirror (i <-> 15 - i of its sixteen) -- for an operation that is
// commutative and associative BIT FOR BIT (integer min, float max / min of NaN-free values, and a float add of two values
// each lane pair holds symmetrically) a mirrored partner serves as well as lane ^ 4 / lane ^ 8, because after the steps
// before it every lane of the partner's quad / half-row holds the same partial result.  Rows are joined with ds_bpermute
// (__shfl_xor 16 / 32).  One v_mov_dpp per 32 bits and step instead of a ds_bpermute round trip (~130 clocks) each.
// A lane whose partner is switched off (another group of the wave that has left) keeps its own value.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL>((int)(uint32_t)b), hi = dpp_i32<CTRL>((int)(uint32_t)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int G>
__device__ __forceinline__ int group_min_i32(int v) {
    v = min(v, dpp_i32<kDppXor1>(v));
    v = min(v, dpp_i32<kDppXor2>(v));
    v = min(v, dpp_i32<kDppHalfMirror>(v));
    if (G >= 16) v = min(v, dpp_i32<kDppMirror>(v));
    if (G >= 32) v = min(v, __shfl_xor(v, 16));
    if (G >= 64) v = min(v, __shfl_xor(v, 32));
    return v;
}
// largest (want_min: smallest) of the group's values; no NaN among them
template <int G>
__device__ __forceinline__ double group_ext_f64(double v, bool want_min) {
    auto pick = [want_min](double a, double b) { return (want_min ? (b < a) : (b > a)) ? b : a; };
    v = pick(v, dpp_f64<kDppXor1>(v));
    v = pick(v, dpp_f64<kDppXor2>(v));
    v = pick(v, dpp_f64<kDppHalfMirror>(v));
    if (G >= 16) v = pick(v, dpp_f64<kDppMirror>(v));
    if (G >= 32) v = pick(v, __shfl_xor(v, 16));
    if (G >= 64) v = pick(v, __shfl_xor(v, 32));
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

// ---- vector memory operations issued by hand (prefetch loops)
// gfx950 counts loads and stores of a wave in ONE in-order counter (vmcnt).  A loop that requests the next group's bytes
// before it works on this group's, and stores a result per round, needs "wait until the prefetched bytes are in" =
// "wait until at most <stores issued after them> operations are outstanding".  The compiler only proves that count in
// straight-line code and waits for everything (the store's round trip included) at a loop boundary, so these loops
// issue their loads, their store and the wait themselves and keep the number of operations per round a constant.
typedef uint32_t lfa_u32x4 __attribute__((ext_vector_type(4)));
struct Raw48 { lfa_u32x4 a, b, c; };
__device__ __forceinline__ void vm_request48(const uint8_t* p, Raw48& r) {                    // 3 operations
    asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c) : "v"(p));
}
// 32 bytes from a 16-byte aligned address (2 operations), and the wait that releases them
__device__ __forceinline__ void vm_request32(const void* p, lfa_u32x4& a, lfa_u32x4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(a), "=&v"(b) : "v"(p));
}
template <int LEFT>
__device__ __forceinline__ void vm_wait32(lfa_u32x4& a, lfa_u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LEFT) : "memory");
    asm volatile("" : "+v"(a), "+v"(b));
}
// three 16-byte pieces at three addresses (a wave's fully coalesced loads of a 3 KiB round), read-once hint
__device__ __forceinline__ void vm_request3nt(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, Raw48& r) {   // 3 operations
    asm volatile("global_load_dwordx4 %0, %3, off nt\n\t"
                 "global_load_dwordx4 %1, %4, off nt\n\t"
                 "global_load_dwordx4 %2, %5, off nt"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c) : "v"(p0), "v"(p1), "v"(p2));
}
__device__ __forceinline__ void vm_store8(void* p, uint32_t lo, uint32_t hi) {                // 1 operation
    const uint2 v = make_uint2(lo, hi);
    asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vm_store2(void* p, uint32_t v) {                              // 1 operation (low 16 bits of v)
    // (no `nt`: the bit plane is read back by k_mask within the call; with the read-once hint k_hist_rgb took 0.479 ms per 256 images
    //  instead of 0.462 -- its time without any store -- and k_mask 0.056 instead of 0.046; -DLFA_STORE2_NT restores the hint)
#ifdef LFA_STORE2_NT
    asm volatile("global_store_short %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_short %0, %1, off" : : "v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void vm_store16(void* p, const lfa_u32x4& v) {                     // 1 operation
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
template <int LEFT>   // wait until at most LEFT vector memory operations of this wave are outstanding, then release r
__device__ __forceinline__ void vm_wait48(Raw48& r) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LEFT) : "memory");
    asm volatile("" : "+v"(r.a), "+v"(r.b), "+v"(r.c));
}
__device__ __forceinline__ void unpack48(const Raw48& r, uint32_t (&w)[12]) {
    w[0] = r.a.x; w[1] = r.a.y; w[2] = r.a.z; w[3] = r.a.w;
    w[4] = r.b.x; w[5] = r.b.y; w[6] = r.b.z; w[7] = r.b.w;
    w[8] = r.c.x; w[9] = r.c.y; w[10] = r.c.z; w[11] = r.c.w;
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

// (rounding order as a run-time value: the few places that evaluate a handful of grays per image)
__device__ inline double gray_u8_rt(int variant, uint32_t p0, uint32_t p1, uint32_t p2) {
    switch (variant) {
        case 0: return gray_u8<0>(p0, p1, p2);
        case 1: return gray_u8<1>(p0, p1, p2);
        case 2: return gray_u8<2>(p0, p1, p2);
        case 3: return gray_u8<3>(p0, p1, p2);
        case 4: return gray_u8<4>(p0, p1, p2);
        default: return gray_u8<5>(p0, p1, p2);
    }
}

// the same dot product on already scaled values, for the pixel types that have no integer form: uint16 colour
// (x_k = fl(pixel_k * fl(1/65535))), float64 colour (x_k = pixel_k: img_as_float leaves floats alone), float64 2-D (gray = pixel)
template <int V>
__device__ inline double gray_mix(double x0, double x1, double x2) {
    const double c0 = 0.2125, c1 = 0.7154, c2 = 0.0721;
    if (V == LFA_GRAY_FMA_102) return fma(x2, c2, fma(x0, c0, x1 * c1));
    if (V == LFA_GRAY_SEQ) return (x0 * c0 + x1 * c1) + x2 * c2;
    if (V == LFA_GRAY_FMA_012) return fma(x2, c2, fma(x1, c1, x0 * c0));
    if (V == LFA_GRAY_FMA_210) return fma(x0, c0, fma(x1, c1, x2 * c2));
    if (V == LFA_GRAY_PAIR) return x0 * c0 + (x1 * c1 + x2 * c2);
    return fma(x1, c1, x0 * c0) + x2 * c2;
}
// gray of pixel px of an image of pixel type `dtype` with C channels (the scalar paths of the colour kernels)
template <int V>
__device__ inline double gray_at(const uint8_t* base, int64_t px, int dtype, int C) {
    if (dtype == LFA_U16) {
        const uint16_t* b = (const uint16_t*)base;
        const double inv = 1.0 / 65535.0;
        return gray_mix<V>((double)b[3 * px] * inv, (double)b[3 * px + 1] * inv, (double)b[3 * px + 2] * inv);
    }
    if (dtype == LFA_F64) {
        const double* b = (const double*)base;
        return C == 1 ? b[px] : gray_mix<V>(b[3 * px], b[3 * px + 1], b[3 * px + 2]);
    }
    return gray_u8<V>(base[3 * px], base[3 * px + 1], base[3 * px + 2]);
}

// ---- integer gray G = 2125*b0 + 7154*b1 + 721*b2 of pixel k (0..15) of a 48-byte group -----------------------------------
// (k_threshold.hip explains why compares on G are exact).  A pixel that straddles two dwords is brought together with
// v_alignbyte; then two v_dot4_u32_u8: weights split as hi*256 + lo, placed where the pixel's bytes sit, the fourth byte weighs 0.
__device__ inline uint32_t gray_G_from_dword(uint32_t px) {
    const uint32_t hi = __builtin_amdgcn_udot4(px, 0x00021B08u, 0u, false);       // 8, 27, 2
    return __builtin_amdgcn_udot4(px, 0x00D1F24Du, hi << 8, false);               // 77, 242, 209
}
__device__ inline uint32_t gray_G16(const uint32_t (&w)[12], int k) {             // k: compile-time constant after unrolling
    const int i = (3 * k) >> 2, s = (3 * k) & 3;
    if (s == 1) {                                                                 // bytes 1-3 of the dword: the weights move instead
        const uint32_t hi = __builtin_amdgcn_udot4(w[i], 0x021B0800u, 0u, false);
        return __builtin_amdgcn_udot4(w[i], 0xD1F24D00u, hi << 8, false);
    }
    const uint32_t px = (s == 0) ? w[i] : __builtin_amdgcn_alignbyte(w[i + 1], w[i], s);
    return gray_G_from_dword(px);
}

// |g - t| <= 2^-40 * t, t finite
__device__ inline bool near_rel(double g, double t) {
    return (t < 1.0e300) && (fabs(g - t) <= t * 9.094947017729282e-13);
}

}  // namespace lfa
#endif
