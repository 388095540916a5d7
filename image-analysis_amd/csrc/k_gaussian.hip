// k_gaussian.hip -- stage S10: separable Gaussian smoothing, LDS-tiled  (gfx950)
//
// Reference: skimage.filters.gaussian (skimage/_shared/filters.py:35-166, only caller LFA-Image_Analysis/detection.py:166)
//   -> img_as_float (uint8: x * fl(1/255), uint16: x * fl(1/65535); skimage/util/dtype.py:328)
//   -> scipy.ndimage.gaussian_filter(mode='nearest', truncate=4.0): correlate1d along axis 0, then along axis 1 of the
//      result; colour images get sigma 0 on the channel axis, i.e. each channel is filtered on its own.
// correlate1d's symmetric-kernel loop (scipy/ndimage/src/ni_filters.c, not vendored as source in the reference tree)
// is restated as: tmp = x[0]*w[R]; for k = R..1: tmp += (x[-k] + x[+k]) * w[R-k]  -- no fused multiply-add.
//
// One block stages 8 rows x kIW element columns (512 for colour, 256 for grey images), one column per thread, and outputs
// the inner kIW - 2*R*C of them (a row of a colour image is W*3 elements with horizontal taps 3 apart).  Vertical pass: every thread owns one element column, walks down the tile's rows (+R halo rows above
// and below) reading the image coalesced, and keeps the 2R+1 most recent float64 values in registers (sliding
// window), so each input byte is read once and no input staging is needed; results go to LDS (float64).  Horizontal
// pass: every thread owns one output column and reads its 2R+1 taps from LDS (stride-1 across lanes: conflict-free),
// writing whole rows (512 B per wave and row).  Where the time goes (PMC + ablations): the float64
// output stream (8 bytes out per byte in) and 30 float64 operations per output that the reference's rounding order does
// not let fuse; the two overlap only partly.
#include "lfa_internal.h"
#include "lfa_device.h"

namespace lfa {

struct GaussW { double w[2 * kGaussMaxRadius + 1]; int radius; };

// tile: kTH rows x kIW staged element columns, one column per thread (output columns = kIW - 2*R*C).  C (1 or 3) is a
// template parameter so that the horizontal taps are immediate offsets of the LDS reads.
template <typename PIX, typename OUT, int C, int R, int kTH, int kIW>
__global__ __launch_bounds__(kIW) void k_gaussian(Job j, GaussW gw, OUT* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int hal = R * C;                   // horizontal halo in elements
    constexpr int kTE = kIW - 2 * hal;           // output columns of the tile
    static_assert(kTE > 0, "tile narrower than the halo");
    double* vres = (double*)smem;                // [kTH][kIW]
    const int img = blockIdx.z;
    const PIX* base = (const PIX*)(j.images + (int64_t)img * j.img_bytes);
    const int E = j.W * C;
    const int e0 = blockIdx.x * kTE, r0 = blockIdx.y * kTH;
    const double scale = (sizeof(PIX) == 1) ? (1.0 / 255.0) : (1.0 / 65535.0);
    const int t = threadIdx.x;
    double w[R + 1];
#pragma unroll
    for (int k = 0; k <= R; ++k) w[k] = gw.w[k];                 // symmetric: w[k] == w[2R-k]

    // ---- vertical pass (axis 0): thread t owns staged column t
    {
        const uint32_t eh = (uint32_t)(e0 + t);                   // element + hal: never negative
        int px = (int)(eh / (uint32_t)C) - R;                     // 'nearest' border, per channel
        const int ch = (int)(eh % (uint32_t)C);
        px = min(max(px, 0), j.W - 1);
        const PIX* col = base + px * C + ch;
        // all kTH + 2R rows of this column are requested before the first is used (one memory latency per tile, not one
        // per row); the tile height is a compile-time constant, so `raw` and the window stay in registers
        PIX raw[kTH + 2 * R];
#pragma unroll
        for (int k = 0; k < kTH + 2 * R; ++k) {
            const int r = min(max(r0 - R + k, 0), j.H - 1);
            raw[k] = col[(int64_t)r * E];
        }
        double win[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) win[k + 1] = (double)raw[k] * scale;          // rows r0-R .. r0+R-1
#pragma unroll
        for (int rr = 0; rr < kTH; ++rr) {
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) win[k] = win[k + 1];
            win[2 * R] = (double)raw[rr + 2 * R] * scale;
            double tmp = win[R] * w[R];
#pragma unroll
            for (int q = R; q >= 1; --q) tmp = tmp + (win[R - q] + win[R + q]) * w[R - q];
            vres[rr * kIW + t] = tmp;
        }
    }
    __syncthreads();
    // ---- horizontal pass (axis 1): thread t owns output column t of the tile.  The float64 result is written once and
    // never read again by this kernel: nontemporal stores (measured +3..5 % on this write-dominated stream).
    const int e = e0 + t;
    if (t < kTE && e < E) {
        const double* rowp = vres + hal + t;
        OUT* o = out + ((int64_t)img * j.H + r0) * E + e;
#pragma unroll
        for (int rr = 0; rr < kTH; ++rr) {
            if (r0 + rr >= j.H) break;
            const double* row = rowp + rr * kIW;
            double tmp = row[0] * w[R];
#pragma unroll
            for (int q = R; q >= 1; --q) tmp = tmp + (row[-q * C] + row[q * C]) * w[R - q];
            __builtin_nontemporal_store((OUT)tmp, o + (int64_t)rr * E);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// float32 output (no counterpart in the reference, whose result is float64): float32 arithmetic end to end -- the caller
// asked for 24 bits -- with the 1/255 (1/65535) scale folded into the vertical taps and fused multiply-adds; results are
// within 1e-6 (relative to the [0, 1] value range) of the float64 path.  Four adjacent element columns per thread: one
// dword (uint8) or two (uint16) per row and thread on the way in, float4 rows in LDS, one 16-byte store per row and thread
// on the way out.  Rows whose element count is not a multiple of 4 take the float64 kernel above with a float store.
// ---------------------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename PIX>
__device__ __forceinline__ f4 load4_fast(const PIX* p) {          // four adjacent elements, naturally aligned
    if (sizeof(PIX) == 1) {
        const uint32_t x = *(const uint32_t*)p;
        return f4{(float)(x & 0xffu), (float)((x >> 8) & 0xffu), (float)((x >> 16) & 0xffu), (float)(x >> 24)};
    }
    const uint2 x = *(const uint2*)p;
    return f4{(float)(x.x & 0xffffu), (float)(x.x >> 16), (float)(x.y & 0xffffu), (float)(x.y >> 16)};
}

template <typename PIX, int C, int R, int kTH>
__global__ __launch_bounds__(256) void k_gaussian_f32(Job j, GaussW gw, float* out) {
    constexpr int kIW = 1024;                                    // staged element columns: 256 threads x 4
    constexpr int hal = R * C;
    constexpr int kTE = kIW - 2 * hal;                           // output columns of the tile (a multiple of 4 when hal is even)
    constexpr int NV = (4 + 2 * hal + 3) / 4;                    // float4s a thread reads per row in the horizontal pass
    __shared__ f4 vres[kTH][kIW / 4];
    const int img = blockIdx.z;
    const PIX* base = (const PIX*)(j.images + (int64_t)img * j.img_bytes);
    const int E = j.W * C;
    const int e0 = blockIdx.x * kTE, r0 = blockIdx.y * kTH;
    const float scale = (sizeof(PIX) == 1) ? (float)(1.0 / 255.0) : (float)(1.0 / 65535.0);
    const int t = threadIdx.x;
    float w[R + 1], ws[R + 1];
#pragma unroll
    for (int k = 0; k <= R; ++k) { w[k] = (float)gw.w[k]; ws[k] = (float)(gw.w[k] * (double)scale); }
    {
        const int p0 = e0 - hal + 4 * t;                          // first of this thread's four source elements
        const bool fast = p0 >= 0 && p0 + 3 < E && ((p0 * (int)sizeof(PIX)) & (4 * (int)sizeof(PIX) - 1)) == 0;
        f4 raw[kTH + 2 * R];
        if (fast) {                                               // every row of the tile in flight before the first is used
#pragma unroll
            for (int k = 0; k < kTH + 2 * R; ++k) {
                const int r = min(max(r0 - R + k, 0), j.H - 1);
                raw[k] = load4_fast(base + (int64_t)r * E + p0);
            }
        } else {                                                  // tiles at the left / right border: 'nearest', per channel
            int off[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int p = p0 + m;
                const int q = p < 0 ? -((-p + C - 1) / C) : p / C;    // floor(p / C)
                off[m] = min(max(q, 0), j.W - 1) * C + (p - q * C);
            }
#pragma unroll
            for (int k = 0; k < kTH + 2 * R; ++k) {
                const PIX* rowp = base + (int64_t)min(max(r0 - R + k, 0), j.H - 1) * E;
                raw[k] = f4{(float)rowp[off[0]], (float)rowp[off[1]], (float)rowp[off[2]], (float)rowp[off[3]]};
            }
        }
#pragma unroll
        for (int rr = 0; rr < kTH; ++rr) {
            f4 tmp = raw[rr + R] * ws[R];
#pragma unroll
            for (int q = R; q >= 1; --q) tmp = __builtin_elementwise_fma(raw[rr + R - q] + raw[rr + R + q], (f4)(ws[R - q]), tmp);
            vres[rr][t] = tmp;
        }
    }
    __syncthreads();
    const int e = e0 + 4 * t;
    if (4 * t < kTE && e < E) {
        float* o = out + ((int64_t)img * j.H + r0) * E + e;
#pragma unroll
        for (int rr = 0; rr < kTH; ++rr) {
            if (r0 + rr >= j.H) break;
            float a[4 * NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f4 x = vres[rr][t + v];
                a[4 * v] = x[0]; a[4 * v + 1] = x[1]; a[4 * v + 2] = x[2]; a[4 * v + 3] = x[3];
            }
            f4 res;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float tmp = a[m + hal] * w[R];
#pragma unroll
                for (int q = R; q >= 1; --q) tmp = __builtin_fmaf(a[m + hal - q * C] + a[m + hal + q * C], w[R - q], tmp);
                res[m] = tmp;
            }
            __builtin_nontemporal_store(res, (f4*)(o + (int64_t)rr * E));
        }
    }
}

template <typename PIX, int C, int R>
static hipError_t launch_f32(const Job& j, const GaussW& gw, float* out, hipStream_t s) {
    constexpr int kTH = 8;
    constexpr int te = 1024 - 2 * R * C;
    const int E = j.W * C;
    const dim3 grid((E + te - 1) / te, (j.H + kTH - 1) / kTH, j.n);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    LFA_LAUNCH("k_gaussian_f32", (k_gaussian_f32<PIX, C, R, kTH>), grid, dim3(256), 0, s, j, gw, out);
    return hipSuccess;
}

template <typename PIX, typename OUT, int C, int R, int kTH, int kIW>
static hipError_t launch_t(const Job& j, const GaussW& gw, OUT* out, hipStream_t s) {
    const int E = j.W * C;
    constexpr int te = kIW - 2 * R * C;
    const size_t lds = (size_t)kTH * kIW * sizeof(double);
    const dim3 grid((E + te - 1) / te, (j.H + kTH - 1) / kTH, j.n);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, C, R, kTH, kIW>), grid, dim3(kIW), lds, s, j, gw, out);
    return hipSuccess;
}

// tile shapes from the sweep in DESIGN.md section 4 (sigma 1): 8 x 512 for colour rows (halo 24 of 512 columns),
// 8 x 256 for grey rows
template <typename PIX, typename OUT, int C, int R>
static hipError_t launch_shape(const Job& j, const GaussW& gw, OUT* out, hipStream_t s) {
    // float output of rows with a multiple of 4 elements (and a halo that keeps the tiles' output width a multiple of 4)
    if constexpr (sizeof(OUT) == 4 && ((2 * R * C) % 4) == 0) {
        if (((j.W * C) & 3) == 0 && j.aligned16 && (((uintptr_t)out) & 15) == 0) return launch_f32<PIX, C, R>(j, gw, (float*)out, s);
    }
#ifndef LFA_GAUSS_TH_RGB
#define LFA_GAUSS_TH_RGB 8
#endif
    if (C == 3) return launch_t<PIX, OUT, C, R, LFA_GAUSS_TH_RGB, 512>(j, gw, out, s);
    return launch_t<PIX, OUT, C, R, 8, 256>(j, gw, out, s);
}

template <typename PIX, typename OUT, int C>
static hipError_t launch_r(const Job& j, const GaussW& gw, OUT* out, hipStream_t s) {
    switch (gw.radius) {
        case 1: return launch_shape<PIX, OUT, C, 1>(j, gw, out, s);
        case 2: return launch_shape<PIX, OUT, C, 2>(j, gw, out, s);
        case 3: return launch_shape<PIX, OUT, C, 3>(j, gw, out, s);
        case 4: return launch_shape<PIX, OUT, C, 4>(j, gw, out, s);
        case 5: return launch_shape<PIX, OUT, C, 5>(j, gw, out, s);
        case 6: return launch_shape<PIX, OUT, C, 6>(j, gw, out, s);
        case 7: return launch_shape<PIX, OUT, C, 7>(j, gw, out, s);
        default: return launch_shape<PIX, OUT, C, 8>(j, gw, out, s);
    }
}

template <typename OUT>
static hipError_t launch_o(const Job& j, const GaussW& gw, OUT* out, hipStream_t s) {
    if (j.dtype == LFA_U8) return j.C == 3 ? launch_r<uint8_t, OUT, 3>(j, gw, out, s) : launch_r<uint8_t, OUT, 1>(j, gw, out, s);
    return launch_r<uint16_t, OUT, 1>(j, gw, out, s);
}

hipError_t launch_gaussian_w(const Job& j, const double* w, int radius, void* out, int out_f32, hipStream_t s) {
    if (radius < 1 || radius > kGaussMaxRadius) return hipErrorInvalidValue;
    GaussW gw;
    gw.radius = radius;
    for (int k = 0; k < 2 * radius + 1; ++k) gw.w[k] = w[k];
    if (j.C != 1 && !(j.C == 3 && j.dtype == LFA_U8)) return hipErrorInvalidValue;
    const hipError_t e = out_f32 ? launch_o<float>(j, gw, (float*)out, s) : launch_o<double>(j, gw, (double*)out, s);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace lfa
