// k_gaussian.hip -- stage S10: separable Gaussian smoothing, LDS-tiled  (gfx950)
//
// Reference: skimage.filters.gaussian (skimage/_shared/filters.py:35-166, only caller LFA-Image_Analysis/detection.py:166)
//   -> img_as_float (uint8: x * fl(1/255), uint16: x * fl(1/65535); skimage/util/dtype.py:328)
//   -> scipy.ndimage.gaussian_filter(mode='nearest', truncate=4.0): correlate1d along axis 0, then along axis 1 of the
//      result; colour images get sigma 0 on the channel axis, i.e. each channel is filtered on its own.
// correlate1d's symmetric-kernel loop (scipy/ndimage/src/ni_filters.c, not vendored as source in the reference tree)
// is restated as: tmp = x[0]*w[R]; for k = R..1: tmp += (x[-k] + x[+k]) * w[R-k]  -- no fused multiply-add.
//
// One 256-thread block stages 16 rows x 256 element columns and outputs the inner 256 - 2*R*C of them (a row of a colour image is W*3 elements with horizontal
// taps 3 apart).  Vertical pass: every thread owns one element column, walks down the tile's rows (+R halo rows above
// and below) reading the image coalesced, and keeps the 2R+1 most recent float64 values in registers (sliding
// window), so each input byte is read once and no input staging is needed; results go to LDS (float64).  Horizontal
// pass: every thread owns one output column and reads its 2R+1 taps from LDS (stride-1 across lanes: conflict-free),
// writing whole rows (2 KiB per wave).  The kernel is bound by the float64 output stream (8 bytes out per byte in).
#include "lfa_internal.h"
#include "lfa_device.h"

namespace lfa {

constexpr int kTH = 16, kIW = 256;    // tile: rows x staged element columns (output columns = kIW - 2*R*C)

struct GaussW { double w[2 * kGaussMaxRadius + 1]; int radius; };

template <typename PIX, typename OUT, int R>
__global__ __launch_bounds__(256) void k_gaussian(Job j, GaussW gw, OUT* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = j.C;
    const int hal = R * C;                       // horizontal halo in elements
    const int IW = kIW;                          // columns the vertical pass produces: exactly one per thread
    const int kTE = kIW - 2 * hal;               // output columns of the tile
    double* vres = (double*)smem;                // [kTH][IW]
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
        const int ee = t;
        const int e = e0 + ee - hal;                              // may be outside [0,E): 'nearest' border per channel
        int px = (e >= 0 ? e / C : -((-e + C - 1) / C));
        const int ch = e - px * C;
        px = px < 0 ? 0 : (px >= j.W ? j.W - 1 : px);
        const PIX* col = base + px * C + ch;
        // all kTH + 2R rows of this column are requested before the first is used (one memory latency per tile, not one
        // per row); the tile height is a compile-time constant, so `raw` and the window stay in registers
        PIX raw[kTH + 2 * R];
#pragma unroll
        for (int k = 0; k < kTH + 2 * R; ++k) {
            int r = r0 - R + k; r = r < 0 ? 0 : (r >= j.H ? j.H - 1 : r);
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
            vres[rr * IW + ee] = tmp;
        }
    }
    __syncthreads();
    // ---- horizontal pass (axis 1)
    const int e = e0 + t;
    if (t < kTE && e < E) {
        const double* rowp = vres + hal + t;
        OUT* o = out + ((int64_t)img * j.H + r0) * E + e;
#pragma unroll
        for (int rr = 0; rr < kTH; ++rr) {
            if (r0 + rr >= j.H) break;
            const double* row = rowp + rr * IW;
            double tmp = row[0] * w[R];
#pragma unroll
            for (int q = R; q >= 1; --q) tmp = tmp + (row[-q * C] + row[q * C]) * w[R - q];
            o[(int64_t)rr * E] = (OUT)tmp;
        }
    }
}

template <typename PIX, typename OUT>
static void launch_r(const Job& j, const GaussW& gw, OUT* out, hipStream_t s) {
    const int E = j.W * j.C;
    const int te = kIW - 2 * gw.radius * j.C;
    const size_t lds = (size_t)kTH * kIW * sizeof(double);
    const dim3 grid((E + te - 1) / te, (j.H + kTH - 1) / kTH, j.n);
    switch (gw.radius) {
        case 1: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 1>), grid, dim3(256), lds, s, j, gw, out); break;
        case 2: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 2>), grid, dim3(256), lds, s, j, gw, out); break;
        case 3: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 3>), grid, dim3(256), lds, s, j, gw, out); break;
        case 4: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 4>), grid, dim3(256), lds, s, j, gw, out); break;
        case 5: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 5>), grid, dim3(256), lds, s, j, gw, out); break;
        case 6: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 6>), grid, dim3(256), lds, s, j, gw, out); break;
        case 7: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 7>), grid, dim3(256), lds, s, j, gw, out); break;
        default: LFA_LAUNCH("k_gaussian", (k_gaussian<PIX, OUT, 8>), grid, dim3(256), lds, s, j, gw, out); break;
    }
}

hipError_t launch_gaussian_w(const Job& j, const double* w, int radius, void* out, int out_f32, hipStream_t s) {
    if (radius < 1 || radius > kGaussMaxRadius) return hipErrorInvalidValue;
    GaussW gw;
    gw.radius = radius;
    for (int k = 0; k < 2 * radius + 1; ++k) gw.w[k] = w[k];
    if (j.dtype == LFA_U8) {
        if (out_f32) launch_r<uint8_t, float>(j, gw, (float*)out, s);
        else         launch_r<uint8_t, double>(j, gw, (double*)out, s);
    } else {
        if (out_f32) launch_r<uint16_t, float>(j, gw, (float*)out, s);
        else         launch_r<uint16_t, double>(j, gw, (double*)out, s);
    }
    return hipGetLastError();
}

}  // namespace lfa
