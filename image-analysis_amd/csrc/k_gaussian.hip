// k_gaussian.hip -- stage S10: separable Gaussian smoothing, LDS-tiled  (gfx950)
//
// Reference: skimage.filters.gaussian (skimage/_shared/filters.py:35-166, only caller LFA-Image_Analysis/detection.py:166)
//   -> img_as_float (uint8: x * fl(1/255), uint16: x * fl(1/65535); skimage/util/dtype.py:328)
//   -> scipy.ndimage.gaussian_filter(mode='nearest', truncate=4.0): correlate1d along axis 0, then along axis 1 of the
//      result; colour images get sigma 0 on the channel axis, i.e. each channel is filtered on its own.
// correlate1d's symmetric-kernel loop (scipy/ndimage/src/ni_filters.c, not vendored as source in the reference tree)
// is restated as: tmp = x[0]*w[R]; for k = R..1: tmp += (x[-k] + x[+k]) * w[R-k]  -- no fused multiply-add.
// One block computes a 32-row x 128-element tile: input tile (+halo) -> LDS, vertical pass -> LDS (float64),
// horizontal pass -> global.  A row of a colour image is treated as W*3 elements with horizontal taps 3 apart.
#include "lfa_internal.h"
#include "lfa_device.h"

namespace lfa {

constexpr int kGR = kGaussMaxRadius;  // max radius (LDS tile stays under 64 KiB)
constexpr int kTH = 32, kTE = 128;    // tile: rows x elements

struct GaussW { double w[2 * kGR + 1]; int radius; };

template <typename PIX, typename OUT>
__global__ __launch_bounds__(256) void k_gaussian(Job j, GaussW gw, OUT* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int R = gw.radius, C = j.C;
    const int hal = R * C;                       // horizontal halo in elements
    const int IW = kTE + 2 * hal;                // staged width
    const int IH = kTH + 2 * R;
    double* vres = (double*)smem;                // [kTH][IW]
    PIX* tin = (PIX*)(smem + (size_t)kTH * IW * sizeof(double));   // [IH][IW]
    const int img = blockIdx.z;
    const PIX* base = (const PIX*)(j.images + (int64_t)img * j.img_bytes);
    const int E = j.W * C;
    const int e0 = blockIdx.x * kTE, r0 = blockIdx.y * kTH;
    const double scale = (sizeof(PIX) == 1) ? (1.0 / 255.0) : (1.0 / 65535.0);
    const int t = threadIdx.x;
    // stage input with 'nearest' borders
    for (int k = t; k < IH * IW; k += 256) {
        const int rr = k / IW, ee = k % IW;
        int r = r0 + rr - R; r = r < 0 ? 0 : (r >= j.H ? j.H - 1 : r);
        const int e = e0 + ee - hal;              // may be outside [0,E)
        int px = (e >= 0 ? e / C : -((-e + C - 1) / C));
        const int ch = e - px * C;
        px = px < 0 ? 0 : (px >= j.W ? j.W - 1 : px);
        tin[k] = base[(int64_t)r * E + px * C + ch];
    }
    __syncthreads();
    // vertical pass (axis 0)
    for (int k = t; k < kTH * IW; k += 256) {
        const int rr = k / IW, ee = k % IW;
        const PIX* col = tin + (rr + R) * IW + ee;
        double tmp = ((double)col[0] * scale) * gw.w[R];
        for (int q = R; q >= 1; --q)
            tmp = tmp + (((double)col[-q * IW] * scale) + ((double)col[q * IW] * scale)) * gw.w[R - q];
        vres[k] = tmp;
    }
    __syncthreads();
    // horizontal pass (axis 1)
    for (int k = t; k < kTH * kTE; k += 256) {
        const int rr = k / kTE, ee = k % kTE;
        const int r = r0 + rr, e = e0 + ee;
        if (r >= j.H || e >= E) continue;
        const double* row = vres + rr * IW + ee + hal;
        double tmp = row[0] * gw.w[R];
        for (int q = R; q >= 1; --q) tmp = tmp + (row[-q * C] + row[q * C]) * gw.w[R - q];
        out[((int64_t)img * j.H + r) * E + e] = (OUT)tmp;
    }
}

hipError_t launch_gaussian_w(const Job& j, const double* w, int radius, void* out, int out_f32, hipStream_t s) {
    GaussW gw;
    gw.radius = radius;
    for (int k = 0; k < 2 * radius + 1; ++k) gw.w[k] = w[k];
    const int E = j.W * j.C;
    const int IW = kTE + 2 * radius * j.C;
    const size_t px = (j.dtype == LFA_U8) ? 1 : 2;
    const size_t lds = (size_t)kTH * IW * sizeof(double) + (size_t)(kTH + 2 * radius) * IW * px;
    const dim3 grid((E + kTE - 1) / kTE, (j.H + kTH - 1) / kTH, j.n);
    if (j.dtype == LFA_U8) {
        if (out_f32) LFA_LAUNCH("k_gaussian", (k_gaussian<uint8_t, float>), grid, dim3(256), lds, s, j, gw, (float*)out);
        else         LFA_LAUNCH("k_gaussian", (k_gaussian<uint8_t, double>), grid, dim3(256), lds, s, j, gw, (double*)out);
    } else {
        if (out_f32) LFA_LAUNCH("k_gaussian", (k_gaussian<uint16_t, float>), grid, dim3(256), lds, s, j, gw, (float*)out);
        else         LFA_LAUNCH("k_gaussian", (k_gaussian<uint16_t, double>), grid, dim3(256), lds, s, j, gw, (double*)out);
    }
    return hipGetLastError();
}

}  // namespace lfa
