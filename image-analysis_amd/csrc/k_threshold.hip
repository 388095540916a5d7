// k_threshold.hip -- stages S1-S3: gray / gamma, min-max, histogram, Otsu threshold  (gfx950)
//
// Reference arithmetic (paths under /root/reference/image-analysis/lib/python3.10/site-packages unless noted):
//   rgb2gray         skimage/color/colorconv.py:834-877, skimage/util/dtype.py:320-329
//   adjust_gamma     skimage/exposure/exposure.py:611-681
//   threshold_otsu   skimage/filters/thresholding.py:312-381, skimage/exposure/exposure.py:35-83, :271-305
//   caller           LFA-Image_Analysis/detection.py:32-47
//
// Integer images (2-D uint8 / uint16): gamma is a monotone look-up table, so the histogram is taken over the INPUT
// values and pushed through the table inside the Otsu kernel; the mask is an integer compare `pixel > x_thr`.
// Colour images: gray is float64; the 257 histogram edges (np.linspace in gamma space) are mapped back to gray space
// on the device by bisection over the bit patterns of the doubles, so that per pixel only compares remain.
#include "lfa_internal.h"
#include "lfa_algo.h"
#include "lfa_device.h"

namespace lfa {

// ------------------------------------------------------------------------------------------------------------------
// init
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_init_info(Job j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n) return;
    LfaImageInfo z;
    z.xmin = 0x7fffffff; z.xmax = -1; z.v_thr = 0; z.x_thr = 0x7fffffff;
    z.gmin = __longlong_as_double(0x7ff0000000000000LL); z.gmax = 0.0;
    z.vmin = 0.0; z.vmax = 0.0; z.v_thr_f = 0.0; z.g_thr = __longlong_as_double(0x7ff0000000000000LL);
    z.otsu_bin = -1; z.all_equal = 0; z.ambiguous = 0;
    z.n_runs = 0; z.n_labels = 0; z.label00 = 0; z.n_candidates = 0; z.n_accepted = 0; z.n_strips = 0; z.status = 0;
    j.info[i] = z;
}

// ------------------------------------------------------------------------------------------------------------------
// min / max
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_u16(Job j) {
    const int img = blockIdx.y;
    const uint16_t* base = (const uint16_t*)(j.images + (int64_t)img * j.img_bytes);
    const int64_t npix = (int64_t)j.H * j.W;
    int lo = 0x7fffffff, hi = -1;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if (j.aligned16) {
        const int64_t ng = npix >> 3;
        const uint4* v = (const uint4*)base;
        for (int64_t g = tid; g < ng; g += nth) {
            const uint4 q = v[g];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = (int)(w[k] & 0xffffu), b = (int)(w[k] >> 16);
                lo = min(lo, min(a, b)); hi = max(hi, max(a, b));
            }
        }
        for (int64_t p = (ng << 3) + tid; p < npix; p += nth) { const int a = base[p]; lo = min(lo, a); hi = max(hi, a); }
    } else {
        for (int64_t p = tid; p < npix; p += nth) { const int a = base[p]; lo = min(lo, a); hi = max(hi, a); }
    }
    lo = wave_min_i32(lo); hi = wave_max_i32(hi);
    __shared__ int slo[4], shi[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[wv] = lo; shi[wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { lo = min(lo, slo[k]); hi = max(hi, shi[k]); }
        if (hi >= 0) { atomicMin(&j.info[img].xmin, lo); atomicMax(&j.info[img].xmax, hi); }
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_minmax_rgb(Job j) {
    const int img = blockIdx.y;
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t npix = (int64_t)j.H * j.W;
    double lo = __longlong_as_double(0x7ff0000000000000LL), hi = 0.0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (j.aligned16) {
        const int64_t ng = npix >> 4;                 // groups of 16 pixels = 48 bytes
        for (int64_t g = tid; g < ng; g += nth) {
            uint32_t w[12];
            load48(base + g * 48, w);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double gr = gray_u8<V>(byte_of(w, 3 * k), byte_of(w, 3 * k + 1), byte_of(w, 3 * k + 2));
                lo = fmin(lo, gr); hi = fmax(hi, gr);
            }
        }
        done = ng << 4;
    }
    for (int64_t p = done + tid; p < npix; p += nth) {
        const double gr = gray_u8<V>(base[3 * p], base[3 * p + 1], base[3 * p + 2]);
        lo = fmin(lo, gr); hi = fmax(hi, gr);
    }
    lo = wave_min_f64(lo); hi = wave_max_f64(hi);
    __shared__ double slo[4], shi[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[wv] = lo; shi[wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { lo = fmin(lo, slo[k]); hi = fmax(hi, shi[k]); }
        // non-negative doubles order like their bit patterns
        atomicMin((unsigned long long*)&j.info[img].gmin, (unsigned long long)__double_as_longlong(lo));
        atomicMax((unsigned long long*)&j.info[img].gmax, (unsigned long long)__double_as_longlong(hi));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// colour path: histogram edges in gray space
// ------------------------------------------------------------------------------------------------------------------
__device__ inline double gamma_pow(double g) { return pow(g, 0.4); }

// smallest double g in [gmin, gmax] with pred(g) true, where pred is monotone (false ... false true ... true);
// +inf if pred(gmax) is false.  Bisection on the bit patterns (non-negative doubles are ordered like int64).
template <class Pred>
__device__ inline double first_true(double gmin, double gmax, const Pred& pred) {
    if (!pred(gmax)) return __longlong_as_double(0x7ff0000000000000LL);
    if (pred(gmin)) return gmin;
    long long lo = __double_as_longlong(gmin), hi = __double_as_longlong(gmax);
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (pred(__longlong_as_double(mid))) hi = mid; else lo = mid;
    }
    return __longlong_as_double(hi);
}

__global__ __launch_bounds__(320) void k_edges_rgb(Job j) {
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    const double gmin = inf->gmin, gmax = inf->gmax;
    const double vmin = gamma_pow(gmin), vmax = gamma_pow(gmax);
    const int i = threadIdx.x;
    if (i == 0) { inf->vmin = vmin; inf->vmax = vmax; inf->all_equal = (vmin == vmax) ? 1 : 0; }
    if (i >= kEdges) return;
    double* t = j.edges_g + (int64_t)img * kEdges;
    if (vmin == vmax) { t[i] = (i == 0) ? gmin : __longlong_as_double(0x7ff0000000000000LL); return; }
    const Linspace lin(vmin, vmax, kEdges);
    const double e = lin(i);
    t[i] = first_true(gmin, gmax, [e](double g) { return gamma_pow(g) >= e; });
}

// ------------------------------------------------------------------------------------------------------------------
// histograms
// ------------------------------------------------------------------------------------------------------------------
constexpr int kCopies = 8;      // LDS sub-histograms, lane l uses copy l & 7 (spreads same-bin lanes over banks)

__global__ __launch_bounds__(256) void k_hist_u8(Job j) {
    __shared__ uint32_t sh[256 * kCopies];
    for (int k = threadIdx.x; k < 256 * kCopies; k += 256) sh[k] = 0;
    __syncthreads();
    const int img = blockIdx.y;
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int cp = threadIdx.x & (kCopies - 1);
    int64_t done = 0;
    if (j.aligned16) {
        const int64_t ng = npix >> 4;
        const uint4* v = (const uint4*)base;
        for (int64_t g = tid; g < ng; g += nth) {
            const uint4 q = v[g];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) atomicAdd(&sh[byte_of(w, k) * kCopies + cp], 1u);
        }
        done = ng << 4;
    }
    for (int64_t p = done + tid; p < npix; p += nth) atomicAdd(&sh[base[p] * kCopies + cp], 1u);
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int c = 0; c < kCopies; ++c) tot += sh[threadIdx.x * kCopies + c];
    if (tot) atomicAdd(&j.hist[(int64_t)img * j.hist_stride + threadIdx.x], tot);
}

constexpr int kLdsBinsU16 = 16384;

__global__ __launch_bounds__(1024) void k_hist_u16(Job j) {
    __shared__ uint32_t sh[kLdsBinsU16];
    const int img = blockIdx.y;
    const int xmin = j.info[img].xmin, xmax = j.info[img].xmax;
    if (xmax <= xmin) return;                       // constant image: no histogram needed (all-equal shortcut)
    const int nb = xmax - xmin + 1;
    const int nlds = nb < kLdsBinsU16 ? nb : kLdsBinsU16;
    for (int k = threadIdx.x; k < nlds; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const uint16_t* base = (const uint16_t*)(j.images + (int64_t)img * j.img_bytes);
    uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (j.aligned16) {
        const int64_t ng = npix >> 3;
        const uint4* v = (const uint4*)base;
        for (int64_t g = tid; g < ng; g += nth) {
            const uint4 q = v[g];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = (int)(w[k] & 0xffffu) - xmin, b = (int)(w[k] >> 16) - xmin;
                if (a < kLdsBinsU16) atomicAdd(&sh[a], 1u); else atomicAdd(&gh[a], 1u);
                if (b < kLdsBinsU16) atomicAdd(&sh[b], 1u); else atomicAdd(&gh[b], 1u);
            }
        }
        done = ng << 3;
    }
    for (int64_t p = done + tid; p < npix; p += nth) {
        const int a = (int)base[p] - xmin;
        if (a < kLdsBinsU16) atomicAdd(&sh[a], 1u); else atomicAdd(&gh[a], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nlds; k += blockDim.x) {
        const uint32_t c = sh[k];
        if (c) atomicAdd(&gh[k], c);
    }
}

// zero the bins [0, xmax-xmin] of each image's histogram before k_hist_u16 (only that range is ever touched)
__global__ void k_zero_hist_u16(Job j) {
    const int img = blockIdx.y;
    const int xmin = j.info[img].xmin, xmax = j.info[img].xmax;
    const int nb = xmax - xmin + 1;
    uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) gh[k] = 0;
}

template <int V>
__global__ __launch_bounds__(256) void k_hist_rgb(Job j) {
    __shared__ uint32_t sh[256 * kCopies];
    __shared__ double st[kEdges + 1];
    const int img = blockIdx.y;
    if (j.info[img].all_equal) return;
    for (int k = threadIdx.x; k < 256 * kCopies; k += 256) sh[k] = 0;
    const double* te = j.edges_g + (int64_t)img * kEdges;
    for (int k = threadIdx.x; k < kEdges; k += 256) st[k] = te[k];
    __syncthreads();
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int cp = threadIdx.x & (kCopies - 1);
    uint32_t amb = 0;
    auto add = [&](double g) {
        // largest i in [0,256] with t[i] <= g   (t[0] = gmin <= g always)
        int lo = 0, hi = 256;
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            const bool ge = st[mid] <= g;
            lo = ge ? mid : lo;
            hi = ge ? hi : mid - 1;
        }
        // pixels within 2^-40 (relative) of an interior edge: their bin could depend on the last bits of pow()
        const int a = lo < 1 ? 1 : (lo > 255 ? 255 : lo);
        const int b = lo + 1 <= 255 ? lo + 1 : 255;
        amb += near_rel(g, st[a]) || near_rel(g, st[b]);
        const int bin = lo > 255 ? 255 : lo;
        atomicAdd(&sh[bin * kCopies + cp], 1u);
    };
    int64_t done = 0;
    if (j.aligned16) {
        const int64_t ng = npix >> 4;
        for (int64_t g = tid; g < ng; g += nth) {
            uint32_t w[12];
            load48(base + g * 48, w);
#pragma unroll 4
            for (int k = 0; k < 16; ++k)
                add(gray_u8<V>(byte_of(w, 3 * k), byte_of(w, 3 * k + 1), byte_of(w, 3 * k + 2)));
        }
        done = ng << 4;
    }
    for (int64_t p = done + tid; p < npix; p += nth) add(gray_u8<V>(base[3 * p], base[3 * p + 1], base[3 * p + 2]));
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int c = 0; c < kCopies; ++c) tot += sh[threadIdx.x * kCopies + c];
    if (tot) atomicAdd(&j.hist[(int64_t)img * j.hist_stride + threadIdx.x], tot);
    amb = wave_sum_u32(amb);
    if ((threadIdx.x & 63) == 0 && amb) atomicAdd(&j.info[img].ambiguous, amb);
}

// ------------------------------------------------------------------------------------------------------------------
// Otsu, integer path.  One 256-thread block per image.  All sums are exact integers (counts < 2^24, so the
// reference's float32 cumsums are exact too); the only floating-point steps are the two divisions, the float32
// product w1*w2 and the final float64 product, done per candidate in the reference's order.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_otsu_int(Job j) {
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    const uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    const int t = threadIdx.x;
    __shared__ int s_lo, s_hi;
    __shared__ unsigned long long s_w[256], s_s[256];
    __shared__ double s_bv[256];
    __shared__ int s_bi[256];
    int xmin, xmax;
    if (j.dtype == LFA_U8) {
        // histogram covers 0..255; find occupied range
        if (t == 0) { s_lo = 0x7fffffff; s_hi = -1; }
        __syncthreads();
        if (gh[t]) { atomicMin(&s_lo, t); atomicMax(&s_hi, t); }
        __syncthreads();
        xmin = s_lo; xmax = s_hi;
        if (t == 0) { inf->xmin = xmin; inf->xmax = xmax; }
    } else {
        xmin = inf->xmin; xmax = inf->xmax;
    }
    const int base_bin = (j.dtype == LFA_U8) ? xmin : 0;        // uint16 histogram is stored relative to xmin
    const int nb = xmax - xmin + 1;
    auto lut = [&](int x) -> int { return (j.dtype == LFA_U8) ? (int)j.lut8[x] : (int)j.lut16[x]; };
    if (nb <= 1) {
        if (t == 0) { inf->all_equal = 1; inf->v_thr = lut(xmin); inf->x_thr = 0x7fffffff; inf->otsu_bin = -1; }
        return;
    }
    const int seg = (nb + 255) / 256;
    const int b0 = t * seg, b1 = min(nb, b0 + seg);
    unsigned long long w = 0, s = 0;
    for (int b = b0; b < b1; ++b) {
        const unsigned long long c = gh[base_bin + b];
        w += c; s += c * (unsigned long long)lut(xmin + b);
    }
    s_w[t] = w; s_s[t] = s;
    __syncthreads();
    if (t == 0) {                                   // exclusive scan over the 256 segment sums
        unsigned long long aw = 0, as = 0;
        for (int k = 0; k < 256; ++k) {
            const unsigned long long cw = s_w[k], cs = s_s[k];
            s_w[k] = aw; s_s[k] = as; aw += cw; as += cs;
        }
        s_lo = 0;   // reuse as scratch
    }
    __syncthreads();
    __shared__ unsigned long long s_totw, s_tots;
    if (t == 255) { s_totw = s_w[255] + w; s_tots = s_s[255] + s; }
    __syncthreads();
    const unsigned long long totw = s_totw, tots = s_tots;
    unsigned long long cw = s_w[t], cs = s_s[t];
    double bv = -1.0; int bi = -1;
    for (int b = b0; b < b1; ++b) {
        const unsigned long long c = gh[base_bin + b];
        const int v = lut(xmin + b);
        cw += c; cs += c * (unsigned long long)v;
        if (b < nb - 1 && lut(xmin + b + 1) != v) {
            const unsigned long long w2 = totw - cw;
            const double m1 = (double)cs / (double)(float)cw;
            const double m2 = (double)(tots - cs) / (double)(float)w2;
            const float ww = (float)cw * (float)w2;
            const double d = m1 - m2;
            const double var = (double)ww * (d * d);
            if (bi < 0 || var > bv) { bv = var; bi = b; }
        }
    }
    s_bv[t] = bv; s_bi[t] = bi;
    __syncthreads();
    if (t == 0) {
        double best = -1.0; int idx = -1;
        for (int k = 0; k < 256; ++k)
            if (s_bi[k] >= 0 && (idx < 0 || s_bv[k] > best)) { best = s_bv[k]; idx = s_bi[k]; }
        if (idx < 0) {                               // gamma table flat over the whole range: all gamma values equal
            inf->all_equal = 1; inf->v_thr = lut(xmin); inf->x_thr = 0x7fffffff; inf->otsu_bin = -1;
        } else {
            const int v = lut(xmin + idx);
            inf->v_thr = v; inf->x_thr = xmin + idx; inf->otsu_bin = v - lut(xmin);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Otsu, colour path: 256 float bins, sequential like the reference (one lane), then the gray-space threshold.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_otsu_rgb(Job j) {
    __shared__ OtsuScratch ws;
    __shared__ uint32_t cnt[256];
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    if (inf->all_equal) {
        if (threadIdx.x == 0) { inf->v_thr_f = inf->vmin; inf->g_thr = __longlong_as_double(0x7ff0000000000000LL); }
        return;
    }
    for (int k = threadIdx.x; k < 256; k += 64) cnt[k] = j.hist[(int64_t)img * j.hist_stride + k];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double centre;
    const int idx = otsu_float256(cnt, inf->vmin, inf->vmax, &centre, &ws);
    inf->otsu_bin = idx; inf->v_thr_f = centre;
    // mask = pow(gray, 0.4) > centre  <=>  gray >= T  with T the first double whose power exceeds the centre
    const double T = first_true(inf->gmin, inf->gmax, [centre](double g) { return gamma_pow(g) > centre; });
    const long long tb = __double_as_longlong(T);
    inf->g_thr = (tb > 0 && tb < 0x7ff0000000000000LL) ? __longlong_as_double(tb - 1) : T;
}

// ------------------------------------------------------------------------------------------------------------------
static int blocks_per_image(const Job& j, int px_per_block) {
    const int64_t npix = (int64_t)j.H * j.W;
    int64_t need = (npix + px_per_block - 1) / px_per_block;
    int64_t want = (4096 + j.n - 1) / j.n;          // keep the whole launch around 4096 blocks
    if (want < 1) want = 1;
    if (need > want) need = want;
    if (need < 1) need = 1;
    return (int)need;
}

#define LFA_DISPATCH_V(KERNEL, GRID, BLOCK, STREAM, JOB)                                         \
    switch ((JOB).gray_variant) {                                                                \
        case 0: LFA_LAUNCH(#KERNEL, KERNEL<0>, GRID, BLOCK, 0, STREAM, JOB); break;              \
        case 1: LFA_LAUNCH(#KERNEL, KERNEL<1>, GRID, BLOCK, 0, STREAM, JOB); break;              \
        case 2: LFA_LAUNCH(#KERNEL, KERNEL<2>, GRID, BLOCK, 0, STREAM, JOB); break;              \
        case 3: LFA_LAUNCH(#KERNEL, KERNEL<3>, GRID, BLOCK, 0, STREAM, JOB); break;              \
        case 4: LFA_LAUNCH(#KERNEL, KERNEL<4>, GRID, BLOCK, 0, STREAM, JOB); break;              \
        default: LFA_LAUNCH(#KERNEL, KERNEL<5>, GRID, BLOCK, 0, STREAM, JOB); break;             \
    }

hipError_t launch_threshold(const Job& j, hipStream_t s) {
    LFA_LAUNCH("k_init_info", k_init_info, dim3((j.n + 255) / 256), dim3(256), 0, s, j);
    if (j.C == 3) {
        hipError_t e = hipMemsetAsync(j.hist, 0, (size_t)j.n * j.hist_stride * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        const dim3 grid(blocks_per_image(j, 256 * 64), j.n);
        LFA_DISPATCH_V(k_minmax_rgb, grid, dim3(256), s, j);
        LFA_LAUNCH("k_edges_rgb", k_edges_rgb, dim3(j.n), dim3(320), 0, s, j);
        LFA_DISPATCH_V(k_hist_rgb, grid, dim3(256), s, j);
        LFA_LAUNCH("k_otsu_rgb", k_otsu_rgb, dim3(j.n), dim3(64), 0, s, j);
    } else if (j.dtype == LFA_U8) {
        hipError_t e = hipMemsetAsync(j.hist, 0, (size_t)j.n * j.hist_stride * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        const dim3 grid(blocks_per_image(j, 256 * 128), j.n);
        LFA_LAUNCH("k_hist_u8", k_hist_u8, grid, dim3(256), 0, s, j);
        LFA_LAUNCH("k_otsu_int", k_otsu_int, dim3(j.n), dim3(256), 0, s, j);
    } else {
        const dim3 grid(blocks_per_image(j, 256 * 64), j.n);
        LFA_LAUNCH("k_minmax_u16", k_minmax_u16, grid, dim3(256), 0, s, j);
        LFA_LAUNCH("k_zero_hist_u16", k_zero_hist_u16, dim3(16, j.n), dim3(256), 0, s, j);
        const dim3 gridh(blocks_per_image(j, 1024 * 64), j.n);
        LFA_LAUNCH("k_hist_u16", k_hist_u16, gridh, dim3(1024), 0, s, j);
        LFA_LAUNCH("k_otsu_int", k_otsu_int, dim3(j.n), dim3(256), 0, s, j);
    }
    return hipGetLastError();
}

}  // namespace lfa
