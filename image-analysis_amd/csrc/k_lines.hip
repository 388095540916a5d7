// k_lines.hip -- detect_lines (LFA-Image_Analysis/detection.py:156-201), SURVEY.md section 8f.3  (gfx950)
//
// The reference defines this function and never calls it.  Its stages (the first Gaussian, detection.py:166, included:
// k_lines_front) are restated here one kernel per dependency -- every stage needs a whole-image scalar of the one before
// (extremes, tile histograms, percentiles), so nothing fuses across them.  Reference arithmetic (SP = /root/reference/
// image-analysis/lib/python3.10/site-packages):
//   threshold_local(gaussian, block 3, 'reflect')   SP/skimage/filters/thresholding.py:236-243 -> scipy correlate1d, 3 taps
//   adjust_gamma(float image, 0.5)                  SP/skimage/exposure/exposure.py:679   (x / 1.0) ** 0.5 * 1.0 * 1
//   equalize_adapthist(clip_limit 0.05)             SP/skimage/exposure/_adapthist.py:72-89, _clahe :117-226, clip_histogram
//                                                   :248-280, map_histogram :305-310 ; img_as_uint SP/skimage/util/dtype.py
//   np.percentile (80, 99), method 'linear'         numpy lib/_function_base_impl.py _quantile / _lerp
//   rescale_intensity(in_range=(p80, p99), (0, 1))  SP/skimage/exposure/exposure.py:577-589
//   label of a non-boolean image                    SP/skimage/measure/_label.py:30-120 -> _ccomp.label_cython on an np.intp copy of
//                                                   the image (equal INTEGER values, 8-connected, 0 = background, numbered in
//                                                   raster order of the first pixel): of a float image in [0, 1] only the
//                                                   pixels equal to 1.0 are foreground, as one class
//   regionprops .bbox + the area / shape filter     detection.py:187-200
// After CLAHE the image holds at most 16384 distinct values (the uint16 `levels`), and everything downstream is a
// monotone function of the level: percentiles come from the level histogram, the float the reference would hold for a
// level is computed once per level (`lval`), and "equal integer values" becomes "equal class" with class = (intp) lval
// (0 = background below 1.0, 1 at 1.0).  Components are labelled on the class image with a pixel union-find
// whose roots are the smallest pixel index, i.e. the raster-first pixel the reference numbers its labels by.
// -ffp-contract=off (the build's flag): every float operation below is the single IEEE operation numpy performs.
#include "lfa_internal.h"
#include "lfa_device.h"
#include <type_traits>

namespace lfa {

// ---- stage 2: local threshold, gamma 0.5, img_as_uint ---------------------------------------------------------------------
__device__ __forceinline__ int reflect_half(int i, int n) {         // scipy 'reflect' (d c b a | a b c d | d c b a), |overshoot| <= 1
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// One kernel for detection.py:166-172: skimage.filters.gaussian(sigma 1) -- img_as_float, then scipy's correlate1d along axis 0
// and along axis 1 with 'nearest' borders, exactly as k_gaussian.hip restates it (tmp = x[0]*w[R]; tmp += (x[-k] + x[+k]) *
// w[R-k] for k = R..1) -- then threshold_local's 3-tap Gaussian ('reflect' borders) of that result, the product with
// (t > 0), sqrt and img_as_uint.  The float64 Gaussian image never leaves the CU (8 bytes per pixel written and read back
// otherwise); it is stored only when the parity tests ask for it (j.g, debug flag 64).
// A block stages kFIW columns x (kFTH + 2 + 2R) rows: vertical pass into LDS (one staged column per thread, all rows
// requested before the first is used, sliding register window), horizontal pass out of LDS and back into it (the
// Gaussian on kFTH + 2 rows x kFIW - 2R columns), then the local filter on the inner kFTH x (kFIW - 2R - 2) pixels.
constexpr int kFR = 4, kFTH = 16, kFIW = 256, kFOC = kFIW - 2 * (kFR + 1);

template <typename PIX>
__global__ __launch_bounds__(kFIW) void k_lines_front(LinesJob j) {
    __shared__ double v[(kFTH + 2) * kFIW];
    __shared__ int s_red[2 * (kFIW / 64)];
    const int img = blockIdx.z, t = threadIdx.x;
    const PIX* __restrict__ base = (const PIX*)(j.images + (int64_t)img * j.img_bytes);
    const int x0 = blockIdx.x * kFOC, r0 = blockIdx.y * kFTH;
    const double scale = (sizeof(PIX) == 1) ? (1.0 / 255.0) : (1.0 / 65535.0);
    double w[kFR + 1];
#pragma unroll
    for (int k = 0; k <= kFR; ++k) w[k] = j.wg[k];
    {   // axis 0: staged column t is image column x0 - (R + 1) + t ('nearest'), rows r0 - 1 - R .. r0 + kFTH + R
        const PIX* col = base + min(max(x0 - (kFR + 1) + t, 0), j.W - 1);
        PIX raw[kFTH + 2 + 2 * kFR];
#pragma unroll
        for (int k = 0; k < kFTH + 2 + 2 * kFR; ++k) raw[k] = col[(int64_t)min(max(r0 - 1 - kFR + k, 0), j.H - 1) * j.W];
        double win[2 * kFR + 1];
#pragma unroll
        for (int k = 0; k < 2 * kFR; ++k) win[k + 1] = (double)raw[k] * scale;
#pragma unroll
        for (int rr = 0; rr < kFTH + 2; ++rr) {
#pragma unroll
            for (int k = 0; k < 2 * kFR; ++k) win[k] = win[k + 1];
            win[2 * kFR] = (double)raw[rr + 2 * kFR] * scale;
            double tmp = win[kFR] * w[kFR];
#pragma unroll
            for (int q = kFR; q >= 1; --q) tmp = tmp + (win[kFR - q] + win[kFR + q]) * w[kFR - q];
            v[rr * kFIW + t] = tmp;
        }
    }
    __syncthreads();
    {   // axis 1, in place: every thread holds its column's results until all have read their taps
        double h[kFTH + 2];
        const bool has = t >= kFR && t < kFIW - kFR;
        if (has) {
#pragma unroll
            for (int rr = 0; rr < kFTH + 2; ++rr) {
                const double* row = v + rr * kFIW + t;
                double tmp = row[0] * w[kFR];
#pragma unroll
                for (int q = kFR; q >= 1; --q) tmp = tmp + (row[-q] + row[q]) * w[kFR - q];
                h[rr] = tmp;
            }
        }
        __syncthreads();
        if (has) {
#pragma unroll
            for (int rr = 0; rr < kFTH + 2; ++rr) v[rr * kFIW + t] = h[rr];
        }
    }
    __syncthreads();
    // the local filter: LDS row rr is image row r0 - 1 + rr, LDS column tt image column x0 - (R + 1) + tt; 'reflect' with an
    // overshoot of one is the clamp of the row / column index
    const int64_t npix = (int64_t)j.H * j.W;
    uint16_t* __restrict__ u = j.u + (int64_t)img * npix;
    const int x = x0 - (kFR + 1) + t;
    int lo = 65535, hi = 0, over = 0;
    if (t >= kFR + 1 && t < kFIW - kFR - 1 && x < j.W) {
        const int tm = max(x - 1, 0) - (x0 - (kFR + 1)), tp = min(x + 1, j.W - 1) - (x0 - (kFR + 1));
#pragma unroll
        for (int rr = 0; rr < kFTH; ++rr) {
            const int y = r0 + rr;
            if (y >= j.H) break;
            const double* rc = v + (rr + 1) * kFIW;
            const double* rm = v + (max(y - 1, 0) - (r0 - 1)) * kFIW;
            const double* rp = v + (min(y + 1, j.H - 1) - (r0 - 1)) * kFIW;
            auto col = [&](int tt) -> double {                        // axis 0 first (scipy.ndimage.gaussian_filter)
                double c = rc[tt] * j.w0;
                c += (rm[tt] + rp[tt]) * j.w1;
                return c;
            };
            const double gv = rc[t];
            const double c0 = col(t), cm = col(tm), cp = col(tp);
            double tl = c0 * j.w0;
            tl += (cm + cp) * j.w1;
            const double xg = (tl - 0.0 > 0.0) ? gv * 1.0 : gv * 0.0; // image_copy * (thresholded_image > 0), detection.py:170
            const double sq = sqrt(xg / 1.0) * 1.0 * 1.0;             // ((image / scale) ** gamma) * scale * gain
            double q = rint(sq * 65535.0);
            q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
            const int qi = (int)q;
            if (sq > 1.0 || sq < -1.0) over = 1;                      // img_as_uint refuses such an image
            u[(int64_t)y * j.W + x] = (uint16_t)qi;
            lo = min(lo, qi); hi = max(hi, qi);
            if (j.g) j.g[(int64_t)img * npix + (int64_t)y * j.W + x] = gv;
        }
    }
    // the frame's extremes: one pair of atomics per block, and none once the frame's running extremes already cover the block's
    lo = wave_min_i32(lo); hi = wave_max_i32(hi);
    if ((t & 63) == 0) { s_red[t >> 6] = lo; s_red[kFIW / 64 + (t >> 6)] = hi; }
    if (over) atomicOr(&j.info[img].status, 2);
    __syncthreads();
    if (t == 0) {
#pragma unroll
        for (int k = 1; k < kFIW / 64; ++k) { lo = min(lo, s_red[k]); hi = max(hi, s_red[kFIW / 64 + k]); }
        if (hi >= lo) {
            if (lo < __atomic_load_n(&j.info[img].umin, __ATOMIC_RELAXED)) atomicMin(&j.info[img].umin, lo);
            if (hi > __atomic_load_n(&j.info[img].umax, __ATOMIC_RELAXED)) atomicMax(&j.info[img].umax, hi);
        }
    }
}

__global__ void k_lines_init(LinesJob j) {
    const int img = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        LfaLinesInfo z;
        z.umin = 65535; z.umax = 0; z.cmin = 65535; z.cmax = 0; z.p80 = 0.0; z.p99 = 0.0; z.n_labels = 0; z.n_boxes = 0; z.status = 0; z.pad = 0;
        j.info[img] = z;
    }
    for (int k = t; k < 16384; k += gridDim.x * blockDim.x) j.lhist[(int64_t)img * 16384 + k] = 0u;
    const int nmaps = j.nty * j.ntx * 256;                               // k_clahe_hist accumulates the tile histograms here
    for (int k = t; k < nmaps; k += gridDim.x * blockDim.x) j.maps[(int64_t)img * nmaps + k] = 0;
}

// ---- stage 3: value -> CLAHE bin table --------------------------------------------------------------------------------------
// rescale_intensity(image, out_range=(0, 16383)), np.round, cast (_adapthist.py:74-76), then lut //= 65 (:133-135)
__global__ void k_lines_binlut(LinesJob j) {
    const int img = blockIdx.y;
    const double imin = (double)j.info[img].umin, imax = (double)j.info[img].umax;
    uint8_t* lut = j.binlut + (int64_t)img * 65536;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < 65536; v += gridDim.x * blockDim.x) {
        double c = (double)v;
        c = c < imin ? imin : (c > imax ? imax : c);
        double r;
        if (imin != imax) { r = (c - imin) / (imax - imin); r = r * (16383.0 - 0.0) + 0.0; }
        else r = c < 0.0 ? 0.0 : (c > 16383.0 ? 16383.0 : c);
        const int v14 = (int)rint(r);
        lut[v] = (uint8_t)(v14 / 65);
    }
}

// np.pad(mode='reflect') of an axis of n samples: index i of the unpadded axis, any overshoot (period 2 (n - 1))
__device__ __forceinline__ int reflect_pad(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

// ---- stage 4: tile histograms (kTileSlices blocks per CLAHE tile), then one block per tile: clip_histogram, map_histogram ----
constexpr int kTileSlices = 4;

__global__ __launch_bounds__(256) void k_clahe_hist(LinesJob j) {
    __shared__ int h[256];
    const int img = blockIdx.y, tile = blockIdx.x / kTileSlices, slice = blockIdx.x - tile * kTileSlices;
    const int ty = tile / j.ntx, tx = tile - ty * j.ntx;
    const int b = threadIdx.x;
    const int64_t npix = (int64_t)j.H * j.W;
    const uint16_t* __restrict__ u = j.u + (int64_t)img * npix;
    const uint8_t* __restrict__ lut = j.binlut + (int64_t)img * 65536;
    h[b] = 0;
    __syncthreads();
    const int y0 = j.kh / 2 + ty * j.kh - j.sy, x0 = j.kw / 2 + tx * j.kw - j.sx;   // the tile in padded coordinates (:145-146), pad removed
    const int per = (j.kh + kTileSlices - 1) / kTileSlices;
    const int r_lo = slice * per, r_hi = min(j.kh, r_lo + per);
    for (int c = b; c < j.kw; c += 256) {                                // a thread keeps its column; four rows' loads go out together
        const int xx = reflect_pad(x0 + c, j.W);
        for (int r0 = r_lo; r0 < r_hi; r0 += 4) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = u[(int64_t)reflect_pad(y0 + min(r0 + k, r_hi - 1), j.H) * j.W + xx];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lut[v[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (r0 + k < r_hi) atomicAdd(&h[v[k]], 1);
        }
    }
    __syncthreads();
    if (h[b]) atomicAdd(&j.maps[((int64_t)img * j.nty * j.ntx + tile) * 256 + b], h[b]);
}

__global__ __launch_bounds__(256) void k_clahe_tiles(LinesJob j) {
    __shared__ int h[256];
    __shared__ int s_sum[4];
    const int img = blockIdx.y, tile = blockIdx.x;
    const int b = threadIdx.x;
    const int n_el = j.kh * j.kw;
    h[b] = j.maps[((int64_t)img * j.nty * j.ntx + tile) * 256 + b];      // the tile's histogram (k_clahe_hist); its map replaces it below
    __syncthreads();
    auto block_sum = [&](int v) -> int {                                  // every thread gets the sum
        v = (int)wave_sum_u32((uint32_t)v);
        __syncthreads();
        if ((b & 63) == 0) s_sum[b >> 6] = v;
        __syncthreads();
        return s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    };
    // clip_histogram (_adapthist.py:248-280); thread b owns bin b
    const int clim = j.clim;
    int hv = h[b];
    int n_excess = block_sum(hv > clim ? hv - clim : 0);
    if (hv > clim) hv = clim;
    const int bin_incr = n_excess / 256;
    const int upper = clim - bin_incr;
    const bool low = hv < upper;
    n_excess -= block_sum(low ? 1 : 0) * bin_incr;
    if (low) hv += bin_incr;
    const bool mid = hv >= upper && hv < clim;
    n_excess += block_sum(mid ? hv - clim : 0);
    if (mid) hv = clim;
    while (n_excess > 0) {
        const int before = n_excess;
        for (int index = 0; index < 256; ++index) {
            const bool under = hv < clim;
            const int cnt = block_sum(under ? 1 : 0);
            int step = cnt / n_excess;
            if (step < 1) step = 1;
            const bool sel = under && b >= index && ((b - index) % step) == 0;
            if (sel) hv += 1;
            n_excess -= block_sum(sel ? 1 : 0);
            if (n_excess <= 0) break;
        }
        if (before == n_excess) break;
    }
    __syncthreads();
    h[b] = hv;
    __syncthreads();
    // map_histogram (:305-310): cumulative sum, scale, clip, truncate
    if (b == 0) { int a = 0; for (int k = 0; k < 256; ++k) { a += h[k]; h[k] = a; } }
    __syncthreads();
    double m = (double)h[b];
    m *= (double)(16383 - 0) / (double)n_el;
    m += 0.0;
    if (m > 16383.0) m = 16383.0;
    j.maps[((int64_t)img * j.nty * j.ntx + tile) * 256 + b] = (int32_t)m;
}

// ---- stage 5: bilinear blend of the four neighbouring tile maps (_adapthist.py:176-210) ------------------------------------
constexpr int kBlendRows = 255;                                      // most rows a block blends (255 x 256 pixels fit its 16-bit counters)

__global__ __launch_bounds__(256) void k_clahe_blend(LinesJob j, int rows_per_block) {
    // level histogram of this block's pixels: 16-bit counters, two per word (a block sees rows_per_block * 256 <= 65 280 pixels),
    // 32 KiB instead of 64 so that five blocks share a CU instead of two
    __shared__ uint32_t sh[8192];
    __shared__ double s_cy[kBlendRows];                                  // per row of the patch: weight of the lower tile row,
    __shared__ int s_ty[kBlendRows][2];                                  // offsets of the two tile rows' maps
    __shared__ uint16_t s_maps[9 * 256];
    for (int k = threadIdx.x; k < 8192; k += 256) sh[k] = 0u;
    const int img = blockIdx.y;
    const int64_t npix = (int64_t)j.H * j.W;
    const uint16_t* __restrict__ u = j.u + (int64_t)img * npix;
    const uint8_t* __restrict__ lut = j.binlut + (int64_t)img * 65536;
    const int32_t* __restrict__ maps = j.maps + (int64_t)img * j.nty * j.ntx * 256;
    uint16_t* __restrict__ lv = j.levels + (int64_t)img * npix;
    const int ncb = (j.W + 255) / 256;
    const int rb = blockIdx.x / ncb, cb = blockIdx.x - rb * ncb;
    const int y0 = rb * rows_per_block, x = cb * 256 + threadIdx.x;
    const int rows = min(rows_per_block, j.H - y0);
    // The tile maps this patch blends between: tile rows ty_a..ty_b, tile columns tx_a..tx_b.  When they are at most nine
    // (always, for patches no larger than a tile) the block copies them into LDS as 16-bit values (a map entry is at most
    // 16 383) and the four lookups per pixel never leave the CU; otherwise they stay gathers from global memory.
    const int x_last = min(j.W, cb * 256 + 256) - 1;
    const int ty_a = min(max((y0 + j.sy) / j.kh - 1, 0), j.nty - 1), ty_b = min(max((y0 + rows - 1 + j.sy) / j.kh, 0), j.nty - 1);
    const int tx_a = min(max((cb * 256 + j.sx) / j.kw - 1, 0), j.ntx - 1), tx_b = min(max((x_last + j.sx) / j.kw, 0), j.ntx - 1);
    const int nyt = ty_b - ty_a + 1, nxt = tx_b - tx_a + 1;
    const bool staged = nyt * nxt <= 9;
    if (staged) {
        for (int k = threadIdx.x; k < nyt * nxt * 256; k += 256) {
            const int t = k >> 8, tyy = ty_a + t / nxt, txx = tx_a + t % nxt;
            s_maps[k] = (uint16_t)maps[((int64_t)tyy * j.ntx + txx) * 256 + (k & 255)];
        }
    }
    // the divisions of _adapthist.py:176-210 are per row and per column, not per pixel: one thread per row of the patch
    // computes the row's, every thread its own column's
    if ((int)threadIdx.x < rows) {
        const int py = y0 + threadIdx.x + j.sy, by = py / j.kh;
        const int t0 = min(max(by - 1, 0), j.nty - 1), t1 = min(max(by, 0), j.nty - 1);
        s_cy[threadIdx.x] = (double)(py % j.kh) / (double)j.kh;
        s_ty[threadIdx.x][0] = staged ? (t0 - ty_a) * nxt * 256 : t0 * j.ntx * 256;
        s_ty[threadIdx.x][1] = staged ? (t1 - ty_a) * nxt * 256 : t1 * j.ntx * 256;
    }
    __syncthreads();
    int lo = 65535, hi = 0;
    auto blend = [&](auto staged_c) {
        constexpr bool kStaged = decltype(staged_c)::value;
        const int px = x + j.sx, bx = px / j.kw;
        const double cx1 = (double)(px % j.kw) / (double)j.kw, cx0 = 1.0 - cx1;
        const int c0 = min(max(bx - 1, 0), j.ntx - 1), c1 = min(max(bx, 0), j.ntx - 1);
        const int tx0 = (kStaged ? c0 - tx_a : c0) * 256, tx1 = (kStaged ? c1 - tx_a : c1) * 256;
        for (int r0 = 0; r0 < rows; r0 += 4) {                           // four rows per step: their loads are issued together
            int bin[4], m[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) bin[k] = u[(int64_t)(y0 + min(r0 + k, rows - 1)) * j.W + x];
#pragma unroll
            for (int k = 0; k < 4; ++k) bin[k] = lut[bin[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = min(r0 + k, rows - 1), t0 = s_ty[r][0], t1 = s_ty[r][1];
                if (kStaged) {
                    m[k][0] = s_maps[t0 + tx0 + bin[k]]; m[k][1] = s_maps[t0 + tx1 + bin[k]];
                    m[k][2] = s_maps[t1 + tx0 + bin[k]]; m[k][3] = s_maps[t1 + tx1 + bin[k]];
                } else {
                    m[k][0] = maps[t0 + tx0 + bin[k]]; m[k][1] = maps[t0 + tx1 + bin[k]];
                    m[k][2] = maps[t1 + tx0 + bin[k]]; m[k][3] = maps[t1 + tx1 + bin[k]];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (r0 + k >= rows) break;
                const double cy1 = s_cy[r0 + k], cy0 = 1.0 - cy1;
                float res = 0.0f;                                        // the reference's float32 accumulator, edge order (0,0) (0,1) (1,0) (1,1)
                res = res + (float)((double)m[k][0] * (cx0 * cy0));
                res = res + (float)((double)m[k][1] * (cx1 * cy0));
                res = res + (float)((double)m[k][2] * (cx0 * cy1));
                res = res + (float)((double)m[k][3] * (cx1 * cy1));
                const int q = (int)res;                                  // result.astype(uint16): truncation
                lv[(int64_t)(y0 + r0 + k) * j.W + x] = (uint16_t)q;
                atomicAdd(&sh[(q & 16383) >> 1], (q & 1) ? 0x10000u : 1u);
                lo = min(lo, q); hi = max(hi, q);
            }
        }
    };
    if (x < j.W) {
        if (staged) blend(std::true_type{}); else blend(std::false_type{});
    }
    lo = wave_min_i32(lo); hi = wave_max_i32(hi);
    if ((threadIdx.x & 63) == 0 && hi >= lo) { atomicMin(&j.info[img].cmin, lo); atomicMax(&j.info[img].cmax, hi); }
    __syncthreads();
    uint32_t* gh = j.lhist + (int64_t)img * 16384;
    // two neighbouring 32-bit bins of the global histogram are one 64-bit word and take one add (the low bin counts pixels of
    // one frame, < 2^30, so it never carries into the high one)
    unsigned long long* gh2 = (unsigned long long*)gh;
    for (int k = threadIdx.x; k < 8192; k += 256) {
        const uint32_t w = sh[k];
        if (w) atomicAdd(&gh2[k], ((unsigned long long)(w >> 16) << 32) | (unsigned long long)(w & 0xffffu));
    }
}

// ---- stage 6: percentiles, the float value of every level, classes of equal value -----------------------------------------
// rescale_intensity of a float64 image with in_range (lo, hi) and out_range (0, 1)  (exposure.py:577-589)
__device__ __forceinline__ double rescale_unit(double x, double lo, double hi) {
    const double c = x < lo ? lo : (x > hi ? hi : x);
    if (lo != hi) return (c - lo) / (hi - lo) * (1.0 - 0.0) + 0.0;
    return c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
}

// exclusive prefix of one value per thread over a block of 1024 (16 waves): shuffles inside the wave, the 16 wave totals
// through LDS; every thread also gets the block's total.  (Two of these replaced two 1024-step loops of thread 0.)
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    __syncthreads();                                                     // (s_w may still be read from the previous scan)
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t w = s_w[k]; all += w; if (k < wv) before += w; }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(1024) void k_lines_levels(LinesJob j) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t part[1024];
    __shared__ int s_level[4];
    __shared__ double s_p[2];
    const int img = blockIdx.x, t = threadIdx.x;
    uint32_t* gh = j.lhist + (int64_t)img * 16384;                       // becomes its own inclusive prefix
    uint32_t* cum = gh;
    LfaLinesInfo* inf = &j.info[img];
    uint32_t loc[16], a = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a += gh[t * 16 + k]; loc[k] = a; }
    uint32_t all_px;
    part[t] = block_scan_1024(a, s_w, &all_px);
#pragma unroll
    for (int k = 0; k < 16; ++k) cum[t * 16 + k] = part[t] + loc[k];
    __syncthreads();
    const double cmin = (double)inf->cmin, cmax = (double)inf->cmax;
    const int64_t n = (int64_t)j.H * j.W;
    // np.percentile(e, (80, 99)): virtual index (n - 1) * q, floor / +1 neighbours, _lerp
    int64_t want[4];
    double gam[2];
    for (int k = 0; k < 2; ++k) {
        const double quant = (k == 0 ? 80.0 : 99.0) / 100.0;
        const double vi = (double)(n - 1) * quant;
        int64_t prev = (int64_t)floor(vi), next = prev + 1;
        if (vi >= (double)(n - 1)) { prev = n - 1; next = n - 1; }       // (index -1 of the sorted array)
        gam[k] = vi - floor(vi);
        want[2 * k] = prev; want[2 * k + 1] = next;
    }
    if (t < 4) {                                                         // smallest level L with cum[L] > rank
        int lo = 0, hi = 16383;
        const uint32_t rank = (uint32_t)want[t];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[mid] > rank) hi = mid; else lo = mid + 1; }
        s_level[t] = lo;
    }
    __syncthreads();
    if (t < 2) {
        const double va = rescale_unit((double)s_level[2 * t], cmin, cmax), vb = rescale_unit((double)s_level[2 * t + 1], cmin, cmax);
        const double g = gam[t], diff = vb - va;
        double r = va + diff * g;
        if (g >= 0.5) r = vb - diff * (1.0 - g);
        s_p[t] = r;
    }
    __syncthreads();
    const double p80 = s_p[0], p99 = s_p[1];
    if (t == 0) { inf->p80 = p80; inf->p99 = p99; }
    // value the reference holds for each level after both rescales, and its class: label() of a non-boolean image works on
    // an np.intp copy of it (skimage/measure/_ccomp: typed buffer of DTYPE = intp), so a float in [0, 1] is truncated -- 0
    // (background) below 1.0, class 1 at exactly 1.0, i.e. at and above the 99th percentile
    double* lval = j.lval + (int64_t)img * 16384;
    uint16_t* cls = j.cls + (int64_t)img * 16384;
    int cl[16], first = 16384, top = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double v = rescale_unit(rescale_unit((double)(t * 16 + k), cmin, cmax), p80, p99);
        lval[t * 16 + k] = v;
        cl[k] = (v >= 1.0) ? (int)(long long)v : 0;                       // (both rescales end in [0, 1]; NaN -> 0 like the cast of a NaN here)
        cls[t * 16 + k] = (uint16_t)cl[k];
        if (cl[k] > 0 && first == 16384) first = t * 16 + k;
        top = max(top, cl[k]);
    }
    const uint32_t all_cls = (uint32_t)__syncthreads_or(top);             // 0 or 1
    // The table is "0 up to a level a, then 1" (both rescales are non-decreasing): k_ccl_init then computes
    // min(max(level - a, 0), m) instead of looking the class up.  Checked here entry by entry; where it does not hold the
    // table is used.
    __syncthreads();                                                     // (s_level: the percentile search above is done with it)
    if (t == 0) { s_level[0] = 16384; s_level[1] = (int)all_cls; s_level[2] = 1; }
    __syncthreads();
    if (first < 16384) atomicMin(&s_level[0], first);
    __syncthreads();
    const int ca = s_level[0] - 1, cm = s_level[1];
    bool same = true;
#pragma unroll
    for (int k = 0; k < 16; ++k) same = same && cl[k] == min(max(t * 16 + k - ca, 0), cm);
    if (!same) s_level[2] = 0;
    __syncthreads();
    if (t == 0) {
        int32_t* fn = j.clsfn + 4 * img;
        fn[0] = ca; fn[1] = cm; fn[2] = (s_level[2] && !j.force_table) ? 1 : 0; fn[3] = 0;
    }
}

// ---- stage 7: components of equal class, 8-connected ------------------------------------------------------------------------
__device__ __forceinline__ int ccl_find(const int32_t* parent, int x) {
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

__device__ __forceinline__ void ccl_union(int32_t* parent, int a, int b) {
    for (;;) {
        a = ccl_find(parent, a); b = ccl_find(parent, b);
        if (a == b) return;
        if (a < b) { const int s = a; a = b; b = s; }                    // the larger root hangs under the smaller one
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_ccl_init(LinesJob j) {
    const int img = blockIdx.y;
    const int64_t npix = (int64_t)j.H * j.W;
    int32_t* __restrict__ parent = j.parent + (int64_t)img * npix;
    const uint16_t* __restrict__ lv = j.levels + (int64_t)img * npix;
    const uint16_t* __restrict__ cls = j.cls + (int64_t)img * 16384;
    uint16_t* __restrict__ cimg = j.cimg + (int64_t)img * npix;
    const int cls_a = j.clsfn[4 * img], cls_m = j.clsfn[4 * img + 1];
    const bool cls_fn = j.clsfn[4 * img + 2] != 0;                      // k_lines_levels: class = min(max(level - a, 0), m)
    // Pixels of one class that sit side by side in a wave's 64 (one row) start out linked to the first of them: the long
    // horizontal runs of a line never form a chain of single links, and k_ccl_merge only joins runs.
    const int lane = threadIdx.x & 63;
    // 32-bit indices (a frame has at most 2^30 pixels, lfa_detect_lines_batch) and the column kept by addition: the only
    // division is the one that finds the thread's first column -- `p % W` per pixel was most of this kernel's instructions
    const int np32 = (int)npix, stride = (int)(gridDim.x * blockDim.x);  // (stride: a multiple of 64, so a wave's pixels start at a multiple of 64)
    const int W = j.W, dx = stride % W;
    int p0 = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int x0 = p0 % W;
    for (; p0 < np32; p0 += 4 * stride) {
        int l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int p = p0 + k * stride; l[k] = p < np32 ? lv[p] : 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k * stride;
            const bool valid = p < np32;                                 // (a valid lane's lower lanes are valid too)
            const int lev = l[k] & 16383;
            const int c = !valid ? 0 : (cls_fn ? min(max(lev - cls_a, 0), cls_m) : (int)cls[lev]);
            const int x = x0;                                            // p % W
            x0 += dx; if (x0 >= W) x0 -= W;
            const int cleft = __shfl_up(c, 1);
            const unsigned long long starts = __ballot(lane == 0 || x == 0 || c != cleft);
            const int first = 63 - __clzll((long long)(starts & (~0ull >> (63 - lane))));
            if (valid) {
                cimg[p] = (uint16_t)c;                                   // the class image: what the labelling compares
                parent[p] = c ? p - (lane - first) : -1;                 // background takes no part
            }
        }
    }
}

// Only edges that no other link already implies are joined: the left neighbour across a wave boundary of k_ccl_init (inside
// one it linked the run), the upper neighbour where a run's overlap with the run above starts (the pixel to the left makes
// the same join otherwise), a diagonal one when neither the upper nor the left pixel carries it.  The components are the
// same, with far fewer walks to a root.  c: the pixel's class (not 0); cl, cu, cul, cur: left, upper, upper left, upper
// right neighbour's class, 0 where there is none.
__device__ __forceinline__ void ccl_join(int32_t* parent, int p, int W, int c, int cl, int cu, int cul, int cur) {
    if (cl == c && (p & 63) == 0) ccl_union(parent, p, p - 1);
    if (cu == c) { if (!(cl == c && cul == c)) ccl_union(parent, p, p - W); }
    else {
        if (cul == c && cl != c) ccl_union(parent, p, p - W - 1);
        if (cur == c) ccl_union(parent, p, p - W + 1);
    }
}

__global__ __launch_bounds__(256) void k_ccl_merge(LinesJob j) {
    const int img = blockIdx.y;
    const int64_t npix = (int64_t)j.H * j.W;
    const uint16_t* __restrict__ cimg = j.cimg + (int64_t)img * npix;
    int32_t* parent = j.parent + (int64_t)img * npix;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((j.W & 7) == 0) {
        // rows of a multiple of 8 pixels: a lane takes 8 neighbouring pixels of one row with one 16-byte load, the 8 above
        // them with another and three single classes at the ends (left, upper left, upper right); two such groups per step
        const uint4* __restrict__ c8 = (const uint4*)cimg;
        const int64_t n8 = npix >> 3;
        const int W8 = j.W >> 3;
        for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < n8; q0 += 2 * stride) {
            uint4 cur[2], abv[2];
            int lft[2], ul[2], ur[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int64_t q = q0 + k * stride; cur[k] = q < n8 ? c8[q] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                abv[k] = make_uint4(0, 0, 0, 0); lft[k] = ul[k] = ur[k] = 0;
                if (cur[k].x | cur[k].y | cur[k].z | cur[k].w) {
                    const int64_t q = q0 + k * stride, p = q << 3;
                    const int y = (int)((uint32_t)p / (uint32_t)j.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)j.W);   // (p < 2^30)
                    if (x > 0) lft[k] = cimg[p - 1];
                    if (y > 0) {
                        abv[k] = c8[q - W8];
                        if (x > 0) ul[k] = cimg[p - j.W - 1];
                        if (x + 8 < j.W) ur[k] = cimg[p - j.W + 8];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!(cur[k].x | cur[k].y | cur[k].z | cur[k].w)) continue;
                const int p = (int)((q0 + k * stride) << 3);
                const uint32_t cw[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w}, aw[4] = {abv[k].x, abv[k].y, abv[k].z, abv[k].w};
                int c[9], a[10];                                         // c[0]: left of the group; a[0], a[9]: upper left / right of it
                c[0] = lft[k]; a[0] = ul[k]; a[9] = ur[k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c[1 + 2 * i] = (int)(cw[i] & 0xffffu); c[2 + 2 * i] = (int)(cw[i] >> 16);
                    a[1 + 2 * i] = (int)(aw[i] & 0xffffu); a[2 + 2 * i] = (int)(aw[i] >> 16);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c[i + 1]) ccl_join(parent, p + i, j.W, c[i + 1], c[i], a[i + 1], a[i], a[i + 2]);
            }
        }
        return;
    }
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < npix; p0 += 4 * stride) {
        // four pixels per step; all their loads are issued before the first walk to a root
        int c[4], cl[4], cu[4], cul[4], cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t p = p0 + k * stride; c[k] = p < npix ? cimg[p] : 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t p = p0 + k * stride, q = p - j.W;
            cl[k] = cu[k] = cul[k] = cur[k] = 0;
            if (c[k]) {
                const int y = (int)((uint32_t)p / (uint32_t)j.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)j.W);   // (p < 2^30)
                const bool hasl = x > 0, hasu = y > 0, hasr = x + 1 < j.W;
                if (hasl) cl[k] = cimg[p - 1];
                if (hasu) cu[k] = cimg[q];
                if (hasu && hasl) cul[k] = cimg[q - 1];
                if (hasu && hasr) cur[k] = cimg[q + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c[k]) ccl_join(parent, (int)(p0 + k * stride), j.W, c[k], cl[k], cu[k], cul[k], cur[k]);
    }
}

// A component's box is kept as three extents from its root, which is its raster-first pixel and so lies in its top row:
// rows down to the bottom edge, columns left to minc, columns right to maxc -- all "larger is wider" with 0 meaning "the
// root's own pixel", so the table needs no initialisation per call: it is all zero between calls (lfa_detect_lines_batch
// clears it when it allocates it, k_ccl_collect puts back the entries it reads).  A root some pixel points to is flagged in
// the class image (0xffff, no class: classes are < 16 385), which the labelling no longer needs at this point.
__global__ __launch_bounds__(256) void k_ccl_boxes(LinesJob j) {
    const int img = blockIdx.y;
    const int64_t npix = (int64_t)j.H * j.W;
    uint16_t* cimg = j.cimg + (int64_t)img * npix;
    int32_t* parent = j.parent + (int64_t)img * npix;
    int32_t* bbox = j.bbox + (int64_t)img * npix * 4;
    const int lane = threadIdx.x & 63;
    uint32_t roots = 0;
    // A wave takes 64 consecutive pixels per step (wave-uniform loop).  Pixels of one component that sit side by side in a
    // wave would all hit the same words of their root: they are combined in the wave first (one leader per distinct
    // root, DPP reductions) -- per-pixel atomics on the few large components cost 1.4 of this kernel's 1.7 ms.
    // one step of the whole wave: lane's pixel p with the parent entry `up` it loaded (-1: background or no pixel)
    auto step = [&](int64_t p, int up) {
        int r = -1, y = 0, x = 0;
        bool need = false;
        if (up == (int)p) ++roots;                                       // (most components of a CLAHE'd frame are single pixels)
        else if (up >= 0) {
            r = ccl_find(parent, up);
            parent[p] = r;                                               // (others only ever point to smaller indices)
            y = (int)((uint32_t)p / (uint32_t)j.W); x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)j.W);
            need = true;
        }
        unsigned long long todo = __ballot(need);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int lr = __shfl(r, leader);
            const bool mine = need && r == lr;
            const unsigned long long mb = __ballot(mine);
            int x0 = x, y1 = y + 1, x1 = x + 1;                          // the leader's own pixel: all there is when no other lane shares its root
            if (mb & (mb - 1)) {                                         // (wave-uniform) several lanes: combine them first
                x0 = wave_min_i32(mine ? x : 0x7fffffff);
                y1 = wave_max_i32(mine ? y + 1 : -1); x1 = wave_max_i32(mine ? x + 1 : -1);
            }
            if (lane == leader) {
                const int yr = (int)((uint32_t)lr / (uint32_t)j.W), xr = (int)((uint32_t)lr - (uint32_t)yr * (uint32_t)j.W);
                int32_t* e = bbox + 4 * (int64_t)lr;
                if (y1 - yr > 1) atomicMax(&e[0], y1 - yr);              // maxr - root row
                if (xr - x0 > 0) atomicMax(&e[1], xr - x0);              // root column - minc
                if (x1 - xr > 1) atomicMax(&e[2], x1 - xr);              // maxc - root column
                cimg[lr] = (uint16_t)0xffff;
            }
            todo &= ~mb;
        }
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((npix & 3) == 0) {
        // 16-byte loads: a lane takes four neighbouring pixels, a wave 256 per load, two loads in flight.  Almost every load
        // holds only background and single-pixel components, which need nothing but the count; a wave in which any lane
        // holds a pixel that points elsewhere goes through the four pixels of its lanes one after the other.
        const int4* __restrict__ p4 = (const int4*)parent;
        const int64_t n4 = npix >> 2;
        for (int64_t qb = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); qb < n4; qb += 2 * stride) {
            int4 ups[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int64_t q = qb + k * stride + lane; ups[k] = q < n4 ? p4[q] : make_int4(-1, -1, -1, -1); }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t p = (qb + k * stride + lane) << 2;
                const int u4[4] = {ups[k].x, ups[k].y, ups[k].z, ups[k].w};
                bool any = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) any = any || (u4[i] >= 0 && u4[i] != (int)(p + i));
                if (__ballot(any)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) step(p + i, u4[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) roots += (u4[i] == (int)(p + i)) ? 1u : 0u;
                }
            }
        }
    } else {
        for (int64_t pb0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); pb0 < npix; pb0 += 4 * stride) {
            int ups[4];                                                  // four segments per step: their loads go out together
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int64_t p = pb0 + k * stride + lane; ups[k] = p < npix ? parent[p] : -1; }
#pragma unroll
            for (int k = 0; k < 4; ++k) step(pb0 + k * stride + lane, ups[k]);
        }
    }
    roots = wave_sum_u32(roots);
    if (lane == 0 && roots) atomicAdd(&j.info[img].n_labels, (int)roots);
}

// one candidate record per component that passes detection.py:191-199: root pixel, minr, maxr
__global__ __launch_bounds__(256) void k_ccl_collect(LinesJob j) {
    const int img = blockIdx.y;
    const int64_t npix = (int64_t)j.H * j.W;
    const uint16_t* __restrict__ cimg = j.cimg + (int64_t)img * npix;
    int4* bbox = (int4*)(j.bbox + (int64_t)img * npix * 4);
    const double min_area = (double)j.W * ((double)j.H * 0.0065), max_area = (double)j.W * ((double)j.H * 0.04);   // :159-160
    int32_t* cand = j.cand + (int64_t)img * j.cap * 3;
    // (a root no pixel points to is a single pixel: one by one, never wider than tall, never a candidate)
    auto visit = [&](int64_t p) {
        const int4 e = bbox[p];
        bbox[p] = make_int4(0, 0, 0, 0);                                 // the table is all zero again when the call ends
        const int yr = (int)((uint32_t)p / (uint32_t)j.W), xr = (int)((uint32_t)p - (uint32_t)yr * (uint32_t)j.W);
        const int minr = yr, maxr = yr + max(e.x, 1), minc = xr - e.y, maxc = xr + max(e.z, 1);
        const int64_t area = (int64_t)(maxc - minc) * (maxr - minr);
        // wider than tall and min_area < area < max_area: the box is widened to the whole image width and kept (:193-200)
        if ((maxc - minc) > (maxr - minr) && (double)area > min_area && (double)area < max_area) {
            const int k2 = atomicAdd(&j.info[img].n_boxes, 1);
            if (k2 < j.cap) { cand[3 * k2] = (int)p; cand[3 * k2 + 1] = minr; cand[3 * k2 + 2] = maxr; }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((npix & 7) == 0) {                                               // every frame's flags start at a multiple of 16 bytes
        const uint4* __restrict__ c8 = (const uint4*)cimg;
        const int64_t n8 = npix >> 3;
        for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < n8; q0 += 2 * stride) {
            uint4 f[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int64_t q = q0 + k * stride; f[k] = q < n8 ? c8[q] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t w[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
                const int64_t p = (q0 + k * stride) << 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((w[i] & 0xffffu) == 0xffffu) visit(p + 2 * i);
                    if ((w[i] >> 16) == 0xffffu) visit(p + 2 * i + 1);
                }
            }
        }
        return;
    }
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < npix; p0 += 4 * stride) {
        int f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t p = p0 + k * stride; f[k] = p < npix ? cimg[p] : 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (f[k] == 0xffff) visit(p0 + k * stride);
    }
}

// regionprops order = label order = raster order of the first pixel = ascending root index
__global__ __launch_bounds__(256) void k_lines_emit(LinesJob j, LfaLineBox* out, int32_t* counts, int max_boxes) {
    const int img = blockIdx.x;
    LfaLinesInfo* inf = &j.info[img];
    const int total = inf->n_boxes, m = min(total, j.cap);
    const int32_t* cand = j.cand + (int64_t)img * j.cap * 3;
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        const int root = cand[3 * k];
        int rank = 0;
        for (int q = 0; q < m; ++q) rank += cand[3 * q] < root;
        if (rank < max_boxes) {
            LfaLineBox b; b.minr = cand[3 * k + 1]; b.maxr = cand[3 * k + 2]; b.minc = 0; b.maxc = j.W;
            out[(int64_t)img * max_boxes + rank] = b;
        }
    }
    if (threadIdx.x == 0) {
        counts[img] = min(total, max_boxes);
        if (total > max_boxes || total > j.cap) atomicOr(&inf->status, 1);
    }
}

hipError_t launch_lines(const LinesJob& j, LfaLineBox* out, int32_t* counts, int max_boxes, hipStream_t s) {
    const int64_t npix = (int64_t)j.H * j.W;
    int bpi = (int)((npix + 256 * 16 - 1) / (256 * 16));
    const int want = (4096 + j.n - 1) / j.n;
    if (bpi > want) bpi = want;
    if (bpi < 1) bpi = 1;
    const dim3 grid(bpi, j.n), blk(256);
    LFA_LAUNCH("k_lines_init", k_lines_init, dim3(4, j.n), blk, 0, s, j);
    {
        const dim3 fgrid((j.W + kFOC - 1) / kFOC, (j.H + kFTH - 1) / kFTH, j.n);
        if (j.dtype == LFA_U16) LFA_LAUNCH("k_lines_front", k_lines_front<uint16_t>, fgrid, dim3(kFIW), 0, s, j);
        else LFA_LAUNCH("k_lines_front", k_lines_front<uint8_t>, fgrid, dim3(kFIW), 0, s, j);
    }
    LFA_LAUNCH("k_lines_binlut", k_lines_binlut, dim3(16, j.n), blk, 0, s, j);
    LFA_LAUNCH("k_clahe_hist", k_clahe_hist, dim3(j.nty * j.ntx * kTileSlices, j.n), blk, 0, s, j);
    LFA_LAUNCH("k_clahe_tiles", k_clahe_tiles, dim3(j.nty * j.ntx, j.n), blk, 0, s, j);
    // rows per blend block: enough blocks to fill the chip (four per CU), then as many rows as possible -- a block ends with
    // one global add per level it has seen, so fewer, taller patches spend less on that
    const int ncb = (j.W + 255) / 256;
    const int brows = (int)min((int64_t)kBlendRows, max((int64_t)16, (int64_t)j.H * ncb * j.n / 1024));
    LFA_LAUNCH("k_clahe_blend", k_clahe_blend, dim3(ncb * ((j.H + brows - 1) / brows), j.n), blk, 0, s, j, brows);
    LFA_LAUNCH("k_lines_levels", k_lines_levels, dim3(j.n), dim3(1024), 0, s, j);
    LFA_LAUNCH("k_ccl_init", k_ccl_init, grid, blk, 0, s, j);
    LFA_LAUNCH("k_ccl_merge", k_ccl_merge, grid, blk, 0, s, j);
    LFA_LAUNCH("k_ccl_boxes", k_ccl_boxes, grid, blk, 0, s, j);
    LFA_LAUNCH("k_ccl_collect", k_ccl_collect, grid, blk, 0, s, j);
    LFA_LAUNCH("k_lines_emit", k_lines_emit, dim3(j.n), blk, 0, s, j, out, counts, max_boxes);
    return hipGetLastError();
}

}  // namespace lfa
