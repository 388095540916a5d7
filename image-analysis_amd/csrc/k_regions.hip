// k_regions.hip -- stage S7: region test, duplicate rejection, outlier removal, sort, analysis windows  (gfx950)
//
// Reference: /root/reference/LFA-Image_Analysis/detection.py:67-93 (loop over regionprops in label order),
//            :118-127 (check_duplicate), :138-153 (remove_outliers), :215-216 (analysis window of a strip).
// Means are exact integer sums followed by the reference's float64 divisions (np.mean of an integer array sums in
// float64, which is exact below 2^53).
//
// The box of a candidate is exactly the crop detect returns for it, and image_analysis(crop, False) profiles a
// window of that crop.  So while k_box_sums reads a box for the mean test it also writes the integer row sums of the
// analysis window into a per-image pool; the profile stage of lfa_analyze_batch then needs no second read of the pixels.
#include "lfa_internal.h"
#include "lfa_device.h"
#include "lfa_regions.h"
#include "lfa_algo.h"

namespace lfa {

constexpr int kMaxStripsLds = 1024;

// Integer sums of the ORIGINAL image over the expanded, clipped box of every candidate (per channel for colour).
// One block per candidate.  Rows are read as aligned dwords (a row of a strip is ~170 bytes) and every (row, dword)
// of the box is an independent item, so each thread keeps several loads in flight; the channel of a byte is its offset
// in the image modulo 3.  The rows of the analysis window are also summed (all channels together) into LDS and
// written to the row-sum pool.
__global__ __launch_bounds__(256) void k_box_sums(Job j) {
    const int img = blockIdx.y;
    const int64_t rbase = (int64_t)img * j.run_cap;
    const int64_t cbase = (int64_t)img * j.cand_cap;
    // The sums feed only the test `mean(box) > labeled[0,0]` (detection.py:67, :81).  When pixel (0,0) is background that
    // is `mean > 0`, which always holds: the box contains the component, and every pixel of a component is above a
    // non-negative threshold, hence non-zero.  So the box pass runs only for images whose corner pixel is foreground.
    if (j.info[img].label00 == 0) return;
    const int nc = j.info[img].n_candidates;
    if (j.dtype == LFA_F64) {
        // float64 images: np.mean of the box is a float sum, whose rounding depends on numpy's order.  2-D box (a strided view
        // reduced over both axes): numpy copies the view in C order into buffers of 8192 elements, sums each with its pairwise
        // sum and adds the buffers up in turn; colour box reduced over (0, 1): one running sum per channel over the pixels in
        // raster order.  (Checked against numpy 2.2 on random boxes, tests/test_cpu_boundary.py; the sums travel to k_finalize
        // as bit patterns in cand_sum.)  One thread per candidate: a rare case -- only images whose corner pixel is foreground.
        if (threadIdx.x != 0) return;
        const double* fb = (const double*)(j.images + (int64_t)img * j.img_bytes);
        for (int ci = blockIdx.x; ci < nc; ci += gridDim.x) {
            const Box bx = cand_box(j, rbase, j.root_of_label[rbase + j.cand[cbase + ci] - 1]);
            const int w = bx.cc - bx.minc, h = bx.cr - bx.minr;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            if (j.C == 1) {
                const int64_t n = (int64_t)h * w;
                const double* org = fb + (int64_t)bx.minr * j.W + bx.minc;
                const int W = j.W;
                auto at = [org, w, W](int i) -> double { const int y = i / w; return org[(int64_t)y * W + (i - y * w)]; };
                for (int64_t o = 0; o < n; o += 8192) {
                    const int m = (int)(n - o < 8192 ? n - o : 8192);
                    const int o32 = (int)o;
                    a0 = a0 + np_pairwise([&](int i) { return at(o32 + i); }, 0, m);
                }
            } else {
                for (int y = bx.minr; y < bx.cr; ++y)
                    for (int x = bx.minc; x < bx.cc; ++x) {
                        const double* px = fb + ((int64_t)y * j.W + x) * 3;
                        a0 = a0 + px[0]; a1 = a1 + px[1]; a2 = a2 + px[2];
                    }
            }
            unsigned long long* cs = j.cand_sum + (cbase + ci) * 3;
            cs[0] = (unsigned long long)__double_as_longlong(a0); cs[1] = (unsigned long long)__double_as_longlong(a1);
            cs[2] = (unsigned long long)__double_as_longlong(a2);
        }
        return;
    }
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    __shared__ unsigned long long part[4][3];
    __shared__ unsigned int wrow[kWinRowsLds];                    // window row sums of this candidate
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const int esz = (j.dtype == LFA_U16) ? 2 : 1;
    const int64_t pxb = (int64_t)esz * j.C;                       // bytes per pixel
    const int64_t row_bytes = (int64_t)j.W * pxb;
    const bool dwords = j.aligned16 != 0 && (row_bytes & 3) == 0;
    for (int ci = blockIdx.x; ci < nc; ci += gridDim.x) {
        const Box bx = cand_box(j, rbase, j.root_of_label[rbase + j.cand[cbase + ci] - 1]);
        const int minr = bx.minr, minc = bx.minc, maxr = bx.cr, maxc = bx.cc;
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (dwords) {
            const int rs_off = j.cand_rs_off[cbase + ci];
            const int h = maxr - minr, w = maxc - minc;
            const int wr0 = (int)((double)h * .25), wr1 = (int)((double)h * .6);    // window rows, relative to minr
            const int64_t b0 = ((int64_t)minr * j.W + minc) * pxb;            // first byte of the box's first row
            const int64_t nbytes = (int64_t)w * pxb;
            const int64_t bw0 = b0 + (int64_t)((int)((double)w * .3)) * pxb;  // window bytes of that row
            const int64_t bw1 = b0 + (int64_t)((int)((double)w * .7)) * pxb;
            const bool want_rows = rs_off >= 0;
            if (want_rows) { for (int k = t; k < wr1 - wr0; k += 256) wrow[k] = 0; __syncthreads(); }
            const int64_t a0 = b0 & ~(int64_t)3;                               // row_bytes % 4 == 0: same phase on every row
            const int ndw = (int)(((b0 + nbytes + 3) & ~(int64_t)3) - a0) >> 2;
            const int items = h * ndw;
            const int ph0 = (int)(a0 % 3);                                     // row_bytes % 3 == 0 for colour: same on every row
            unsigned int u0 = 0, u1 = 0, u2 = 0;
            int flush = 0;
#pragma unroll 8
            for (int it = t; it < items; it += 256) {
                const int rr = it / ndw, d = it - rr * ndw;
                const int64_t a = a0 + 4 * (int64_t)d;
                unsigned int x = *(const unsigned int*)(base + (int64_t)rr * row_bytes + a);
                unsigned int vm = 0xffffffffu;
                if (a < b0) vm &= 0xffffffffu << (8 * (int)(b0 - a));
                if (a + 4 > b0 + nbytes) vm &= 0xffffffffu >> (8 * (int)(a + 4 - (b0 + nbytes)));
                x &= vm;
                if (j.C == 3) {
                    const int ph = (ph0 + d) % 3;                              // (a0 + 4d) % 3, since 4 = 1 (mod 3)
                    const unsigned int m0 = ph == 0 ? 0xff0000ffu : (ph == 1 ? 0x00ff0000u : 0x0000ff00u);
                    const unsigned int m1 = ph == 0 ? 0x0000ff00u : (ph == 1 ? 0xff0000ffu : 0x00ff0000u);
                    const unsigned int m2 = ph == 0 ? 0x00ff0000u : (ph == 1 ? 0x0000ff00u : 0xff0000ffu);
                    u0 = __builtin_amdgcn_sad_u8(x & m0, 0u, u0);
                    u1 = __builtin_amdgcn_sad_u8(x & m1, 0u, u1);
                    u2 = __builtin_amdgcn_sad_u8(x & m2, 0u, u2);
                } else if (esz == 1) {
                    u0 = __builtin_amdgcn_sad_u8(x, 0u, u0);
                } else {
                    u0 += (x & 0xffffu) + (x >> 16);
                }
                if (++flush == 8192) { s0 += u0; s1 += u1; s2 += u2; u0 = u1 = u2 = 0; flush = 0; }   // keep 32-bit partials exact
                if (want_rows && rr >= wr0 && rr < wr1 && a + 4 > bw0 && a < bw1) {
                    unsigned int wm = 0xffffffffu;
                    if (a < bw0) wm &= 0xffffffffu << (8 * (int)(bw0 - a));
                    if (a + 4 > bw1) wm &= 0xffffffffu >> (8 * (int)(a + 4 - bw1));
                    const unsigned int y = x & wm;
                    const unsigned int sw = (esz == 1) ? __builtin_amdgcn_sad_u8(y, 0u, 0u) : ((y & 0xffffu) + (y >> 16));
                    if (sw) atomicAdd(&wrow[rr - wr0], sw);
                }
            }
            s0 += u0; s1 += u1; s2 += u2;
            if (want_rows) {
                __syncthreads();
                uint32_t* dst = j.rowsum + (int64_t)img * j.rowsum_cap + rs_off;
                for (int k = t; k < wr1 - wr0; k += 256) dst[k] = wrow[k];
            }
        } else {
            for (int r = minr + wv; r < maxr; r += 4) {
                const int64_t rowpx = (int64_t)r * j.W;
                for (int c = minc + lane; c < maxc; c += 64) {
                    const int64_t px = rowpx + c;
                    if (j.C == 3 && esz == 2) { const uint16_t* b16 = (const uint16_t*)base; s0 += b16[3 * px]; s1 += b16[3 * px + 1]; s2 += b16[3 * px + 2]; }
                    else if (j.C == 3) { s0 += base[3 * px]; s1 += base[3 * px + 1]; s2 += base[3 * px + 2]; }
                    else if (esz == 1) s0 += base[px];
                    else s0 += ((const uint16_t*)base)[px];
                }
            }
        }
        const unsigned long long w0 = wave_sum_u64(s0), w1 = wave_sum_u64(s1), w2 = wave_sum_u64(s2);
        if (lane == 0) { part[wv][0] = w0; part[wv][1] = w1; part[wv][2] = w2; }
        __syncthreads();
        if (t < 3) j.cand_sum[(cbase + ci) * 3 + t] = part[0][t] + part[1][t] + part[2][t] + part[3][t];
        __syncthreads();
    }
}

struct Acc { int minr, maxr, minc, maxc, area, label, rs_off; };

// one wave per image: accept in label order with the inclusive-overlap test, drop wide outliers, stable sort
__global__ __launch_bounds__(64) void k_finalize(Job j, OutPtrs o) {
    const int img = blockIdx.x;
    const int64_t rbase = (int64_t)img * j.run_cap;
    LfaImageInfo* inf = &j.info[img];
    const int nc = inf->n_candidates;
    const int lane = threadIdx.x;
    const int cap = min(j.max_strips, kMaxStripsLds);
    // accepted / kept lists: dynamic LDS sized to the call's max_strips.  This wave runs beside the other chunk stream's
    // image passes, which leave little LDS free on a CU: with the lists at their 1024-entry maximum (56 KiB) the block
    // waited for several of those blocks to retire (measured: 1.5 % of the whole pipeline).
    extern __shared__ __align__(16) unsigned char fin_smem[];
    Acc* acc = (Acc*)fin_smem;
    Acc* kept = acc + cap;
    int na = 0;
    int status = 0;
    const int64_t cbase = (int64_t)img * j.cand_cap;
    const double bg = (double)inf->label00;                   // detection.py:67: the label id at (0,0)
    // The accept loop is sequential (a candidate is tested against everything accepted before it), but what it reads is
    // not: the lanes fetch label, root, box, area and the region test of kStage candidates at once into LDS -- three dependent
    // global loads per candidate that the loop would otherwise wait for one candidate at a time (images with hundreds
    // of candidates: 120 us -> a few) -- and the loop then runs on LDS.
    constexpr int kStage = 256;
    __shared__ int s_box[kStage][4];
    __shared__ int s_area[kStage], s_rs[kStage], s_lab[kStage];
    __shared__ unsigned char s_ok[kStage];
    for (int c0 = 0; c0 < nc && !status; c0 += kStage) {
        const int m = min(kStage, nc - c0);
        for (int k = lane; k < m; k += 64) {
            const int ci = c0 + k;
            const int lab = j.cand[cbase + ci];
            const int root = j.root_of_label[rbase + lab - 1];
            const Box bx = cand_box(j, rbase, root);
            bool ok = true;
            if (inf->label00 != 0) {   // np.mean of the clipped box (detection.py:76-81); always > 0 when the corner is background
                // exact integer sums, then the reference's divisions
                const double cnt = (double)((long long)(bx.cr - bx.minr) * (long long)(bx.cc - bx.minc));
                const unsigned long long* cs = j.cand_sum + (cbase + ci) * 3;
                double mean;
                const bool f64 = j.dtype == LFA_F64;                  // (float images: the sums are float64 bit patterns, k_box_sums)
                const double s0 = f64 ? __longlong_as_double((long long)cs[0]) : (double)cs[0];
                if (j.C == 3) {
                    const double s1 = f64 ? __longlong_as_double((long long)cs[1]) : (double)cs[1];
                    const double s2 = f64 ? __longlong_as_double((long long)cs[2]) : (double)cs[2];
                    const double m0 = s0 / cnt, m1 = s1 / cnt, m2 = s2 / cnt;
                    mean = ((m0 + m1) + m2) / 3.0;
                } else {
                    mean = s0 / cnt;
                }
                ok = mean > bg;
            }
            s_box[k][0] = bx.minr; s_box[k][1] = bx.maxr; s_box[k][2] = bx.minc; s_box[k][3] = bx.maxc;
            s_area[k] = j.area[rbase + root]; s_rs[k] = j.cand_rs_off[cbase + ci];     // for the accepted ones: fetched here, not one by one in the loop
            s_lab[k] = lab; s_ok[k] = ok ? 1 : 0;
        }
        __syncthreads();
        for (int k = 0; k < m; ++k) {                          // uniform: every lane walks the same candidates
            if (!s_ok[k]) continue;
            const int minr = s_box[k][0], maxr = s_box[k][1], minc = s_box[k][2], maxc = s_box[k][3];
            bool hit = false;
            for (int q = lane; q < na; q += 64) {
                const Acc a = acc[q];
                hit |= (minr <= a.maxr && maxr >= a.minr && minc <= a.maxc && maxc >= a.minc);
            }
            if (__any(hit)) continue;
            if (na < cap) {
                if (lane == 0) {
                    Acc a; a.minr = minr; a.maxr = maxr; a.minc = minc; a.maxc = maxc;
                    a.area = s_area[k]; a.label = s_lab[k]; a.rs_off = s_rs[k];
                    acc[na] = a;
                }
                __syncthreads();
                ++na;
            } else {
                status |= 1;                                  // out of room: later duplicate tests would be incomplete
                ++na;
                break;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const int na_c = min(na, cap);
    // remove_outliers: width uses the expanded maxc (already applied)
    int wmin = 0x7fffffff;
    for (int k = lane; k < na_c; k += 64) wmin = min(wmin, acc[k].maxc - acc[k].minc);
    wmin = wave_min_i32(wmin);
    __shared__ int nk_s;
    if (lane == 0) {
        int nk = 0;
        for (int k = 0; k < na_c; ++k)
            if ((long long)(acc[k].maxc - acc[k].minc) <= 2ll * wmin) kept[nk++] = acc[k];
        nk_s = nk;
    }
    __syncthreads();
    const int nk = nk_s;
    // stable sort by (minc, maxc): rank by counting
    LfaStrip* out = o.strips + (int64_t)img * j.max_strips;
    LfaRoi* rois = o.rois ? o.rois + (int64_t)img * j.max_strips : nullptr;
    for (int i = lane; i < nk; i += 64) {
        const Acc a = kept[i];
        int rank = 0;
        for (int k = 0; k < nk; ++k) {
            const Acc b = kept[k];
            const bool less = (b.minc < a.minc) || (b.minc == a.minc && (b.maxc < a.maxc || (b.maxc == a.maxc && k < i)));
            rank += less ? 1 : 0;
        }
        LfaStrip sref;
        sref.minr = a.minr; sref.maxr = a.maxr; sref.minc = a.minc; sref.maxc = a.maxc; sref.area = a.area; sref.label = a.label;
        out[rank] = sref;
        if (rois) {
            // window of image_analysis(strip, selected=False): detection.py:215-216 on the numpy-clipped crop
            const int h = min(a.maxr, j.H) - a.minr, w = min(a.maxc, j.W) - a.minc;
            LfaRoi r;
            r.image = img;
            r.r0 = a.minr + (int)((double)h * .25); r.r1 = a.minr + (int)((double)h * .6);
            r.c0 = a.minc + (int)((double)w * .3);  r.c1 = a.minc + (int)((double)w * .7);
            rois[rank] = r;
            if (j.plane_stats && r.r1 > r.r0) atomicMax(&j.plane_stats[4], (uint32_t)(r.r1 - r.r0));
            j.slot_rs_off[(int64_t)img * j.max_strips + rank] = a.rs_off;
        }
    }
    if (lane == 0) {
        inf->n_accepted = na; inf->n_strips = nk; inf->status |= status;
        o.counts[img] = nk;
        if (o.info) o.info[img] = *inf;
    }
}

hipError_t launch_regions(const Job& j, const OutPtrs& o, hipStream_t s) {
    // one block per candidate, blocks loop when there are more.  Almost every call needs no box sum at all (the kernel's blocks
    // return at once unless pixel (0,0) of their image is foreground), and then the launch costs what its blocks take to be
    // dispatched through a chip that is busy with another lane's image pass: 8192 blocks 130 us, in the lane's serial chain
    // (kernel timeline of configs[2], round 3) -- so about a thousand blocks, not eight
    int gx = (1024 + j.n - 1) / j.n;
    if (gx < 2) gx = 2;
    if (gx > 256) gx = 256;
    LFA_LAUNCH("k_box_sums", k_box_sums, dim3(gx, j.n), dim3(256), 0, s, j);
    const size_t fin_lds = (size_t)2 * (j.max_strips < kMaxStripsLds ? j.max_strips : kMaxStripsLds) * sizeof(Acc);
    LFA_LAUNCH("k_finalize", k_finalize, dim3(j.n), dim3(64), fin_lds, s, j, o);
    return hipGetLastError();
}

}  // namespace lfa
