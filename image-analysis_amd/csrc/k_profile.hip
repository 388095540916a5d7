// k_profile.hip -- stages S8-S9: per-ROI row-mean profiles (segmented reduction) and top-2 peaks  (gfx950)
//
// Reference: /root/reference/LFA-Image_Analysis/detection.py:205-245 (image_analysis), :247-281
// (inverse_image_analysis), :283-316 (analyze_peaks).  profile[k] = np.nanmean(window, axis=cols(+channels))[k]; for the
// integer images this path handles that is an exact integer row sum and one float64 division.  The peak logic is the
// shared sequential code of lfa_algo.h (bit-for-bit numpy summation order) run by a group of 16 or 64 lanes with a
// group-parallel first-extremum search.
#include <algorithm>
#include <cstdlib>

#include "lfa_internal.h"
#include "lfa_algo.h"
#include "lfa_device.h"

#ifdef LFA_PEAKS_CLOCK
__device__ unsigned long long lfa_ck[16 * 1024];
extern "C" int lfa_debug_clocks(unsigned long long* out16) {       // dev aid: read and clear the phase clocks
    static unsigned long long h[16 * 1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(lfa_ck), sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < 16; ++k) { out16[k] = 0; for (int b = 0; b < 1024; ++b) out16[k] += h[k * 1024 + b]; }
    for (auto& v : h) v = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(lfa_ck), h, sizeof h) == hipSuccess ? 0 : -1;
}
#endif

namespace lfa {

// np.argmax / np.argmin over p[0..n): first extremum, first NaN wins.  Called by the G lanes of one group (G = 16 or 64
// consecutive lanes of a wave) with identical arguments; other groups of the wave may run other profiles.
// The walk from the extremum to a foot of the peak (lfa_algo.h walk_goes_on), G positions per step: lane k of the group
// tests position from + dir*(step*G + k); the first position where the walk stops is the answer.  Every lane of the group
// calls this with identical arguments and gets the same result.
template <int G>
struct GroupWalk {
    __device__ __forceinline__ int operator()(const double* p, int n, int from, int dir, double thr, bool inverse) const {
        const int lane = threadIdx.x & 63, lg = lane & (G - 1), gshift = lane & ~(G - 1);
        for (int base = from;; base += dir * G) {
            const int i = base + dir * lg;
            const bool inside = i >= 0 && i < n;
            const bool stop = !inside || !walk_goes_on(p, n, i, dir, thr, inverse);
            const unsigned long long bal = __ballot(stop);
            const unsigned long long mine = (G == 64) ? bal : ((bal >> gshift) & ((1ull << (G & 63)) - 1ull));
            if (mine) return base + dir * (int)__builtin_ctzll(mine);
        }
    }
};

// np.argmax / np.argmin over p[0..n): first extremum, first NaN wins.  Every lane scans the elements lg, lg + G, lg + 2G, ...
// -- eight LDS reads requested before the first compare, the compares written as selects (round 3: one read per iteration,
// each waited for, and a branch per condition made the three searches of a profile the largest part of the kernel) -- and
// keeps (value, index, is-NaN) of its first extremum; a NaN, once met, stays.  Then over the group: the extreme value
// (group_ext_f64), and the smallest index among the lanes that hold it (group_min_i32); if any lane met a NaN, the smallest
// index among those instead.  Two reductions on the DPP path in place of log2(G) rounds of four ds_bpermute and a compare tree.
template <int G>
struct GroupArgExt {
    __device__ __forceinline__ int operator()(const double* p, int n, bool want_min) const {
        const int lg = threadIdx.x & (G - 1);
        constexpr int kNone = 0x7fffffff;
        double m = 0.0; int idx = kNone; int isn = 0;
        constexpr int U = 8;
        for (int i0 = lg; i0 < n; i0 += G * U) {
            double x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int i = i0 + u * G; x[u] = p[i < n ? i : i0]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * G;
                const int better = want_min ? !(x[u] >= m) : !(x[u] <= m);            // also true for a NaN
                const int upd = (i < n) & (isn ^ 1) & ((idx == kNone) | better);
                m = upd ? x[u] : m;
                idx = upd ? i : idx;
                isn |= upd & (int)(x[u] != x[u]);
            }
        }
        const int first_nan = group_min_i32<G>(isn ? idx : kNone);
        const int plain = (idx != kNone) & (isn ^ 1);
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        const double mine = plain ? m : (want_min ? inf : -inf);
        const double ext = group_ext_f64<G>(mine, want_min);
        const int best = group_min_i32<G>((plain & (int)(mine == ext)) ? idx : kNone);
        return first_nan != kNone ? first_nan : best;
    }
};

__device__ inline void store_peaks(LfaPeaks* dst, const TwoPeaks& t, int length) {
    LfaPeaks r;
    r.n_peaks = t.n; r.length = length;
    for (int k = 0; k < 2; ++k) {
        const bool have = k < t.n;
        r.start[k] = have ? t.pk[k].start : -1;
        r.end[k] = have ? t.pk[k].end : -1;
        r.area[k] = have ? t.pk[k].area : 0.0;
    }
    *dst = r;
}

__device__ inline void store_bad(LfaPeaks* dst, int len) {
    LfaPeaks r; r.n_peaks = -1; r.length = len; r.start[0] = r.start[1] = r.end[0] = r.end[1] = -1;
    r.area[0] = r.area[1] = 0.0;
    *dst = r;
}

// bytes [b0, b0+nbytes) of a row, dword at byte offset a: sum of the pixels inside (uint8: bytes, uint16: halves)
__device__ inline unsigned int masked_sum(unsigned int x, int64_t a, int64_t b0, int64_t nbytes, int esz, unsigned int acc) {
    unsigned int vm = 0xffffffffu;
    if (a + 4 <= b0 || a >= b0 + nbytes) return acc;
    if (a < b0) vm &= 0xffffffffu << (8 * (int)(b0 - a));
    if (a + 4 > b0 + nbytes) vm &= 0xffffffffu >> (8 * (int)(a + 4 - (b0 + nbytes)));
    const unsigned int y = x & vm;
    return (esz == 1) ? __builtin_amdgcn_sad_u8(y, 0u, acc) : (acc + (y & 0xffffu) + (y >> 16));
}

// same with a point-operation table applied to every pixel first (lfa_profile_lut_batch): l8 = 256 bytes in LDS, l16 =
// 65536 halves in global memory
__device__ inline unsigned int masked_lut_sum(unsigned int x, int64_t a, int64_t b0, int64_t nbytes, int esz, unsigned int acc,
                                              const uint8_t* l8, const uint16_t* l16) {
    if (a + 4 <= b0 || a >= b0 + nbytes) return acc;
    if (esz == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t pos = a + k;
            if (pos >= b0 && pos < b0 + nbytes) acc += l8[(x >> (8 * k)) & 0xffu];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t pos = a + 2 * k;
            if (pos >= b0 && pos < b0 + nbytes) acc += l16[(x >> (16 * k)) & 0xffffu];
        }
    }
    return acc;
}

// G lanes per ROI (16: four ROIs per wave; 64: one), blockDim.x / G ROIs per block.  The peak search is sequential per
// profile, so the parallelism is across ROIs; each ROI has `lstride` float64 slots of LDS.
//   pooled: lfa_analyze_batch -- k_box_sums already wrote the window's integer row sums, no pixel is read here;
//   wide:   the ROI's rows are read as aligned 16-byte (or 4-byte) pieces, every (row, piece) an independent item with
//           several loads in flight per lane; integer row sums accumulate in LDS as exact float64 (ds_add_f64);
//   plain:  lanes stride over a row's elements (unaligned batches, or profiles longer than the LDS slots: then the
//           peak search reads the profile from global memory).
// In every mode profile[r] = integer row sum / element count in float64.
// LUT: every pixel goes through a point-operation table first (luts: n_luts tables of 256 u8 / 65536 u16; roi_lut: table
// index per ROI or null = table 0); the uint8 table of a ROI is copied to LDS behind the accumulators.
template <int G, bool LUT>
__global__ __launch_bounds__(256) void k_profile(Job j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots,
                                                 const int32_t* rs_off, int inverse, int rng, double* profiles, int stride,
                                                 LfaPeaks* peaks, int lstride, const uint8_t* luts, const int32_t* roi_lut,
                                                 int n_luts, const int32_t* skip_if_sorted) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (skip_if_sorted && skip_if_sorted[0] == 0) return;        // sorted ROI array: k_profile_band has done the profiles (launch_profiles)
    const int rpb = blockDim.x / G;                              // ROIs per block
    const int slot = threadIdx.x / G, lg = threadIdx.x & (G - 1);
    double* sp = (double*)smem + (int64_t)slot * lstride;
    const int ri = blockIdx.x * rpb + slot;
    bool valid = ri < n_rois;
    if (valid && counts && (ri % slots) >= counts[ri / slots]) valid = false;   // analyze path: slot beyond this image's strips
    LfaRoi roi; roi.image = 0; roi.r0 = roi.r1 = roi.c0 = roi.c1 = 0;
    if (valid) roi = rois[ri];
    const int len = roi.r1 - roi.r0;
    const int li = (LUT && valid && roi_lut) ? roi_lut[ri] : 0;
    const bool bad = valid && (roi.image < 0 || roi.image >= j.n || roi.r0 < 0 || roi.r1 > j.H || roi.c0 < 0 || roi.c1 > j.W ||
                               len < 0 || roi.c1 < roi.c0 || len > stride || li < 0 || (LUT && li >= n_luts));
    uint8_t* l8 = smem + (int64_t)rpb * lstride * 8 + slot * 256;             // LUT, uint8 images only
    const uint16_t* l16 = (const uint16_t*)luts + (int64_t)(bad ? 0 : li) * 65536;
    if (LUT && j.dtype != LFA_U16 && valid && !bad)
        for (int k = lg; k < 256; k += G) l8[k] = luts[(int64_t)li * 256 + k];
    const bool work = valid && !bad;
    double* prof = profiles + (int64_t)(valid ? ri : 0) * stride;
    const uint8_t* base = j.images + (int64_t)(work ? roi.image : 0) * j.img_bytes;
    const int C = j.C;
    const int esz = (j.dtype == LFA_U16) ? 2 : 1;
    const int ne = (roi.c1 - roi.c0) * C;                 // elements per row
    const double cnt = (double)ne;
    const bool use_lds = lstride > 0 && len <= lstride;
    const int64_t row_bytes = (int64_t)j.W * C * esz;
    const int pooled = (work && rs_off && use_lds) ? rs_off[ri] : -1;
    const bool wide = work && pooled < 0 && use_lds && j.aligned16 && (row_bytes & 3) == 0 && ne > 0;
    if (wide) for (int r = lg; r < len; r += G) sp[r] = 0.0;
    __syncthreads();
    if (pooled >= 0) {
        const uint32_t* rsum = j.rowsum + (int64_t)roi.image * j.rowsum_cap + pooled;
        for (int r = lg; r < len; r += G) {
            prof[r] = (double)rsum[r] / cnt;
        }
    } else if (wide) {
        const int64_t b0 = ((int64_t)roi.r0 * j.W + roi.c0) * C * esz;       // first byte of row 0 of the ROI
        const int64_t nbytes = (int64_t)ne * esz;
        if ((row_bytes & 15) == 0) {
            const int64_t a0 = b0 & ~(int64_t)15;                              // row_bytes % 16 == 0: same phase on every row
            const int nch = (int)(((b0 + nbytes + 15) & ~(int64_t)15) - a0) >> 4;
            const int items = len * nch;
            const int qg = G / nch, mg = G - qg * nch;                         // (row, piece) advance by G items without a division
            int r = lg / nch, d = lg - r * nch;
            // Only the first and the last piece of a row reach outside the window: their byte masks are the same on
            // every row, so they are built once per ROI (first: bytes before b0 off; last: bytes from b0 + nbytes on off)
            // and an item costs two selects, an AND and a v_sad_u8 per dword.
            uint32_t mfirst[4], mlast[4];
            {
                const int lo = (int)(b0 - a0);                                 // 0..15 bytes to drop at the front of piece 0
                const int hi = (int)(b0 + nbytes - (a0 + 16 * (int64_t)(nch - 1)));   // 1..16 bytes to keep of the last piece
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int l = min(max(lo - 4 * k, 0), 4), h = min(max(hi - 4 * k, 0), 4);
                    mfirst[k] = l >= 4 ? 0u : (0xffffffffu << (8 * l));
                    mlast[k] = h >= 4 ? 0xffffffffu : ~(0xffffffffu << (8 * h));
                }
            }
            constexpr int U = LUT ? 4 : 8;                                     // loads in flight per lane
            for (int it0 = lg; it0 < items; it0 += G * U) {
                uint4 x[U];
                int rr[U], dd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    rr[u] = r; dd[u] = d;
                    if (it0 + G * u < items) x[u] = *(const uint4*)(base + (int64_t)r * row_bytes + a0 + 16 * (int64_t)d);
                    else x[u] = make_uint4(0u, 0u, 0u, 0u);
                    r += qg; d += mg;
                    if (d >= nch) { d -= nch; ++r; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (it0 + G * u >= items) break;
                    unsigned int sv;
                    if (LUT) {
                        const int64_t a = a0 + 16 * (int64_t)dd[u];
                        sv = masked_lut_sum(x[u].x, a, b0, nbytes, esz, 0u, l8, l16);
                        sv = masked_lut_sum(x[u].y, a + 4, b0, nbytes, esz, sv, l8, l16);
                        sv = masked_lut_sum(x[u].z, a + 8, b0, nbytes, esz, sv, l8, l16);
                        sv = masked_lut_sum(x[u].w, a + 12, b0, nbytes, esz, sv, l8, l16);
                    } else {
                        const bool f = dd[u] == 0, l = dd[u] == nch - 1;
                        const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                        sv = 0u;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t y = w[k] & (f ? mfirst[k] : 0xffffffffu) & (l ? mlast[k] : 0xffffffffu);
                            sv = (esz == 1) ? __builtin_amdgcn_sad_u8(y, 0u, sv) : (sv + (y & 0xffffu) + (y >> 16));
                        }
                    }
                    if (sv) unsafeAtomicAdd(&sp[rr[u]], (double)sv);
                }
            }
        } else {
            const int64_t a0 = b0 & ~(int64_t)3;                               // row_bytes % 4 == 0: same phase on every row
            const int ndw = (int)(((b0 + nbytes + 3) & ~(int64_t)3) - a0) >> 2;
            const int items = len * ndw;
            const int qg = G / ndw, mg = G - qg * ndw;
            int r = lg / ndw, d = lg - r * ndw;
            for (int it0 = lg; it0 < items; it0 += G * 8) {
                unsigned int x[8];
                int rr[8], dd[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    rr[u] = r; dd[u] = d;
                    x[u] = (it0 + G * u < items) ? *(const unsigned int*)(base + (int64_t)r * row_bytes + a0 + 4 * (int64_t)d) : 0u;
                    r += qg; d += mg;
                    if (d >= ndw) { d -= ndw; ++r; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (it0 + G * u >= items) break;
                    const unsigned int sv = LUT ? masked_lut_sum(x[u], a0 + 4 * (int64_t)dd[u], b0, nbytes, esz, 0u, l8, l16)
                                                : masked_sum(x[u], a0 + 4 * (int64_t)dd[u], b0, nbytes, esz, 0u);
                    if (sv) unsafeAtomicAdd(&sp[rr[u]], (double)sv);
                }
            }
        }
    } else if (work) {
        for (int r = 0; r < len; ++r) {                     // G lanes share a row
            const int64_t e0 = ((int64_t)(roi.r0 + r) * j.W + roi.c0) * C;
            unsigned long long sm = 0;
            if (esz == 1) { for (int k = lg; k < ne; k += G) sm += LUT ? l8[base[e0 + k]] : base[e0 + k]; }
            else { const uint16_t* b16 = (const uint16_t*)base; for (int k = lg; k < ne; k += G) sm += LUT ? l16[b16[e0 + k]] : b16[e0 + k]; }
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            if (lg == 0) {
                const double m = (double)sm / cnt;        // ne == 0 -> 0/0 = NaN, like np.nanmean of an empty slice
                prof[r] = m;
            }
        }
    }
    __syncthreads();
    if (wide)
        for (int r = lg; r < len; r += G) {
            prof[r] = sp[r] / cnt;
        }
    if (bad && lg == 0) store_bad(&peaks[ri], len);               // k_peaks skips ROIs it finds bad by the same test
}

// ---------------------------------------------------------------------------------------------------------------
// Dense ROI sets (BASELINE configs[4]: hundreds of ROIs per image, together covering the image about once, overlapping):
// one block owns a band of RB rows of one image and serves EVERY ROI that crosses the band, so each image byte is
// fetched once per band -- coalesced, whole rows, staged in LDS -- instead of once per ROI in 48..192-byte row pieces
// (k_profile above: 1.6x line over-fetch and every overlap fetched again).  Needs the ROI array sorted by image
// (k_roi_index checks that and writes the first ROI of every image); a band crossed by few ROI rows skips the staging and
// reads the row pieces from global memory.  profile[r] = exact integer row sum / element count, one (ROI, row) per
// 16-lane group, so no accumulation order and no atomics are involved.
// ---------------------------------------------------------------------------------------------------------------

// idx[0] (cleared by the launcher beforehand) becomes 1 iff rois[].image somewhere DEcreases; idx[1 + k] = first ROI with image
// >= k, k = 0..n: thread i writes the entries of the images (image[i-1], image[i]] -- each entry has exactly one writer when the
// array is sorted, and nobody reads them when it is not.  (One 1024-thread block walked the whole array in rounds 1-2: 47 us
// per call of 65 536 ROIs, 7 % of configs[4]'s step.)
__global__ __launch_bounds__(256) void k_roi_index(const LfaRoi* rois, int n_rois, int n_images, int32_t* idx) {
    int unsorted = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rois; i += gridDim.x * blockDim.x) {
        const int cur = rois[i].image;
        const int prev = i ? rois[i - 1].image : -1;
        if (cur < prev) unsorted = 1;
        // images (prev, cur] start at i (clipped to the images there are)
        int lo = prev + 1 < 0 ? 0 : prev + 1, hi = cur < n_images ? cur : n_images;
        for (int k = lo; k <= hi; ++k) idx[1 + k] = i;
        if (i == n_rois - 1)
            for (int k = (cur + 1 < 0 ? 0 : cur + 1); k <= n_images; ++k) idx[1 + k] = n_rois;
    }
    if (__any(unsorted) && (threadIdx.x & 63) == 0) idx[0] = 1;
}

// sum of the elements in bytes [b0, b0 + nbytes) of the row at `row` (16-byte aligned, LDS or global), by one lane:
// 16-byte pieces, the first and the last masked per byte.  The two byte masks are built once per call as pairs of 64-bit
// words (a shift and a select each) and applied through selects -- no branch inside the loop (round 3: the per-dword
// min / max / shift ladder of the first version was a third of the band kernel's vector instructions).
template <class PTR>
__device__ __forceinline__ uint32_t lane_row_sum(PTR row, int b0, int nbytes, int esz) {
    const int a0 = b0 & ~15;
    const int nch = (((b0 + nbytes + 15) & ~15) - a0) >> 4;
    const int lo = b0 - a0;                                           // 0..15 bytes to drop at the front of piece 0
    const int hi = b0 + nbytes - (a0 + 16 * (nch - 1));               // 1..16 bytes to keep of the last piece
    const unsigned long long ones = ~0ull;
    const unsigned long long f0 = lo >= 8 ? 0ull : (ones << (8 * lo)), f1 = lo >= 8 ? (ones << (8 * (lo - 8))) : ones;
    const unsigned long long l0 = hi >= 8 ? ones : ~(ones << (8 * hi));
    const unsigned long long l1 = hi >= 16 ? ones : (hi > 8 ? ~(ones << (8 * (hi - 8))) : 0ull);
    uint32_t sv = 0u;
    for (int d = 0; d < nch; ++d) {
        const uint4 x = *(const uint4*)(row + a0 + 16 * d);
        const unsigned long long m0 = (d == 0 ? f0 : ones) & (d == nch - 1 ? l0 : ones);
        const unsigned long long m1 = (d == 0 ? f1 : ones) & (d == nch - 1 ? l1 : ones);
        const uint32_t w[4] = {x.x & (uint32_t)m0, x.y & (uint32_t)(m0 >> 32), x.z & (uint32_t)m1, x.w & (uint32_t)(m1 >> 32)};
#pragma unroll
        for (int k = 0; k < 4; ++k) sv = (esz == 1) ? __builtin_amdgcn_sad_u8(w[k], 0u, sv) : (sv + (w[k] & 0xffffu) + (w[k] >> 16));
    }
    return sv;
}

constexpr int kBandList = 512;         // ROIs of one image handled per round of a band block
constexpr int kBandRegs = 6;           // 16-byte pieces of the band a thread holds in registers on the way to LDS (24 KiB per block)

// RBL = log2(rows per band); the band (<= 256 * kBandRegs * 16 bytes) is requested before anything else and lands in LDS
// while the block looks up its image's ROIs
__global__ __launch_bounds__(256) void k_profile_band(Job j, const LfaRoi* rois, const int32_t* idx, int RBL, int stride,
                                                      double* profiles) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int4 s_box[kBandList];                            // r0, r1, first byte and byte count of a row of the window
    __shared__ int s_roi[kBandList];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int image = blockIdx.y;
    const int RB = 1 << RBL;
    const int y0 = blockIdx.x << RBL, y1 = min(y0 + RB, j.H);
    const int esz = (j.dtype == LFA_U16) ? 2 : 1;
    const int row_bytes = j.W * j.C * esz;
    const uint8_t* img = j.images + (int64_t)image * j.img_bytes;
    const int n16 = ((y1 - y0) * row_bytes) >> 4;
    const uint4* src = (const uint4*)(img + (int64_t)y0 * row_bytes);
    uint4 hold[kBandRegs];
#pragma unroll
    for (int u = 0; u < kBandRegs; ++u) hold[u] = (tid + 256 * u < n16) ? src[tid + 256 * u] : make_uint4(0u, 0u, 0u, 0u);
    const int unsorted = idx[0];
    const int first = idx[1 + image], end = idx[2 + image];
    if (unsorted) return;                                        // unsorted ROI array: k_profile runs instead
    bool staged = false;
    for (int base = first; base < end; base += kBandList) {
        if (tid == 0) s_n = 0;
        __syncthreads();
        const int lim = min(end, base + kBandList);
        for (int i = base + tid; i < lim; i += blockDim.x) {
            const LfaRoi r = rois[i];
            const int len = r.r1 - r.r0;
            const bool bad = r.r0 < 0 || r.r1 > j.H || r.c0 < 0 || r.c1 > j.W || len < 0 || r.c1 < r.c0 || len > stride;
            if (!bad && r.image == image && r.r0 < y1 && r.r1 > y0) {
                const int k = atomicAdd(&s_n, 1);
                s_roi[k] = i;
                s_box[k] = make_int4(r.r0, r.r1, r.c0 * j.C * esz, (r.c1 - r.c0) * j.C * esz);
            }
        }
        if (!staged) {
#pragma unroll
            for (int u = 0; u < kBandRegs; ++u) if (tid + 256 * u < n16) ((uint4*)smem)[tid + 256 * u] = hold[u];
            staged = true;
        }
        __syncthreads();
        const int nl = s_n;
        for (int it = tid; it < (nl << RBL); it += blockDim.x) {  // one (ROI, row) per lane
            const int q = it >> RBL, y = y0 + (it & (RB - 1));
            const int4 bx = s_box[q];
            if (y < bx.x || y >= bx.y) continue;
            const uint32_t sv = lane_row_sum((const unsigned char*)smem + (y - y0) * row_bytes, bx.z, bx.w, esz);
            profiles[(int64_t)s_roi[q] * stride + (y - bx.x)] = (double)sv / (double)(bx.w / esz);
        }
        __syncthreads();
    }
}

// numpy's pairwise sum (lfa_algo.h np_pairwise) with the eight running sums of a leaf spread over eight lanes: lane j adds
// the terms j, j+8, j+16, ... in that order -- exactly the chain numpy's r[j] follows -- then the eight partial sums are
// exchanged and combined as numpy combines them.
template <int G>
struct GroupLeaf {                                               // a piece of at most 128 terms
    template <class F>
    __device__ __forceinline__ double operator()(const F& f, int off, int n) const {
        if (n < 8) return np_pairwise_leaf(f, off, n);
        const int j = threadIdx.x & 7;
        const int lim = n - (n % 8);
        double r = f(off + j);
        int i = 8 + j;
        for (; i + 24 < lim; i += 32) {                              // four terms fetched before the first is added (same order of adds)
            const double t0 = f(off + i), t1 = f(off + i + 8), t2 = f(off + i + 16), t3 = f(off + i + 24);
            r = r + t0; r = r + t1; r = r + t2; r = r + t3;
        }
        for (; i < lim; i += 8) r = r + f(off + i);
        // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) with r_j on lane j of every eight: lane ^ 1, lane ^ 2, then the
        // mirrored lane of the eight, which holds the other quad's sum (IEEE addition commutes bit for bit)
        double res = r + dpp_f64<kDppXor1>(r);
        res = res + dpp_f64<kDppXor2>(res);
        res = res + dpp_f64<kDppHalfMirror>(res);
        for (int i = lim; i < n; ++i) res = res + f(off + i);
        return res;
    }
};

template <int G>
struct GroupSum {
    template <class F>
    __device__ __forceinline__ double operator()(const F& f, int off, int n) const {
        if (n <= 128) return GroupLeaf<G>()(f, off, n);
        return np_pairwise_big(f, off, n, GroupLeaf<G>());         // out of line; splits like numpy, leaves as above
    }
};

// the peak search of one profile, run by the G lanes of its group with identical arguments; `dst` non-null on one lane
template <int G>
__device__ __forceinline__ void peaks_search(const double* p, int len, int inverse, int window, int two, int rng, LfaPeaks* dst) {
    TwoPeaks tp;
    if (two) {
        tp = two_peaks_t(p, len, inverse != 0, rng, inverse == 0, window, GroupArgExt<G>(), GroupWalk<G>(), GroupSum<G>());
    } else {
        tp.n = 0;
        if (len > 0) {
            const int pk = GroupArgExt<G>()(p, len, inverse != 0);
            tp.n = 1; tp.pk[0] = analyze_peaks_at_t(p, len, pk, inverse != 0, window, GroupWalk<G>(), GroupSum<G>()); tp.pk[1] = tp.pk[0];
        }
    }
    if (dst) store_peaks(dst, tp, len);
}

template <int G>
__device__ __noinline__ void peaks_search_global(const double* p, int len, int inverse, int window, int two, int rng, LfaPeaks* dst) {
    peaks_search<G>(p, len, inverse, window, two, rng, dst);
}

// analyze_peaks / the two-peak logic on profiles in global memory: G lanes per profile, profiles staged in LDS.
// Lengths come from `lengths` (lfa_peaks_batch) or from the ROI list the profiles were computed for (rois != null).
#ifndef LFA_PEAKS_WPE
#define LFA_PEAKS_WPE 4
#endif
template <int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFA_PEAKS_WPE))) void k_peaks(const double* profiles, const int32_t* lengths, int n, int stride,
                                               int inverse, int window, int two, int rng, LfaPeaks* out, int lstride,
                                               Job j, const LfaRoi* rois, const int32_t* counts, int slots,
                                               const int32_t* roi_lut, int n_luts) {
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef LFA_PEAKS_CLOCK
    const long long ck0 = __builtin_readcyclecounter();
#endif
    const int ppb = blockDim.x / G;
    const int slot = threadIdx.x / G, lg = threadIdx.x & (G - 1);
    double* sp = (double*)smem + (int64_t)slot * lstride;
    const int i = blockIdx.x * ppb + slot;
    bool valid = i < n;
    int len = 0;
    bool bad = false;
    if (valid && rois) {
        if (counts && (i % slots) >= counts[i / slots]) valid = false;
        if (valid) {
            const LfaRoi roi = rois[i];
            len = roi.r1 - roi.r0;
            bad = roi.image < 0 || roi.image >= j.n || roi.r0 < 0 || roi.r1 > j.H || roi.c0 < 0 || roi.c1 > j.W ||
                  len < 0 || roi.c1 < roi.c0 || len > stride;
            if (roi_lut && (roi_lut[i] < 0 || roi_lut[i] >= n_luts)) bad = true;
        }
    } else if (valid) {
        len = lengths[i];
        bad = len < 0 || len > stride;
    }
    const double* g = profiles + (int64_t)(valid ? i : 0) * stride;
    const bool work = valid && !bad;
    const bool use_lds = lstride > 0 && len <= lstride;
#ifdef LFA_PEAKS_CLOCK
    const long long ck1 = __builtin_readcyclecounter();
#endif
    if (work && use_lds) {
        constexpr int U = 8;                                         // eight loads in flight per lane (one at a time: 11 000 of the
        for (int k0 = lg; k0 < len; k0 += G * U) {                   //  kernel's 47 000 clocks per profile went into this copy)
            double x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int k = k0 + u * G; x[u] = g[k < len ? k : k0]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { const int k = k0 + u * G; if (k < len) sp[k] = x[u]; }
        }
    }
    __syncthreads();
#ifdef LFA_PEAKS_CLOCK
    const long long ck2 = __builtin_readcyclecounter();
#endif
    if (bad) { if (lg == 0) store_bad(&out[i], len); return; }
    if (!work) return;
    // the search is inlined for the LDS copy of the profile (ds_read on the dependent chains); the profile that does not
    // fit the LDS slots (more than ~8000 rows) takes an out-of-line copy of the same code on the global pointer
    if (use_lds) peaks_search<G>(sp, len, inverse, window, two, rng, lg == 0 ? &out[i] : nullptr);
    else peaks_search_global<G>(g, len, inverse, window, two, rng, lg == 0 ? &out[i] : nullptr);
#ifdef LFA_PEAKS_CLOCK
    // dev aid: phase lengths in shader clocks instead of the areas (results are wrong on purpose)
    const long long ck3 = __builtin_readcyclecounter();
    if (lg == 0) { out[i].area[0] = (double)(ck1 - ck0) + 1e-6 * (double)(ck2 - ck1); out[i].area[1] = (double)(ck3 - ck2); }
#endif
}

// LDS plan: `lstride` float64 slots per profile, ONE wave per block (round 3: blocks of four waves ended with their slowest
// profile and the kernel ran in two uneven rounds -- 0.116 -> 0.099 ms for the 16 384 profiles of configs[4]'s 64-image call).
// 16 lanes per profile (four per wave) while the slots are short; 32 lanes once a slot passes 2.5 KiB -- the search is a
// chain of dependent steps whose length hardly depends on the group width, so wider groups cost nothing per profile and
// halve what a wave keeps in LDS (0.099 -> 0.076 ms at 512-row slots; at 288 rows 16 lanes are 10 % faster); one wave per
// profile when even two do not fit 64 KiB, no LDS staging at all beyond that.  (More waves per SIMD through a register cap,
// amdgpu_waves_per_eu 5 / 6 / 8: 0.075 / 0.085 / 0.090 ms -- it is not occupancy that bounds the kernel.)
struct LdsPlan { int g, threads, lstride; size_t bytes; };

static LdsPlan plan_lds(int max_len) {
    LdsPlan p;
    const int ls = (max_len + 1) & ~1;
    const size_t per = (size_t)ls * 8;
    static const int env_g = [] { const char* e = getenv("LFA_PEAKS_G"); return e ? atoi(e) : 0; }();      // experiments only
    int g = per <= 2560 ? 16 : (per * 2 <= 65536 ? 32 : 64);
    if ((env_g == 8 || env_g == 16 || env_g == 32 || env_g == 64) && per * (64 / env_g) <= 65536) g = env_g;
    if (per * (64 / g) <= 65536) {
        p.g = g; p.threads = 64; p.lstride = ls; p.bytes = per * (64 / g);
    } else {
        p.g = 64; p.threads = 256; p.lstride = 0; p.bytes = 0;
    }
    return p;
}

static void launch_peaks_k(const double* profiles, const int32_t* lengths, int n, int stride, int max_len, int inverse,
                           int window, int two, int rng, LfaPeaks* out, const Job& j, const LfaRoi* rois,
                           const int32_t* counts, int slots, const int32_t* roi_lut, int n_luts, hipStream_t s) {
    const LdsPlan p = plan_lds(max_len);
    const int ppb = p.threads / p.g;
    const dim3 grid((n + ppb - 1) / ppb), block(p.threads);
    if (p.g == 16)
        LFA_LAUNCH("k_peaks", k_peaks<16>, grid, block, p.bytes, s, profiles, lengths, n, stride, inverse, window, two, rng, out,
                   p.lstride, j, rois, counts, slots, roi_lut, n_luts);
    else if (p.g == 32)
        LFA_LAUNCH("k_peaks", k_peaks<32>, grid, block, p.bytes, s, profiles, lengths, n, stride, inverse, window, two, rng, out,
                   p.lstride, j, rois, counts, slots, roi_lut, n_luts);
    else if (p.g == 8)
        LFA_LAUNCH("k_peaks", k_peaks<8>, grid, block, p.bytes, s, profiles, lengths, n, stride, inverse, window, two, rng, out,
                   p.lstride, j, rois, counts, slots, roi_lut, n_luts);
    else
        LFA_LAUNCH("k_peaks", k_peaks<64>, grid, block, p.bytes, s, profiles, lengths, n, stride, inverse, window, two, rng, out,
                   p.lstride, j, rois, counts, slots, roi_lut, n_luts);
}

hipError_t launch_profiles(const Job& j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots,
                           const int32_t* rs_off, int inverse, int rng, double* profiles, int stride, LfaPeaks* peaks,
                           const uint8_t* luts, int n_luts, const int32_t* roi_lut, hipStream_t s, int32_t* roi_idx) {
    if (n_rois <= 0) return hipSuccess;
    // dense ROI sets take the band kernel (see k_profile_band): plain profiles of aligned images with rows that fit the LDS
    // band, at least 32 ROIs per image on average, a ROI array sorted by image (checked on the device: k_roi_index; an
    // unsorted array makes the band kernel return at once and k_profile, launched behind it with the opposite test, run)
    const int64_t row_bytes = (int64_t)j.W * j.C * (j.dtype == LFA_U16 ? 2 : 1);
    int band_rows = 0;
    if (roi_idx && !luts && !counts && j.aligned16 && (row_bytes & 15) == 0 && row_bytes <= 256 * kBandRegs * 16 && (int64_t)n_rois >= 32ll * j.n) {
        static const int band_kb = [] { const char* e = getenv("LFA_BAND_KB"); return e ? atoi(e) : 24; }();
        int rbl = 3;                                             // rows per band: 8, 4, 2 or 1 -- the band must fit its registers / LDS
        while (rbl > 0 && (((int64_t)row_bytes << rbl) > (int64_t)band_kb * 1024 || ((int64_t)row_bytes << rbl) > 256 * kBandRegs * 16)) --rbl;
        band_rows = 1 << rbl;
    }
    if (band_rows) {
        int rbl = 0;
        while ((1 << rbl) < band_rows) ++rbl;
        const hipError_t e0 = hipMemsetAsync(roi_idx, 0, 4, s);
        if (e0 != hipSuccess) return e0;
        LFA_LAUNCH("k_roi_index", k_roi_index, dim3((unsigned)std::min<int64_t>(((int64_t)n_rois + 255) / 256, 1024)), dim3(256), 0, s, rois, n_rois, j.n, roi_idx);
        const dim3 grid((j.H + band_rows - 1) / band_rows, j.n);
        LFA_LAUNCH("k_profile_band", k_profile_band, grid, dim3(256), (size_t)band_rows * row_bytes, s, j, rois, roi_idx, rbl, stride,
                   profiles);
    }
    // the analyze path only ever profiles windows of int(.6h) - int(.25h) <= 0.35*H + 1 rows: size the LDS for that
    int max_len = stride;
    if (counts) {
        const int wl = (int)(0.35 * j.H) + 2;
        if (wl < max_len) max_len = wl;
        if (j.win_hint > 0 && j.win_hint < max_len) max_len = j.win_hint;     // what the last call needed; longer windows fall back
    }
    // row sums: one wave per ROI (all 64 lanes keep loads in flight), float64 accumulators in LDS
    {
        const int ls = (max_len + 1) & ~1;
        int waves = 4;                               // (1, 2 or 4 ROIs per block measure the same: 0.18 ms for 16 384 ROIs)
        while (waves > 1 && (size_t)waves * ls * 8 > 65536) waves >>= 1;
        const int lstride = ((size_t)waves * ls * 8 <= 65536) ? ls : 0;
        if (!lstride) waves = 4;
        const dim3 grid((n_rois + waves - 1) / waves), block(64 * waves);
        const size_t acc_bytes = (size_t)waves * lstride * 8;
        if (luts)
            LFA_LAUNCH("k_profile_lut", (k_profile<64, true>), grid, block, acc_bytes + (size_t)waves * 256, s, j, rois, n_rois, counts,
                       slots, rs_off, inverse, rng, profiles, stride, peaks, lstride, luts, roi_lut, n_luts, nullptr);
        else {
            // Chained lanes (lfa_chain_passes): this kernel and k_peaks behind it are kept out of the next chunk's histogram pass -- which
            // is bound by its VALU and LDS work and stretches by a tenth with them beside it -- by asking for 60 KiB of LDS: that does
            // not fit beside three k_hist_rgb blocks (or two of k_hist_u16) nor into the room one retiring block leaves, so the
            // kernel runs in the seam after the pass, where the tail's other half runs anyway.  Measured with the faster
            // k_runs_count (its successor k_components then starts BEFORE the histogram pass and the rest of the tail followed it
            // into the pass): headline 3.98 -> 3.83 ms, configs[2] 3.40 -> 3.35 ms per 1024 frames.  Not behind the equal-channel pass
            // (k_grey_hist reads at the HBM ceiling and has VALU time to spare: gated 2.97 ms, not gated 2.90) and not for lanes
            // that run free (configs[3]).
            const size_t gate = (j.wait_event && !j.grey_mode && counts) ? (size_t)60 * 1024 : (size_t)0;
            LFA_LAUNCH("k_profile", (k_profile<64, false>), grid, block, acc_bytes > gate ? acc_bytes : gate, s, j, rois, n_rois, counts, slots, rs_off,
                       inverse, rng, profiles, stride, peaks, lstride, luts, roi_lut, n_luts, band_rows ? roi_idx : nullptr);
        }
    }
    // peaks: the sequential search, 16 lanes per profile
    launch_peaks_k(profiles, nullptr, n_rois, stride, max_len, inverse, 25, 1, rng, peaks, j, rois, counts, slots,
                   luts ? roi_lut : nullptr, n_luts, s);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Regions of the other element types np.nanmean takes (lfa_profile_any_batch): int8 / int16 / int32 / uint32 -- np.mean
// accumulates them in float64, exact while the sums stay below 2^53, so an integer row sum and one division -- and
// float64: NaNs count as 0 and are left out of the divisor (numpy nanfunctions _replace_nan, sum, _divide_by_count), and
// the row sum is numpy's pairwise sum over the w * c elements of the (copied, hence contiguous) window row
// (lfa_algo.h np_pairwise; checked row by row against fixtures made by the reference's own lines).  One wave per ROI, a
// lane per row: these are single regions handed over by a caller, not the batched hot path.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
struct NanZeroAt {                     // element i of a row, NaN -> 0
    const T* p;
    __device__ double operator()(int i) const { const double x = (double)p[i]; return x != x ? 0.0 : x; }
};

template <class T>
__global__ __launch_bounds__(64) void k_profile_any(const T* data, int n, int H, int W, int C, const LfaRoi* rois, int n_rois,
                                                    double* profiles, int stride, LfaPeaks* peaks) {
    const int ri = blockIdx.x;
    const LfaRoi roi = rois[ri];
    const int len = roi.r1 - roi.r0;
    const bool bad = roi.image < 0 || roi.image >= n || roi.r0 < 0 || roi.r1 > H || roi.c0 < 0 || roi.c1 > W || len < 0 ||
                     roi.c1 < roi.c0 || len > stride;
    if (bad) { if (threadIdx.x == 0) store_bad(&peaks[ri], len); return; }     // k_peaks skips it by the same test
    const int ne = (roi.c1 - roi.c0) * C;
    const T* base = data + (int64_t)roi.image * H * W * C;
    double* prof = profiles + (int64_t)ri * stride;
    for (int r = threadIdx.x; r < len; r += 64) {
        const T* row = base + ((int64_t)(roi.r0 + r) * W + roi.c0) * C;
        if constexpr (sizeof(T) == 8) {                                       // float64
            int cnt = 0;
            for (int k = 0; k < ne; ++k) { const double x = (double)row[k]; cnt += (x == x); }
            const double tot = np_pairwise(NanZeroAt<T>{row}, 0, ne);
            prof[r] = tot / (double)cnt;                                      // no finite element: 0 / 0 = NaN, as numpy
        } else {
            long long sm = 0;
            for (int k = 0; k < ne; ++k) sm += (long long)row[k];
            prof[r] = (double)sm / (double)ne;
        }
    }
}

hipError_t launch_profiles_any(const void* data, int elem, int n, int H, int W, int C, const LfaRoi* rois, int n_rois, int inverse,
                               int rng, double* profiles, int stride, LfaPeaks* peaks, hipStream_t s) {
    if (n_rois <= 0) return hipSuccess;
    const dim3 grid(n_rois), block(64);
    switch (elem) {
        case LFA_E_I8:  LFA_LAUNCH("k_profile_any", k_profile_any<int8_t>, grid, block, 0, s, (const int8_t*)data, n, H, W, C, rois, n_rois, profiles, stride, peaks); break;
        case LFA_E_I16: LFA_LAUNCH("k_profile_any", k_profile_any<int16_t>, grid, block, 0, s, (const int16_t*)data, n, H, W, C, rois, n_rois, profiles, stride, peaks); break;
        case LFA_E_I32: LFA_LAUNCH("k_profile_any", k_profile_any<int32_t>, grid, block, 0, s, (const int32_t*)data, n, H, W, C, rois, n_rois, profiles, stride, peaks); break;
        case LFA_E_U32: LFA_LAUNCH("k_profile_any", k_profile_any<uint32_t>, grid, block, 0, s, (const uint32_t*)data, n, H, W, C, rois, n_rois, profiles, stride, peaks); break;
        case LFA_E_F64: LFA_LAUNCH("k_profile_any", k_profile_any<double>, grid, block, 0, s, (const double*)data, n, H, W, C, rois, n_rois, profiles, stride, peaks); break;
        default: return hipErrorInvalidValue;
    }
    Job j{};
    j.n = n; j.H = H; j.W = W; j.C = C;
    launch_peaks_k(profiles, nullptr, n_rois, stride, stride, inverse, 25, 1, rng, peaks, j, rois, nullptr, 0, nullptr, 0, s);
    return hipGetLastError();
}

hipError_t launch_peaks(const double* profiles, const int32_t* lengths, int n, int stride, int inverse, int window,
                        int two, int rng, LfaPeaks* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    Job j{};
    launch_peaks_k(profiles, lengths, n, stride, stride, inverse, window, two, rng, out, j, nullptr, nullptr, 0, nullptr, 0, s);
    return hipGetLastError();
}

}  // namespace lfa
