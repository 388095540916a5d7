// k_profile.hip -- stages S8-S9: per-ROI row-mean profiles (segmented reduction) and top-2 peaks  (gfx950)
//
// Reference: /root/reference/LFA-Image_Analysis/detection.py:205-245 (image_analysis), :247-281
// (inverse_image_analysis), :283-316 (analyze_peaks).  profile[k] = np.nanmean(window, axis=cols(+channels))[k]; for the
// integer images this path handles that is an exact integer row sum and one float64 division.  The peak logic is the
// shared sequential code of lfa_algo.h (bit-for-bit numpy summation order) run by one wave with a wave-parallel
// first-extremum search.
#include "lfa_internal.h"
#include "lfa_algo.h"
#include "lfa_device.h"

namespace lfa {

constexpr int kProfLds = 4096;     // profiles up to this length are staged in LDS for the peak search

// np.argmax / np.argmin over p[0..n): first extremum, first NaN wins.  Called by all 64 lanes of one wave.
struct WaveArgExt {
    __device__ int operator()(const double* p, int n, bool want_min) const {
        const int lane = threadIdx.x & 63;
        double m = 0.0; int idx = 0x7fffffff; bool isn = false;
        for (int i = lane; i < n; i += 64) {
            const double x = p[i];
            if (idx == 0x7fffffff) { m = x; idx = i; isn = (x != x); if (isn) break; continue; }
            const bool upd = want_min ? !(x >= m) : !(x <= m);
            if (upd) { m = x; idx = i; if (x != x) { isn = true; break; } }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double m2 = __shfl_xor(m, o);
            const int i2 = __shfl_xor(idx, o);
            const int n2 = __shfl_xor((int)isn, o);
            bool take;
            if (i2 == 0x7fffffff) take = false;
            else if (idx == 0x7fffffff) take = true;
            else if (isn || n2) take = n2 && (!isn || i2 < idx);
            else if (want_min ? (m2 < m) : (m2 > m)) take = true;
            else if (m2 == m) take = i2 < idx;
            else take = false;
            if (take) { m = m2; idx = i2; isn = n2 != 0; }
        }
        return idx;
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

// One WAVE per ROI, up to four ROIs per block (the peak search is sequential per profile, so the parallelism is across
// ROIs; the LDS per wave is `lstride` float64 + `lstride` uint32 slots, chosen at launch from the profile stride).
//   pooled: lfa_analyze_batch -- k_box_sums already wrote the window's integer row sums, no pixel is read here;
//   dwords: the ROI's rows are read as aligned dwords, every (row, dword) an independent item, row sums in LDS atomics;
//   plain:  16 lanes per row (unaligned batches, or profiles longer than the LDS slots: then the peak search reads
//           the profile from global memory).
// In every mode profile[r] = integer row sum / element count in float64.
__global__ __launch_bounds__(256) void k_profile(Job j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots,
                                                 const int32_t* rs_off, int inverse, int rng, double* profiles, int stride,
                                                 LfaPeaks* peaks, int lstride) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int wpb = blockDim.x >> 6;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* sp = (double*)smem + (int64_t)wv * lstride;
    unsigned int* rs = (unsigned int*)((double*)smem + (int64_t)wpb * lstride) + (int64_t)wv * lstride;
    const int ri = blockIdx.x * wpb + wv;
    bool valid = ri < n_rois;
    if (valid && counts && (ri % slots) >= counts[ri / slots]) valid = false;   // analyze path: slot beyond this image's strips
    LfaRoi roi; roi.image = 0; roi.r0 = roi.r1 = roi.c0 = roi.c1 = 0;
    if (valid) roi = rois[ri];
    const int len = roi.r1 - roi.r0;
    const bool bad = valid && (roi.image < 0 || roi.image >= j.n || roi.r0 < 0 || roi.r1 > j.H || roi.c0 < 0 || roi.c1 > j.W ||
                               len < 0 || roi.c1 < roi.c0 || len > stride);
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
    // u8 row sums fit 32 bits for any supported width; u16 rows up to 65536 elements too
    const bool dwords = work && pooled < 0 && use_lds && j.aligned16 && (row_bytes & 3) == 0 && ne > 0 && (esz == 1 || ne <= 65536);
    if (dwords) for (int r = lane; r < len; r += 64) rs[r] = 0;
    __syncthreads();
    if (pooled >= 0) {
        const uint32_t* rsum = j.rowsum + (int64_t)roi.image * j.rowsum_cap + pooled;
        for (int r = lane; r < len; r += 64) {
            const double m = (double)rsum[r] / cnt;
            prof[r] = m; sp[r] = m;
        }
    } else if (dwords) {
        const int64_t b0 = ((int64_t)roi.r0 * j.W + roi.c0) * C * esz;       // first byte of row 0 of the ROI
        const int64_t nbytes = (int64_t)ne * esz;
        if ((row_bytes & 15) == 0) {
            // 16-byte chunks (a quarter of the load instructions of the dword form); same item scheme, four in flight per lane
            const int64_t a0 = b0 & ~(int64_t)15;                              // row_bytes % 16 == 0: same phase on every row
            const int nch = (int)(((b0 + nbytes + 15) & ~(int64_t)15) - a0) >> 4;
            const int items = len * nch;
            const int q64 = 64 / nch, m64 = 64 - q64 * nch;
            int r = lane / nch, d = lane - r * nch;
            for (int it0 = lane; it0 < items; it0 += 64 * 4) {
                uint4 x[4];
                int rr[4], dd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rr[u] = r; dd[u] = d;
                    if (it0 + 64 * u < items) x[u] = *(const uint4*)(base + (int64_t)r * row_bytes + a0 + 16 * (int64_t)d);
                    else x[u] = make_uint4(0u, 0u, 0u, 0u);
                    r += q64; d += m64;
                    if (d >= nch) { d -= nch; ++r; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (it0 + 64 * u >= items) break;
                    const unsigned int wds[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                    unsigned int sv = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int64_t a = a0 + 16 * (int64_t)dd[u] + 4 * k;
                        unsigned int vm = 0xffffffffu;
                        if (a + 4 <= b0 || a >= b0 + nbytes) vm = 0u;
                        else {
                            if (a < b0) vm &= 0xffffffffu << (8 * (int)(b0 - a));
                            if (a + 4 > b0 + nbytes) vm &= 0xffffffffu >> (8 * (int)(a + 4 - (b0 + nbytes)));
                        }
                        const unsigned int y = wds[k] & vm;
                        sv = (esz == 1) ? __builtin_amdgcn_sad_u8(y, 0u, sv) : (sv + (y & 0xffffu) + (y >> 16));
                    }
                    if (sv) atomicAdd(&rs[rr[u]], sv);
                }
            }
        } else {
        const int64_t a0 = b0 & ~(int64_t)3;                                   // row_bytes % 4 == 0: same phase on every row
        const int ndw = (int)(((b0 + nbytes + 3) & ~(int64_t)3) - a0) >> 2;    // dwords covering one row
        const int items = len * ndw;
        // eight independent loads per lane are issued before any of them is used (the loop is latency-bound otherwise);
        // (row, dword) of item `it` advance by 64 items per step without a division
        const int q64 = 64 / ndw, m64 = 64 - q64 * ndw;
        int r = lane / ndw, d = lane - r * ndw;
        for (int it0 = lane; it0 < items; it0 += 64 * 8) {
            unsigned int x[8];
            int rr[8], dd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rr[u] = r; dd[u] = d;
                x[u] = (it0 + 64 * u < items) ? *(const unsigned int*)(base + (int64_t)r * row_bytes + a0 + 4 * (int64_t)d) : 0u;
                r += q64; d += m64;
                if (d >= ndw) { d -= ndw; ++r; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (it0 + 64 * u >= items) break;
                const int64_t a = a0 + 4 * (int64_t)dd[u];
                unsigned int vm = 0xffffffffu;
                if (a < b0) vm &= 0xffffffffu << (8 * (int)(b0 - a));
                if (a + 4 > b0 + nbytes) vm &= 0xffffffffu >> (8 * (int)(a + 4 - (b0 + nbytes)));
                const unsigned int y = x[u] & vm;
                const unsigned int sv = (esz == 1) ? __builtin_amdgcn_sad_u8(y, 0u, 0u) : ((y & 0xffffu) + (y >> 16));
                if (sv) atomicAdd(&rs[rr[u]], sv);
            }
        }
        }
    } else if (work) {
        const int sub = lane & 15, grp = lane >> 4;
        for (int r = grp; r < len + 3; r += 4) {            // all 16 lanes of a group stay together for the shuffles
            const bool in = r < len;
            const int64_t e0 = ((int64_t)(roi.r0 + (in ? r : 0)) * j.W + roi.c0) * C;
            unsigned long long s = 0;
            if (in) {
                if (esz == 1) { for (int k = sub; k < ne; k += 16) s += base[e0 + k]; }
                else { const uint16_t* b16 = (const uint16_t*)base; for (int k = sub; k < ne; k += 16) s += b16[e0 + k]; }
            }
            s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if (in && sub == 0) {
                const double m = (double)s / cnt;         // ne == 0 -> 0/0 = NaN, like np.nanmean of an empty slice
                prof[r] = m;
                if (use_lds) sp[r] = m;
            }
        }
    }
    __syncthreads();
    if (dwords)
        for (int r = lane; r < len; r += 64) {
            const double m = (double)rs[r] / cnt;
            prof[r] = m; sp[r] = m;
        }
    __syncthreads();
    if (bad) {
        if (lane == 0) { LfaPeaks r; r.n_peaks = -1; r.length = len; r.start[0] = r.start[1] = r.end[0] = r.end[1] = -1;
                         r.area[0] = r.area[1] = 0.0; peaks[ri] = r; }
        return;
    }
    if (!work) return;
    const double* p = use_lds ? sp : prof;
    const TwoPeaks tp = two_peaks_t(p, len, inverse != 0, rng, inverse == 0, 25, WaveArgExt());
    if (lane == 0) store_peaks(&peaks[ri], tp, len);
}

// analyze_peaks / the two-peak logic on caller-supplied profiles: one wave per profile
__global__ __launch_bounds__(64) void k_peaks(const double* profiles, const int32_t* lengths, int n, int stride,
                                              int inverse, int window, int two, int rng, LfaPeaks* out) {
    __shared__ double sp[kProfLds];
    const int i = blockIdx.x;
    if (i >= n) return;
    const int len = lengths[i];
    const double* g = profiles + (int64_t)i * stride;
    const bool bad = len < 0 || len > stride;
    if (bad) { if (threadIdx.x == 0) { LfaPeaks r; r.n_peaks = -1; r.length = len; r.start[0] = r.start[1] = r.end[0] = r.end[1] = -1;
                                        r.area[0] = r.area[1] = 0.0; out[i] = r; } return; }
    const bool use_lds = len <= kProfLds;
    if (use_lds) for (int k = threadIdx.x; k < len; k += 64) sp[k] = g[k];
    __syncthreads();
    const double* p = use_lds ? sp : g;
    TwoPeaks tp;
    if (two) {
        tp = two_peaks_t(p, len, inverse != 0, rng, inverse == 0, window, WaveArgExt());
    } else {
        tp.n = 0;
        if (len > 0) {
            const int pk = WaveArgExt()(p, len, inverse != 0);
            tp.n = 1; tp.pk[0] = analyze_peaks_at(p, len, pk, inverse != 0, window); tp.pk[1] = tp.pk[0];
        }
    }
    if (threadIdx.x == 0) store_peaks(&out[i], tp, len);
}

hipError_t launch_profiles(const Job& j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots,
                           const int32_t* rs_off, int inverse, int rng, double* profiles, int stride, LfaPeaks* peaks,
                           hipStream_t s) {
    if (n_rois <= 0) return hipSuccess;
    // LDS per wave: 12 bytes per profile slot; up to 4 waves per block within 64 KiB; no LDS staging beyond that
    int lstride = (stride + 1) & ~1;
    int wpb = 4;
    while (wpb > 1 && (size_t)wpb * lstride * 12 > 65536) wpb >>= 1;
    if ((size_t)wpb * lstride * 12 > 65536) lstride = 0;
    const size_t lds = (size_t)wpb * lstride * 12;
    LFA_LAUNCH("k_profile", k_profile, dim3((n_rois + wpb - 1) / wpb), dim3(64 * wpb), lds, s, j, rois, n_rois, counts, slots,
               rs_off, inverse, rng, profiles, stride, peaks, lstride);
    return hipGetLastError();
}

hipError_t launch_peaks(const double* profiles, const int32_t* lengths, int n, int stride, int inverse, int window,
                        int two, int rng, LfaPeaks* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    LFA_LAUNCH("k_peaks", k_peaks, dim3(n), dim3(64), 0, s, profiles, lengths, n, stride, inverse, window, two, rng, out);
    return hipGetLastError();
}

}  // namespace lfa
