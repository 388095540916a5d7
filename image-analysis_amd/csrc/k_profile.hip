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

// One 256-thread block per ROI.  The ROI's rows are read as aligned dwords; every (row, dword) pair is an independent
// work item (so many loads are in flight per thread), the integer row sums are accumulated with LDS atomics, then
// profile[r] = sum / count in float64.  Rows beyond kProfLds (or unaligned batches) take the plain path.
__global__ __launch_bounds__(256) void k_profile(Job j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots,
                                                 int inverse, int rng, double* profiles, int stride, LfaPeaks* peaks) {
    __shared__ double sp[kProfLds];
    __shared__ unsigned int rs[kProfLds];
    const int ri = blockIdx.x;
    if (ri >= n_rois) return;
    if (counts && (ri % slots) >= counts[ri / slots]) return;        // analyze path: slot beyond this image's strips
    const LfaRoi roi = rois[ri];
    const int len = roi.r1 - roi.r0;
    double* prof = profiles + (int64_t)ri * stride;
    const int t = threadIdx.x;
    const bool bad = roi.image < 0 || roi.image >= j.n || roi.r0 < 0 || roi.r1 > j.H || roi.c0 < 0 || roi.c1 > j.W ||
                     len < 0 || roi.c1 < roi.c0 || len > stride;
    if (bad) {
        if (t == 0) { LfaPeaks r; r.n_peaks = -1; r.length = len; r.start[0] = r.start[1] = r.end[0] = r.end[1] = -1;
                      r.area[0] = r.area[1] = 0.0; peaks[ri] = r; }
        return;
    }
    const uint8_t* base = j.images + (int64_t)roi.image * j.img_bytes;
    const int C = j.C;
    const int esz = (j.dtype == LFA_U16) ? 2 : 1;
    const int ne = (roi.c1 - roi.c0) * C;                 // elements per row
    const double cnt = (double)ne;
    const bool use_lds = len <= kProfLds;
    const int64_t row_bytes = (int64_t)j.W * C * esz;
    // u8 row sums fit 32 bits for any supported width; u16 rows up to 65535 elements too (65535*65535 < 2^32)
    if (use_lds && j.aligned16 && (row_bytes & 3) == 0 && ne > 0 && (esz == 1 || ne <= 65536)) {
        for (int r = t; r < len; r += 256) rs[r] = 0;
        __syncthreads();
        const int64_t b0 = ((int64_t)roi.r0 * j.W + roi.c0) * C * esz;       // first byte of row 0 of the ROI
        const int64_t nbytes = (int64_t)ne * esz;
        const int64_t a0 = b0 & ~(int64_t)3;                                   // row_bytes % 4 == 0: same phase on every row
        const int ndw = (int)(((b0 + nbytes + 3) & ~(int64_t)3) - a0) >> 2;    // dwords covering one row
        const int items = len * ndw;
#pragma unroll 4
        for (int it = t; it < items; it += 256) {
            const int r = it / ndw, d = it - r * ndw;
            const int64_t a = a0 + 4 * (int64_t)d;                             // byte offset inside row r's line
            unsigned int x = *(const unsigned int*)(base + (int64_t)r * row_bytes + a);
            unsigned int vm = 0xffffffffu;
            if (a < b0) vm &= 0xffffffffu << (8 * (int)(b0 - a));
            if (a + 4 > b0 + nbytes) vm &= 0xffffffffu >> (8 * (int)(a + 4 - (b0 + nbytes)));
            x &= vm;
            const unsigned int s = (esz == 1) ? __builtin_amdgcn_sad_u8(x, 0u, 0u) : ((x & 0xffffu) + (x >> 16));
            if (s) atomicAdd(&rs[r], s);
        }
        __syncthreads();
        for (int r = t; r < len; r += 256) {
            const double m = (double)rs[r] / cnt;
            prof[r] = m; sp[r] = m;
        }
    } else {
        const int sub = t & 15, grp = t >> 4;
        for (int r = grp; r < len; r += 16) {
            const int64_t e0 = ((int64_t)(roi.r0 + r) * j.W + roi.c0) * C;
            unsigned long long s = 0;
            if (esz == 1) { for (int k = sub; k < ne; k += 16) s += base[e0 + k]; }
            else { const uint16_t* b16 = (const uint16_t*)base; for (int k = sub; k < ne; k += 16) s += b16[e0 + k]; }
            s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if (sub == 0) {
                const double m = (double)s / cnt;         // ne == 0 -> 0/0 = NaN, like np.nanmean of an empty slice
                prof[r] = m;
                if (use_lds) sp[r] = m;
            }
        }
    }
    __syncthreads();
    if (t >= 64) return;
    const double* p = use_lds ? sp : prof;
    const TwoPeaks tp = two_peaks_t(p, len, inverse != 0, rng, inverse == 0, 25, WaveArgExt());
    if (t == 0) store_peaks(&peaks[ri], tp, len);
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

hipError_t launch_profiles(const Job& j, const LfaRoi* rois, int n_rois, const int32_t* counts, int slots, int inverse,
                           int rng, double* profiles, int stride, LfaPeaks* peaks, hipStream_t s) {
    if (n_rois <= 0) return hipSuccess;
    LFA_LAUNCH("k_profile", k_profile, dim3(n_rois), dim3(256), 0, s, j, rois, n_rois, counts, slots, inverse, rng, profiles,
                       stride, peaks);
    return hipGetLastError();
}

hipError_t launch_peaks(const double* profiles, const int32_t* lengths, int n, int stride, int inverse, int window,
                        int two, int rng, LfaPeaks* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    LFA_LAUNCH("k_peaks", k_peaks, dim3(n), dim3(64), 0, s, profiles, lengths, n, stride, inverse, window, two, rng, out);
    return hipGetLastError();
}

}  // namespace lfa
