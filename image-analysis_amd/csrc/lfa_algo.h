// lfa_algo.h -- sequential arithmetic shared by the HIP kernels (and compiled for the host by the unit-test shim
// tests/hostcheck/, which exists only to exercise these functions on a machine without a GPU).
//
// Everything here must reproduce numpy's results bit for bit, so this file is compiled with -ffp-contract=off and
// spells out every rounding step.  References (relative to /root/reference):
//   analyze_peaks / image_analysis peak logic   LFA-Image_Analysis/detection.py:221-245, :257-281, :283-316
//   float Otsu                                   image-analysis/lib/python3.10/site-packages/skimage/filters/thresholding.py:364-381
//   np.histogram edges (linspace)                .../numpy/core/function_base.py:145-167
//   np.add.reduce pairwise summation             numpy/core/src/umath/loops_utils.h.src (*_pairwise_sum), not vendored as source
//                                                in the reference tree; order pinned empirically (tests/test_host_algo.py)
#ifndef LFA_ALGO_H_
#define LFA_ALGO_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define LFA_HD __host__ __device__ __forceinline__
#define LFA_HD_NOINLINE __host__ __device__ __noinline__
#else
#define LFA_HD inline
#define LFA_HD_NOINLINE inline
#endif

// dev aid (tools/r3_peaks_clock.py, variant build -DLFA_PEAKS_CLOCK): shader clocks per phase of the peak search, summed over
// the waves' first lanes into lfa_ck[]; expands to nothing in the product build
#if defined(LFA_PEAKS_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long lfa_ck[16 * 1024];
#define LFA_CK_BEGIN() long long ck_prev_ = __builtin_readcyclecounter()
#define LFA_CK(k) do { const long long t_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) atomicAdd(&lfa_ck[(k) * 1024 + (blockIdx.x & 1023)], (unsigned long long)(t_ - ck_prev_)); ck_prev_ = __builtin_readcyclecounter(); } while (0)
#else
#define LFA_CK_BEGIN() do {} while (0)
#define LFA_CK(k) do {} while (0)
#endif

namespace lfa {

// ---------------------------------------------------------------------------------------------------------------
// numpy's pairwise sum of f(0..n-1).  n < 8: plain loop; n <= 128: eight running sums, combined as a tree, then the
// remainder one by one; larger: split at n/2 rounded down to a multiple of 8.
// ---------------------------------------------------------------------------------------------------------------
template <class F>
LFA_HD double np_pairwise_leaf(const F& f, int off, int n) {       // n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + f(off + i);
        return res;
    }
    double r0 = f(off + 0), r1 = f(off + 1), r2 = f(off + 2), r3 = f(off + 3);
    double r4 = f(off + 4), r5 = f(off + 5), r6 = f(off + 6), r7 = f(off + 7);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
        r0 = r0 + f(off + i + 0); r1 = r1 + f(off + i + 1); r2 = r2 + f(off + i + 2); r3 = r3 + f(off + i + 3);
        r4 = r4 + f(off + i + 4); r5 = r5 + f(off + i + 5); r6 = r6 + f(off + i + 6); r7 = r7 + f(off + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + f(off + i);
    return res;
}

// The recursion of numpy's pairwise sum (split at n/2 rounded down to a multiple of 8 until a piece has <= 128 elements;
// result = sum(left) + sum(right)) as a post-order walk over the leaves, left to right, WITHOUT a stack in memory: the only
// state per level is the finished sum of a left child that waits for its right sibling (`acc`, kept in registers -- it
// is only ever indexed by unrolled compile-time constants, the level wanted is picked by compares) and one bit that says
// whether the walk is in the left or the right child there (`path`); the (offset, length) of a node is recomputed by
// descending from the root along those bits.  (Rounds 1-2 kept four 28-entry stack arrays: 576 bytes of scratch memory
// per lane for every kernel that inlines np_pairwise, and a scratch access per step.)
// kPwDepth levels cover n <= 128 * 2^(kPwDepth-1) elements and more (pieces are at most n/2 + 8 long): 1 M elements.
constexpr int kPwDepth = 14;
constexpr int kPwMaxN = 1 << 20;

LFA_HD int np_pairwise_split(int n) { int n2 = n / 2; return n2 - (n2 % 8); }

struct SeqLeaf {
    template <class F>
    LFA_HD double operator()(const F& f, int off, int n) const { return np_pairwise_leaf(f, off, n); }
};

template <class F, class LEAF>
LFA_HD_NOINLINE double np_pairwise_big(const F& f, int off, int n, const LEAF& leaf) {
    double acc[kPwDepth];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < kPwDepth; ++k) acc[k] = 0.0;
    unsigned path = 0;                       // bit d: at depth d the walk is in the right child
    int depth = 0, o = off, m = n;
    for (;;) {
        while (m > 128 && depth < kPwDepth) { m = np_pairwise_split(m); ++depth; }     // down the left spine (bits stay 0)
        double r = leaf(f, o, m);
        while (depth > 0 && ((path >> (depth - 1)) & 1u)) {                            // a right child is done: left + right
            double l = 0.0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int k = 0; k < kPwDepth; ++k) l = (k == depth - 1) ? acc[k] : l;
            r = l + r;
            --depth;
        }
        if (depth == 0) return r;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int k = 0; k < kPwDepth; ++k) acc[k] = (k == depth - 1) ? r : acc[k];     // a left child is done: park it,
        path = (path & ((1u << (depth - 1)) - 1u)) | (1u << (depth - 1));              // go to its right sibling
        o = off; m = n;
        for (int d = 0; d < depth; ++d) {
            const int n2 = np_pairwise_split(m);
            if ((path >> d) & 1u) { o += n2; m -= n2; } else { m = n2; }
        }
    }
}

template <class F>
LFA_HD double np_pairwise(const F& f, int off, int n) {
    if (n <= 128) return np_pairwise_leaf(f, off, n);
    return np_pairwise_big(f, off, n, SeqLeaf());
}

struct ArrAt {
    const double* p;
    LFA_HD double operator()(int i) const { return p[i]; }
};

// np.trapz(y[0..n)) with dx = 1:  sum((1.0 * (y[1:] + y[:-1])) / 2.0)
struct TrapzAt {
    const double* y;
    LFA_HD double operator()(int i) const { return (1.0 * (y[i + 1] + y[i])) / 2.0; }
};

// np.linspace(start, stop, num)[i]   (numpy/core/function_base.py:145-167)
struct Linspace {
    double start, stop, step, delta;
    int num, div;
    bool step_zero;
    LFA_HD Linspace(double a, double b, int n) : start(a), stop(b), num(n) {
        div = n - 1;
        delta = b - a;
        step = (div > 0) ? delta / (double)div : 0.0;
        step_zero = (div > 0) && (step == 0.0);
    }
    LFA_HD double operator()(int i) const {
        if (num > 1 && i == num - 1) return stop;
        double y = (double)i;
        if (div > 0) {
            if (step_zero) { y = y / (double)div; y = y * delta; }
            else           { y = y * step; }
        } else {
            y = y * delta;
        }
        return y + start;
    }
};

struct TrapzLin {
    Linspace l;
    LFA_HD double operator()(int i) const { return (1.0 * (l(i + 1) + l(i))) / 2.0; }
};

LFA_HD bool lfa_isnan(double x) { return x != x; }

// np.argmax / np.argmin on float64 (first extremum; the first NaN wins)
LFA_HD int np_argext(const double* p, int n, bool want_min) {
    double m = p[0];
    int idx = 0;
    if (lfa_isnan(m)) return 0;
    for (int i = 1; i < n; ++i) {
        const double x = p[i];
        const bool upd = want_min ? !(x >= m) : !(x <= m);
        if (upd) {
            m = x; idx = i;
            if (lfa_isnan(m)) break;
        }
    }
    return idx;
}

struct Peak { int start, end; double area; };

// The two walks from the extremum to the feet of the peak (detection.py:291-300).  dir = -1: the largest s <= from with
// s == 0 or !(p[s] OP p[s-1] || p[s] CMP thr); dir = +1: the smallest e >= from with e == n-1 or !(p[e] OP p[e+1] || p[e] CMP thr)
// (OP, CMP = >=, > for peaks; <=, < for valleys).
LFA_HD bool walk_goes_on(const double* p, int n, int i, int dir, double thr, bool inverse) {
    if (dir < 0 ? (i <= 0) : (i >= n - 1)) return false;
    const double a = p[i], b = p[i + dir];
    return inverse ? (a <= b || a < thr) : (a >= b || a > thr);
}

struct SeqWalk {
    LFA_HD int operator()(const double* p, int n, int from, int dir, double thr, bool inverse) const {
        int i = from;
        while (walk_goes_on(p, n, i, dir, thr, inverse)) i += dir;
        return i;
    }
};

// detection.py:285-316 given the extremum index (detection.py:284)
struct SeqSum {
    template <class F>
    LFA_HD double operator()(const F& f, int off, int n) const { return np_pairwise(f, off, n); }
};

// `walk` and `sum` are policies: sequential on the host (SeqWalk, SeqSum), cooperative over the lanes of a group on the
// device (k_profile.hip GroupWalk, GroupSum) -- same values, same rounding order.
template <class WK, class SM>
LFA_HD Peak analyze_peaks_at_t(const double* p, int n, int peak, bool inverse, int window, const WK& walk, const SM& sum) {
    LFA_CK_BEGIN();
    int ws = peak - window; if (ws < 0) ws = 0;
    int we = peak + window + 1; if (we > n) we = n;
    const double thr = sum(ArrAt{p}, ws, we - ws) / (double)(we - ws);
    LFA_CK(0);
    const int s = walk(p, n, peak, -1, thr, inverse);
    const int e = walk(p, n, peak, +1, thr, inverse);
    LFA_CK(1);
    const int m = e - s + 1;
    const double seg = sum(TrapzAt{p + s}, 0, m - 1);
    LFA_CK(2);
    const double base = sum(TrapzLin{Linspace(p[s], p[e], m)}, 0, m - 1);
    LFA_CK(3);
    Peak r;
    r.start = s; r.end = e;
    r.area = inverse ? (base - seg) : (seg - base);
    return r;
}

LFA_HD Peak analyze_peaks_at(const double* p, int n, int peak, bool inverse, int window) {
    return analyze_peaks_at_t(p, n, peak, inverse, window, SeqWalk(), SeqSum());
}

struct SeqArgExt {
    LFA_HD int operator()(const double* p, int n, bool want_min) const { return np_argext(p, n, want_min); }
};

LFA_HD Peak analyze_peaks(const double* p, int n, bool inverse, int window) {
    return analyze_peaks_at(p, n, np_argext(p, n, inverse), inverse, window);
}

// Python slice p[:k] -> length
LFA_HD int py_head_len(int k, int n) {
    if (k < 0) { k += n; if (k < 0) k = 0; }
    if (k > n) k = n;
    return k;
}

struct TwoPeaks { int n; Peak pk[2]; };

// detection.py:221-245 (fallback = true, image_analysis) and :257-281 (fallback = false, inverse_image_analysis).
// `argext(p, n, want_min)` is np.argmax / np.argmin; on the device it is a wave-cooperative search and every lane of
// the wave runs this function with identical arguments.
template <class AE, class WK, class SM>
LFA_HD TwoPeaks two_peaks_t(const double* p, int n, bool inverse, int rng, bool fallback, int window, const AE& argext,
                            const WK& walk, const SM& sum) {
    TwoPeaks out;
    out.n = 0;
    if (n <= 0) return out;
    LFA_CK_BEGIN();
    const int i0 = argext(p, n, inverse);
    LFA_CK(4);
    const Peak a = analyze_peaks_at_t(p, n, i0, inverse, window, walk, sum);
    LFA_CK(5);
    out.n = 1; out.pk[0] = a; out.pk[1] = a;
    const int s = a.start, e = a.end;
    if (!(s > 1 && e < n - 2)) return out;
    int left_len, right_off;
    if (fallback) {
        left_len = (s - rng > 0) ? py_head_len(s - rng, n) : py_head_len(s - 1, n);
        right_off = (e + rng < n) ? (e + rng) : (e + 1);
    } else {
        left_len = py_head_len(s - rng, n);
        right_off = e + rng;
    }
    if (right_off < 0) { right_off += n; if (right_off < 0) right_off = 0; }   // python p[k:] with negative k
    if (right_off > n) right_off = n;
    const int right_len = n - right_off;
    if (!(left_len > 1 && right_len > 1)) return out;
    const double* right = p + right_off;
    const int il = argext(p, left_len, inverse);
    const int ir = argext(right, right_len, inverse);
    LFA_CK(6);
    const bool take_left = inverse ? (right[ir] > p[il]) : (right[ir] < p[il]);
    // one call site for both sides (the function is inlined on the device: every call site is a full copy)
    Peak b = analyze_peaks_at_t(take_left ? p : right, take_left ? left_len : right_len, take_left ? il : ir, inverse, window, walk, sum);
    if (!take_left) {
        // image_analysis adds `end_index + range` (detection.py:238-239, also after its [end_index + 1:] fallback slice);
        // inverse_image_analysis slices with `range` (:260-261) but adds the literal `end_index + 1` (:274-275)
        const int off = fallback ? e + rng : e + 1;
        b.start += off;
        b.end += off;
    }
    LFA_CK(7);
    const int s2 = b.start, e2 = b.end;
    if ((s2 <= s && e2 > s) || (s2 <= e && e2 > e) || (s2 >= s && e2 <= e)) return out;
    out.n = 2; out.pk[1] = b;
    return out;
}

LFA_HD TwoPeaks two_peaks(const double* p, int n, bool inverse, int rng, bool fallback, int window) {
    return two_peaks_t(p, n, inverse, rng, fallback, window, SeqArgExt(), SeqWalk(), SeqSum());
}

// ---------------------------------------------------------------------------------------------------------------
// float Otsu on 256 bins (thresholding.py:364-381 with exposure.py:298-301): returns argmax index, writes centre
// ---------------------------------------------------------------------------------------------------------------
struct OtsuScratch { float w1[256]; float w2[256]; double m1[256]; double m2[256]; };

LFA_HD int otsu_float256(const uint32_t* cnt, double vmin, double vmax, double* centre_out, OtsuScratch* ws) {
    const int NB = 256;
    const Linspace edges(vmin, vmax, NB + 1);
    float* w1 = ws->w1; float* w2 = ws->w2;
    double* m1 = ws->m1; double* m2 = ws->m2;
    {
        float acc = 0.0f;
        for (int i = 0; i < NB; ++i) { acc = acc + (float)cnt[i]; w1[i] = acc; }
        acc = 0.0f;
        for (int i = NB - 1; i >= 0; --i) { acc = acc + (float)cnt[i]; w2[i] = acc; }
        double s = 0.0;
        for (int i = 0; i < NB; ++i) {
            const double c = (edges(i) + edges(i + 1)) / 2.0;
            s = s + (double)(float)cnt[i] * c;
            m1[i] = s / (double)w1[i];
        }
        s = 0.0;
        for (int i = NB - 1; i >= 0; --i) {
            const double c = (edges(i) + edges(i + 1)) / 2.0;
            s = s + (double)(float)cnt[i] * c;
            m2[i] = s / (double)w2[i];
        }
    }
    int best = 0;
    double bv = 0.0;
    for (int i = 0; i < NB - 1; ++i) {
        const float ww = w1[i] * w2[i + 1];
        const double d = m1[i] - m2[i + 1];
        const double v = (double)ww * (d * d);
        if (i == 0) { bv = v; best = 0; if (lfa_isnan(v)) break; }
        else if (!(v <= bv)) { bv = v; best = i; if (lfa_isnan(v)) break; }
    }
    *centre_out = (edges(best) + edges(best + 1)) / 2.0;
    return best;
}

// ---------------------------------------------------------------------------------------------------------------
// threshold_otsu on an INTEGER image with more than 2^24 pixels (thresholding.py:309, :365-381).  The reference casts the
// per-value counts to float32 and accumulates them in float32, one bin after the other; beyond 2^24 those sums round,
// and the result depends on that order.  The histogram here is over input values x; the reference's is over gamma
// values v = lut[x], so the count of a gamma bin is the (exact) sum over the run of x that share v, cast to float32
// once, at the end of the run.  `lv[b]` is the gamma value of bin b; a run ends where lv changes.
//   forward : w[b] = weight1 and s[b] = cumsum(counts*centers) after the run containing b ... valid at run ends
//   backward: the same from the top, valid at run starts
// The products fl32(count) * v are integers below 2^53, so the float64 sums are exact whatever their order.
// ---------------------------------------------------------------------------------------------------------------
struct OtsuSeq { float a; double sa; unsigned long long gc; };

// cnt[0..n), lv[0..n]: lv[n] is the gamma value of the bin after this tile (or -1 after the last bin)
LFA_HD void otsu_seq_fwd(const uint32_t* cnt, const int32_t* lv, int n, OtsuSeq& st, float* w, double* s) {
    for (int b = 0; b < n; ++b) {
        st.gc += cnt[b];
        if (lv[b + 1] != lv[b]) {
            const float c = (float)st.gc;
            st.a = st.a + c;
            st.sa = st.sa + (double)c * (double)lv[b];
            st.gc = 0;
        }
        w[b] = st.a; s[b] = st.sa;
    }
}

// cnt[0..n), lvm[0..n]: lvm[b + 1] is the gamma value of bin b, lvm[0] that of the bin before this tile (or -1 before bin 0)
LFA_HD void otsu_seq_bwd(const uint32_t* cnt, const int32_t* lvm, int n, OtsuSeq& st, float* w, double* s) {
    for (int b = n - 1; b >= 0; --b) {
        st.gc += cnt[b];
        if (lvm[b] != lvm[b + 1]) {
            const float c = (float)st.gc;
            st.a = st.a + c;
            st.sa = st.sa + (double)c * (double)lvm[b + 1];
            st.gc = 0;
        }
        w[b] = st.a; s[b] = st.sa;
    }
}

// between-class variance of the split after bin b (a run end): weight1/mean1 at b, weight2/mean2 at b + 1
LFA_HD double otsu_seq_variance(float w1, double s1, float w2, double s2) {
    const double m1 = s1 / (double)w1;
    const double m2 = s2 / (double)w2;
    const float ww = w1 * w2;
    const double d = m1 - m2;
    return (double)ww * (d * d);
}

}  // namespace lfa
#endif  // LFA_ALGO_H_
