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
#include <cstdlib>

#include "lfa_internal.h"
#include "lfa_algo.h"
#include "lfa_device.h"
#include <type_traits>

namespace lfa {

// ------------------------------------------------------------------------------------------------------------------
// init
// ------------------------------------------------------------------------------------------------------------------
// One launch resets everything a call accumulates into: the per-image info records, the 256-bin histograms (uint8 and
// colour images; the uint16 histogram is cleared over its occupied range by k_zero_hist_u16 once min/max are known) and
// the per-row root counters of the labelling stage.  Grid: (blocks, n); block x of image y strides over the words.
__global__ __launch_bounds__(256) void k_init_info(Job j) {
    const int img = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (t < kStatsWords && img == 0 && j.plane_stats) j.plane_stats[t] = 0u;
    if (t == 0) {
        LfaImageInfo z;
        z.xmin = 0x7fffffff; z.xmax = -1; z.v_thr = 0; z.x_thr = 0x7fffffff;
        z.gmin = __longlong_as_double(0x7ff0000000000000LL); z.gmax = 0.0;
        z.vmin = 0.0; z.vmax = 0.0; z.v_thr_f = 0.0; z.g_thr = __longlong_as_double(0x7ff0000000000000LL);
        z.otsu_bin = -1; z.all_equal = 0; z.ambiguous = 0;
        z.n_runs = 0; z.n_labels = 0; z.label00 = 0; z.n_candidates = 0; z.n_accepted = 0; z.n_strips = 0; z.status = 0;
        j.info[img] = z;
        j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Grey] = (j.C == 3 && j.dtype == LFA_U8 && j.aligned16) ? 1u : 0u;
        j.edges_i[(int64_t)img * kEdgesI + kEdgesI_PredCnt] = 0u;
        if (j.thr_list) j.thr_list[img] = 0u;
    }
    if (j.hist_stride == 256) {
        uint32_t* gh = j.hist + (int64_t)img * 256;
        for (int k = t; k < 256; k += nt) gh[k] = 0;
    }
    uint32_t* rc = j.rowcnt + (int64_t)img * (j.H + 1);
    for (int k = t; k <= j.H; k += nt) rc[k] = 0;
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

// ------------------------------------------------------------------------------------------------------------------
// colour path: histogram edges in gray space
// ------------------------------------------------------------------------------------------------------------------
__device__ inline double gamma_pow(double g) { return pow(g, 0.4); }

// smallest double g in [gmin, gmax] with pred(g) true, where pred is monotone (false ... false true ... true);
// +inf if pred(gmax) is false.  Bisection on the bit patterns (non-negative doubles are ordered like int64).
// `guess` only speeds the search up (gallop outwards from it in steps of 16^k bit patterns, then bisect); any value works.
template <class Pred>
__device__ inline double first_true(double gmin, double gmax, double guess, const Pred& pred) {
    if (!pred(gmax)) return __longlong_as_double(0x7ff0000000000000LL);
    if (pred(gmin)) return gmin;
    long long lo = __double_as_longlong(gmin), hi = __double_as_longlong(gmax);   // pred(lo) false, pred(hi) true
    if (hi - lo > 1 && guess == guess) {
        long long g0 = __double_as_longlong(guess < 0.0 ? 0.0 : guess);
        g0 = g0 <= lo ? lo + 1 : (g0 >= hi ? hi - 1 : g0);
        long long step = 16;
        if (pred(__longlong_as_double(g0))) {
            hi = g0;
            while (hi - step > lo) {
                const long long c = hi - step;
                if (pred(__longlong_as_double(c))) { hi = c; if (step < (1LL << 56)) step <<= 4; } else { lo = c; break; }
            }
        } else {
            lo = g0;
            while (lo + step < hi) {
                const long long c = lo + step;
                if (!pred(__longlong_as_double(c))) { lo = c; if (step < (1LL << 56)) step <<= 4; } else { hi = c; break; }
            }
        }
    }
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (pred(__longlong_as_double(mid))) hi = mid; else lo = mid;
    }
    return __longlong_as_double(hi);
}

// ---- integer form of a gray-space threshold ------------------------------------------------------------------------------
// gray(b0,b1,b2) is within a few ulps of G / 2 550 000 with the INTEGER G = 2125*b0 + 7154*b1 + 721*b2 (0.2125, 0.7154,
// 0.0721 scaled by 10^4, times 255).  Different G differ by >= 4e-7 relative, the float64 rounding of any of the
// summation orders by < 4e-15, so `gray >= t` equals `G >= ceil(t * 2 550 000)` unless an integer lies within a
// 2^-39 relative band around t * 2 550 000 -- then the threshold is "critical" and the image takes the float64 path.
// The band also covers the 2^-40 ambiguity census, so an image on the integer path has no ambiguous pixel by construction.
constexpr double kGScale = 2550000.0;
constexpr uint32_t kGInf = 0xffffffffu;

__device__ inline uint32_t gray_G(uint32_t p0, uint32_t p1, uint32_t p2) { return 2125u * p0 + 7154u * p1 + 721u * p2; }

__device__ inline uint32_t g_threshold(double t, bool* critical) {
    if (!(t < 1.0e300)) return kGInf;                      // unreachable edge
    const double A = t * kGScale;
    const double r = rint(A);
    if (fabs(r - A) <= A * 1.8189894035458565e-12) *critical = true;     // 2^-39
    const double c = ceil(A);
    return c >= 4294967295.0 ? kGInf : (uint32_t)c;
}

// 256 threads = one wave per SIMD at 72 VGPRs: that is what still fits on a CU beside the other chunk stream's histogram
// blocks, so this kernel starts at once instead of waiting for them to retire -- with 320 threads it lasted 270 us on
// average in the pipeline, 10 us alone.  Thread 0 also takes edge 256.
// (Computing the edges at the end of k_minmax_rgb instead, in whichever block of an image delivers its extremes last --
// ticket by atomics, no fence -- was bit-exact and took this launch off the critical path, but the bisection then shares
// its SIMDs with the pass's own waves: the pass grew by 48-56 us for the 12-28 us saved.  Round 2, not kept.)
__global__ __launch_bounds__(256) void k_edges_rgb(Job j) {
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    uint32_t* egf = j.edges_i + (int64_t)img * kEdgesI;
    // Grey mode: j.hist holds byte counts from k_grey_hist -- of a grey image all of them, of any other image a part.  They
    // leave the buffer here (the histogram pass adds into it); for a grey image the extremes of gray follow from the
    // smallest and the largest byte present, and the 256-bin histogram is built below, once the edges are known.
    __shared__ uint32_t s_cnt[256];
    __shared__ int s_b[2];
    __shared__ double s_t[kEdges];
    __shared__ uint32_t s_nh[256];
    const bool grey = j.grey_mode && egf[kEdgesI_Grey] != 0u;
    if (j.grey_mode && j.pred_mode && !grey && threadIdx.x == 0) egf[kEdgesI_PredCnt] = 0u;   // k_hist_rgb lists this image's groups itself
    if (j.grey_mode) {
        uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
        if (threadIdx.x == 0) { s_b[0] = 255; s_b[1] = 0; }
        __syncthreads();
        const uint32_t c = gh[threadIdx.x];
        s_cnt[threadIdx.x] = c;
        s_nh[threadIdx.x] = 0u;
        gh[threadIdx.x] = 0u;
        if (grey && c) { atomicMin(&s_b[0], (int)threadIdx.x); atomicMax(&s_b[1], (int)threadIdx.x); }
        __syncthreads();
        if (grey && threadIdx.x == 0) {
            inf->gmin = gray_u8_rt(j.gray_variant, s_b[0], s_b[0], s_b[0]);
            inf->gmax = gray_u8_rt(j.gray_variant, s_b[1], s_b[1], s_b[1]);
        }
    }
    const double gmin = grey ? gray_u8_rt(j.gray_variant, s_b[0], s_b[0], s_b[0]) : inf->gmin;
    const double gmax = grey ? gray_u8_rt(j.gray_variant, s_b[1], s_b[1], s_b[1]) : inf->gmax;
    // lfa_set_host_pow: vmin, vmax and the gray-space edges come from the caller's record -- no pow() is evaluated here
    const double* hp = j.host_pow ? j.host_pow + (int64_t)img * kHostPowDoubles : nullptr;
    const double vmin = hp ? hp[0] : gamma_pow(gmin), vmax = hp ? hp[1] : gamma_pow(gmax);
    uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
    if (threadIdx.x == 0) { inf->vmin = vmin; inf->vmax = vmax; inf->all_equal = (vmin == vmax) ? 1 : 0; if (hp) inf->status |= 4; }
    bool anycrit = false;
    for (int i = threadIdx.x; i < kEdges; i += 256) {
        bool crit = false;
        double* t = j.edges_g + (int64_t)img * kEdges;
        double ti;
        if (hp) {
            ti = hp[2 + i];
        } else if (vmin == vmax) {
            ti = (i == 0) ? gmin : __longlong_as_double(0x7ff0000000000000LL);
        } else {
            const Linspace lin(vmin, vmax, kEdges);
            const double e = lin(i);
            ti = first_true(gmin, gmax, pow(e, 2.5), [e](double g) { return gamma_pow(g) >= e; });
            if ((j.debug_flags & (256 | 512)) && i >= 1 && i <= 255 && ti < 1.0e300 && ti > 0.0)   // test hook: as if pow() differed in its last bit
                ti = __longlong_as_double(__double_as_longlong(ti) + ((j.debug_flags & 256) ? 1 : -1));
        }
        t[i] = ti;
        s_t[i] = ti;
        eg[i] = (i == 0) ? 0u : g_threshold(ti, &crit);       // edge 0 (= gmin) is below every pixel
        anycrit |= (i >= 1 && i <= 255 && crit);
    }
    // (uint16 colour and float64 pixels: the integer form is built for 8-bit channels, they always take the float64 path)
    const int any = __syncthreads_or((anycrit || (j.debug_flags & 1) || j.dtype != LFA_U8) ? 1 : 0);
    if (threadIdx.x == 0) eg[kEdgesI_Crit] = (uint32_t)any;
    if (grey && vmin != vmax) {
        // the histogram pass of a grey image, 256 values instead of H*W pixels: byte b lands in the largest bin k with
        // edge[k] <= gray(b) (the last bin closed) -- the float64 compare k_hist_rgb makes per pixel, ambiguity census included
        const uint32_t c = s_cnt[threadIdx.x];
        if (c) {
            const double g = gray_u8_rt(j.gray_variant, threadIdx.x, threadIdx.x, threadIdx.x);
            int lo = 0, hi = 255;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                const bool ge = s_t[mid] <= g;
                lo = ge ? mid : lo;
                hi = ge ? hi : mid - 1;
            }
            atomicAdd(&s_nh[lo], c);
            const int a = lo < 1 ? 1 : lo, b = lo + 1 <= 255 ? lo + 1 : 255;
            if (near_rel(g, s_t[a]) || near_rel(g, s_t[b])) atomicAdd(&inf->ambiguous, c);
        }
        __syncthreads();
        j.hist[(int64_t)img * j.hist_stride + threadIdx.x] = s_nh[threadIdx.x];
    }
}

// ---- images with three equal channels (what cv2.imread hands the reference for a grey file, detection.py:454) ------------
// != 0 iff one of the 16 pixels in w has unequal channel bytes.  z = stream ^ (stream shifted by one byte) has byte j =
// s[j] ^ s[j+1]; a grey pixel needs it zero at j = 3k and 3k + 1, position 3k + 2 compares two different pixels and is
// masked out (the pattern repeats every three dwords).  Three instructions per dword.
__device__ __forceinline__ uint32_t not_grey16(const uint32_t (&w)[12]) {
    uint32_t ng = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const uint32_t nxt = __builtin_amdgcn_alignbyte(w[i + 1 < 12 ? i + 1 : i], w[i], 1);
        const uint32_t mask = (i % 3 == 0) ? 0xff00ffffu : (i % 3 == 1 ? 0xffff00ffu : 0x00ffff00u);
        ng |= (w[i] ^ nxt) & mask;
    }
    return ng;
}

#ifndef LFA_HIST_COPIES
#define LFA_HIST_COPIES 16
#endif
constexpr int kHC = LFA_HIST_COPIES;

// Grey mode (Job::grey_mode: most colour images of the handle's last call had three equal channels): ONE pass takes the
// byte histogram of channel 0 and checks every pixel's channels.  The gray of a grey pixel is a function of its byte b alone
// -- gray(b) = the host's rounding order applied to (b/255, b/255, b/255), increasing in b -- so the extremes are gray(bmin),
// gray(bmax) and the 256-bin histogram of gray^0.4 is the byte histogram pushed through the edges: k_edges_rgb does both from
// these 256 counts, and the min/max pass and the histogram pass skip the image (3 bytes per pixel read once instead of
// twice: photo-like batches 6.2 -> 4.9 ms per 1024 images).  An image with a single unequal pixel clears its flag and takes
// the two ordinary passes; its partial byte counts are thrown away by k_edges_rgb.  Same read shape as k_minmax_rgb.
__global__ __launch_bounds__(256) void k_grey_hist(Job j) {
    __shared__ uint32_t sh[256 * kHC];
    __shared__ lfa_u32x4 turn[4][192];
    const int img = blockIdx.y;
    for (int k = threadIdx.x; k < 256 * kHC; k += 256) sh[k] = 0;
    __syncthreads();
    uint32_t* shl = sh + (threadIdx.x & (kHC - 1));
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t ng = ((int64_t)j.H * j.W) >> 4;          // groups of 16 pixels = 48 bytes (aligned16: the image is whole groups)
    const int lane = threadIdx.x & 63, wv4 = threadIdx.x >> 6;
    const int64_t rounds = (ng + 63) >> 6;
    const int64_t per_block = (rounds + gridDim.x - 1) / gridDim.x;
    const int64_t r_end = min(rounds, (int64_t)(blockIdx.x + 1) * per_block);
    const int64_t n16 = ng * 3;
    auto piece = [&](int64_t r, int k) -> const uint8_t* {           // past the image's end: the round's first piece again
        const int64_t i = r * 192 + 64 * k + lane;
        return base + (i < n16 ? i : r * 192) * 16;
    };
    uint32_t notgrey = 0;
    uint8_t* gplane = j.grey_plane ? j.grey_plane + (int64_t)img * j.grey_plane_stride : nullptr;
    // loads, the plane's store and the waits issued by hand, as in k_hist_rgb (lfa_device.h): the next round's bytes are
    // requested before this round is worked on, and the store of a round stays in flight across the loop boundary -- with
    // compiler-placed waits (vmcnt(0) at the boundary: the store's round trip included) the pass took 0.63 ms per 256
    // images instead of 0.37 without a plane.  Every round has lane 0 inside the image, so the store is one operation of the
    // wave in every round, which the counted wait relies on.
    // kind 0: byte histogram only; 1: also the byte plane; 2: also the predicted-threshold bit plane and the list of undecided groups
    // (Job::pred_mode, as k_hist_rgb writes them: a grey pixel has G = 10000 * b, so `G >= hi` is `b >= ceil(hi / 10000)` -- against the
    // window's ends before they were widened to whole buckets, pred_lo_raw / pred_hi_raw; an image that turns out not to be grey gets
    // its bits from k_hist_rgb instead, k_edges_rgb drops what this pass listed for it)
    uint16_t* pbits = (uint16_t*)(j.bins + (int64_t)img * (((int64_t)j.H * j.W) >> 3));
    uint32_t* pcnt = j.edges_i + (int64_t)img * kEdgesI + kEdgesI_PredCnt;
    uint32_t* plist = j.pred_list + (int64_t)img * j.pred_stride;
    const uint32_t pcap = (uint32_t)j.pred_cap, b_lo = (j.pred_lo_raw + 9999u) / 10000u, b_hi = (j.pred_hi_raw + 9999u) / 10000u;
    auto pass = [&](auto kind) {
        constexpr int kKind = decltype(kind)::value;
        constexpr bool kPlane = kKind != 0;
        Raw48 nx;
        int64_t r = (int64_t)blockIdx.x * per_block + wv4;
        if (r >= r_end) return;
        vm_request3nt(piece(r, 0), piece(r, 1), piece(r, 2), nx);
        vm_wait48<0>(nx);
        for (; r < r_end; r += 4) {
            turn[wv4][lane] = nx.a; turn[wv4][64 + lane] = nx.b; turn[wv4][128 + lane] = nx.c;
            const int64_t rn = (r + 4 < r_end) ? r + 4 : r;
            vm_request3nt(piece(rn, 0), piece(rn, 1), piece(rn, 2), nx);
            uint32_t w[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const lfa_u32x4 v = turn[wv4][3 * lane + k];
                w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
            }
            if (r * 64 + lane < ng) {
                notgrey |= not_grey16(w);
#pragma unroll
                for (int k = 0; k < 16; ++k) atomicAdd(&shl[byte_of(w, 3 * k) * kHC], 1u);
                if (kKind == 2) {
                    uint32_t set = 0, und = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t b = byte_of(w, 3 * k);
                        set |= (b >= b_hi ? 1u : 0u) << k;
                        und |= (b >= b_lo && b < b_hi) ? 1u : 0u;
                    }
                    const unsigned long long um = __ballot(und != 0u);   // (over the lanes inside the image)
                    if (um) {
                        const int leader = __ffsll(um) - 1;
                        uint32_t slot0 = 0;
                        if (lane == leader) slot0 = atomicAdd(pcnt, (uint32_t)__popcll(um));
                        slot0 = (uint32_t)__shfl((int)slot0, leader);
                        const uint32_t slot = slot0 + (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
                        if (und != 0u && slot < pcap) plist[slot] = (uint32_t)(r * 64 + lane);
                    }
                    vm_store2(pbits + (r * 64 + lane), set);
                }
                if (kKind == 1) {
                    // channel 0 of the 16 pixels: stream bytes 0, 3, 6, ... -- two v_perm per four pixels -- as one 16-byte store
                    lfa_u32x4 o;
                    uint32_t t;
                    t = __builtin_amdgcn_perm(w[1], w[0], 0x0c060300u); o.x = __builtin_amdgcn_perm(w[2], t, 0x05020100u);    // w0.b0, w0.b3, w1.b2 | w2.b1
                    t = __builtin_amdgcn_perm(w[4], w[3], 0x0c060300u); o.y = __builtin_amdgcn_perm(w[5], t, 0x05020100u);
                    t = __builtin_amdgcn_perm(w[7], w[6], 0x0c060300u); o.z = __builtin_amdgcn_perm(w[8], t, 0x05020100u);
                    t = __builtin_amdgcn_perm(w[10], w[9], 0x0c060300u); o.w = __builtin_amdgcn_perm(w[11], t, 0x05020100u);
                    vm_store16(gplane + (r * 64 + lane) * 16, o);
                }
            }
            vm_wait48<kPlane ? 1 : 0>(nx);
        }
    };
    if (j.pred_mode && j.bins) pass(std::integral_constant<int, 2>{});
    else if (gplane) pass(std::integral_constant<int, 1>{});
    else pass(std::integral_constant<int, 0>{});
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int c = 0; c < kHC; ++c) tot += sh[threadIdx.x * kHC + ((c + threadIdx.x) & (kHC - 1))];
    if (tot) atomicAdd(&j.hist[(int64_t)img * j.hist_stride + threadIdx.x], tot);
    if (notgrey) j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Grey] = 0u;
}

// Pixels with a smaller integer G = 2125*b0 + 7154*b1 + 721*b2 have a strictly smaller float64 gray (different G are
// >= 3.9e-7 apart in gray, the rounding of any summation order is < 4e-15), so the extremes of gray lie among the pixels
// with the extreme G.  Per 16-pixel group the integer G of every pixel is computed (two v_dot4 each) and compared with
// the lane's running extremes; the float64 gray is evaluated only for a pixel that sets or ties one of them with a byte
// triple not seen there before (equal G from different triples can differ in the last bits of gray: fmin / fmax decide).
// A lane updates its extremes a handful of times over its whole share of the image, so the pass costs the integer work.
//
// MinMaxLane: a lane's record of the extreme it has seen (G, byte triple, float gray) and the wave-wide bounds Wlo / Whi (the
// best G any lane of the wave holds), which screen the pixels: one that does not reach them cannot be an extreme of the image.
template <int V>
struct MinMaxLane {
    double lo, hi;
    uint32_t Glo, Ghi, tlo, thi, Wlo, Whi, Alo, Ahi, notgrey;
    __device__ __forceinline__ void init() {
        lo = __longlong_as_double(0x7ff0000000000000LL); hi = 0.0;
        Glo = 0x7fffffffu; Ghi = 0u; tlo = 0xffffffffu; thi = 0xffffffffu; Wlo = 0x7fffffffu; Whi = 0u; Alo = 0xffffffffu; Ahi = 0u; notgrey = 0u;
    }
    // One round: w = the lane's 16 pixels (48 bytes), t48 = the same twelve dwords in LDS (the wave's turn buffer).  ALL 64
    // lanes of the wave must be here (wave reductions).
    __device__ __forceinline__ void round(const uint32_t (&w)[12], const uint32_t* t48, bool look_grey) {
        auto pixel = [&](int k) -> uint32_t {             // bytes 3k .. 3k+2 of the group at the low end of a dword (k constant after unrolling)
            const int i = (3 * k) >> 2, sft = (3 * k) & 3;
            return sft == 0 ? w[i] : (sft == 1 ? (w[i] >> 8) : __builtin_amdgcn_alignbyte(w[i + 1 < 12 ? i + 1 : i], w[i], sft));
        };
        if (look_grey) notgrey |= not_grey16(w);
        // Screen with ONE dot product per pixel: g = 76*b0 + 255*b1 + 26*b2 has G - 28*g = -3*b0 + 14*b1 - 7*b2, i.e.
        // 28*g - 2550 <= G <= 28*g + 3570 for every byte triple, so a pixel with G <= Wlo has g <= (Wlo + 2550) / 28 and one with
        // G >= Whi has g >= (Whi - 3570) / 28 (Alo / Ahi, kept beside the bounds).  The group's smallest and largest g come from
        // eight v_min3 / v_max3; the few pixels that pass for nothing (within 0.6 grey levels of an extreme) are sorted out by the
        // exact walk below.  2.75 instead of 6.75 vector instructions per pixel.
        uint32_t gmn = 0xffffffffu, gmx = 0u;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const uint32_t ga = __builtin_amdgcn_udot4(pixel(k), 0x001AFF4Cu, 0u, false);       // (the fourth byte weighs 0)
            const uint32_t gb = __builtin_amdgcn_udot4(pixel(k + 1), 0x001AFF4Cu, 0u, false);
            gmn = min(gmn, min(ga, gb));
            gmx = max(gmx, max(ga, gb));
        }
        if (!__any(gmn <= Alo || gmx >= Ahi)) return;     // the usual case: nothing in the wave's 1024 pixels reaches a bound
        // Some pixel reaches a bound: it matters only with a byte triple other than the one this lane already holds there
        // (the same triple again -- a black or saturated area -- changes nothing).  Rare, so this part is written for few
        // registers, not for speed (the screen above runs all the time and its occupancy is what counts): the pixels are
        // walked one by one out of the lane's 48 bytes in LDS.
        bool changed = false;
        auto one = [&](uint32_t t) {
            const uint32_t Gk = gray_G_from_dword(t);
            const bool nlo = Gk <= Wlo && Gk <= Glo && t != tlo, nhi = Gk >= Whi && Gk >= Ghi && t != thi;
            if (nlo || nhi) {
                const double gr = gray_u8<V>(t & 0xffu, (t >> 8) & 0xffu, t >> 16);
                if (nlo) { lo = (Gk < Glo) ? gr : fmin(lo, gr); Glo = Gk; tlo = t; }
                if (nhi) { hi = (Gk > Ghi) ? gr : fmax(hi, gr); Ghi = Gk; thi = t; }
                changed = true;
            }
        };
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
            const int b0 = 3 * k;
            const uint32_t lo32 = t48[b0 >> 2], hi32 = t48[min((b0 >> 2) + 1, 11)];
            one((uint32_t)((((uint64_t)hi32 << 32) | lo32) >> (8 * (b0 & 3))) & 0xffffffu);
        }
        if (!__any(changed)) return;
        Wlo = (uint32_t)wave_min_i32((int)Glo);               // (G <= 2 550 000 orders the same signed or unsigned; a lane that has
        Whi = (uint32_t)wave_max_i32((int)Ghi);               //  seen nothing yet reports INT_MAX / 0)
        Alo = (Wlo + 2550u) / 28u;                            // (Wlo <= INT_MAX: no overflow)
        Ahi = Whi > 3570u ? (Whi - 3570u + 27u) / 28u : 0u;
    }
    // The block's extremes into the image's record (bit patterns of non-negative doubles order like the doubles).  Every thread
    // of the 256-thread block must call it; slo / shi: four doubles of LDS each.
    __device__ __forceinline__ void commit(double* slo, double* shi, unsigned long long* gmin_bits, unsigned long long* gmax_bits) {
        lo = wave_min_f64(lo); hi = wave_max_f64(hi);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { slo[wv] = lo; shi[wv] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k) { lo = fmin(lo, slo[k]); hi = fmax(hi, shi[k]); }
            atomicMin(gmin_bits, (unsigned long long)__double_as_longlong(lo));
            atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(hi));
        }
    }
};

template <int V>
__global__ __launch_bounds__(256) void k_minmax_rgb(Job j) {
    const int img = blockIdx.y;
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    uint32_t* grey_flag = j.edges_i + (int64_t)img * kEdgesI + kEdgesI_Grey;
    if (j.grey_mode && *grey_flag) return;            // k_grey_hist found three equal channels throughout: k_edges_rgb has the extremes
    MinMaxLane<V> m;                                  // (otherwise this pass looks out for such images on the way, for the handle's next call)
    m.init();
    if (j.aligned16) {
        const int64_t ng = npix >> 4;                 // groups of 16 pixels = 48 bytes
        const int lane = threadIdx.x & 63, wv4 = threadIdx.x >> 6;
        // Memory shape (tools/read_probe.hip): a lane that loads its own 48 bytes as three 16-byte pieces reads at 6.2 TB/s; fully
        // coalesced 16-byte nontemporal loads over a contiguous run per block read at 7.1.  So a wave takes a round of 64 groups
        // (3 KiB) as three coalesced loads -- lane l gets pieces l, 64 + l, 128 + l -- and turns them through its own 3 KiB of
        // LDS into 48 contiguous bytes per lane (both sides conflict-free: 16-byte accesses at strides of 16 and 48 bytes).
        // Block b owns a contiguous run of rounds of its image, its four waves take them in turn.
        __shared__ lfa_u32x4 turn[4][192];
        // (32-bit round and piece numbers -- an image has fewer than 2^32 16-byte pieces --, so that a request is three adds and
        //  compares, not three 64-bit multiply-adds, between the arrival of one round's bytes and the next round's loads)
        const int rounds = (int)((ng + 63) >> 6);
        const int per_block = (rounds + (int)gridDim.x - 1) / (int)gridDim.x;
        const int r_end = min(rounds, (int)(blockIdx.x + 1) * per_block);
        const lfa_u32x4* img16 = (const lfa_u32x4*)base;
        const uint32_t n16 = (uint32_t)(ng * 3);
        auto request = [&](int r, lfa_u32x4 (&q)[3]) {
            const uint32_t i0 = (uint32_t)r * 192u;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t i = i0 + 64u * k + lane;
                q[k] = __builtin_nontemporal_load(img16 + (i < n16 ? i : i0));      // past the image's end: the round's first piece again
            }
        };
        lfa_u32x4 nx[3];                                          // (two rounds ahead measured no better: 1.54 vs 1.50 ms per 1024 images, twice)
        int r = (int)blockIdx.x * per_block + wv4;
        if (r < r_end) request(r, nx);
        for (; r < r_end; r += 4) {                               // wave-uniform trip count: the reductions below need every lane
#pragma unroll
            for (int k = 0; k < 3; ++k) turn[wv4][64 * k + lane] = nx[k];
            if (r + 4 < r_end) request(r + 4, nx);                // the next round's bytes travel while this round computes
            const int mine48 = ((int64_t)r * 64 + lane < ng) ? lane : 0;   // a lane past the end works on the round's first group again: changes nothing
            uint32_t w[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const lfa_u32x4 v = turn[wv4][3 * mine48 + k];
                w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
            }
            // (the grey look is a hint for the handle's next call, so one round in four is looked at: +1 % on this pass instead of
            //  +4 %; the equal-channel pass itself checks every pixel)
            m.round(w, (const uint32_t*)&turn[wv4][3 * mine48], !j.grey_mode && ((r >> 2) & 3) == 0);
        }
        done = ng << 4;
        if (m.notgrey) *grey_flag = 0u;
    }
    // uint16 colour and float64 images: this scalar loop only (Job::aligned16 is 0 for them)
    for (int64_t p = done + tid; p < npix; p += nth) {
        const double gr = gray_at<V>(base, p, j.dtype, j.C);
        m.lo = fmin(m.lo, gr); m.hi = fmax(m.hi, gr);
    }
    __shared__ double slo[4], shi[4];
    m.commit(slo, shi, (unsigned long long*)&j.info[img].gmin, (unsigned long long*)&j.info[img].gmax);
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


// KIND 0: histogram only; 1: also the 4-bit class plane (Job::u16_plane); 2: also the predicted-threshold bit plane and the list of
// undecided groups (Job::pred_mode) -- three kernels, so that each keeps only its own pointers and bounds in scalar registers (the
// one kernel for all three needed 102 of them: seven waves per SIMD, i.e. ONE 1024-thread block per CU instead of two)
template <int KIND>
__global__ __launch_bounds__(1024) void k_hist_u16(Job j) {
    __shared__ uint32_t sh[kLdsBinsU16];
    const int img = blockIdx.y;
    // one-pass mode (Job::hist_abs): bins are the pixel values themselves, no range known or needed beforehand
    const int xmin = j.hist_abs ? 0 : j.info[img].xmin, xmax = j.hist_abs ? 65535 : j.info[img].xmax;
    if (xmax <= xmin) return;                       // constant image: no histogram needed (all-equal shortcut)
    const int nb = xmax - xmin + 1;
    // The first nl bins are counted in LDS, the rest with global atomics.  nl is a power of two and the LDS words left
    // over hold copies of the table (lane l uses copy l & (copies-1)): pixels of a frame crowd on a few grey levels, and
    // lanes adding to the same word serialise.  Two-pass form: nl covers the occupied range; one-pass form: nl covers
    // the largest value of the handle's last call (Job::u16_lds_bins), larger values still count exactly, only slower.
    int want = j.hist_abs ? j.u16_lds_bins : nb;
    want = want < 256 ? 256 : (want > kLdsBinsU16 ? kLdsBinsU16 : want);
    int lg = 8;
    while ((1 << lg) < want) ++lg;
    const int nl = 1 << lg;                         // 256 .. 16384
    const int cshift = 14 - lg;                     // copies = 16384 / nl
    const uint32_t cmask = (1u << cshift) - 1u;
    const uint32_t cp = threadIdx.x & cmask;
    const int nlds = nb < nl ? nb : nl;
    for (int k = threadIdx.x; k < kLdsBinsU16; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const uint16_t* base = (const uint16_t*)(j.images + (int64_t)img * j.img_bytes);
    uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    int olo = 0x7fffffff, ohi = -1;                              // occupied range seen by this thread (bins; one-pass form)
    auto count = [&](int a) {
        if (a < nl) atomicAdd(&sh[((uint32_t)a << cshift) | cp], 1u);
        else { atomicAdd(&gh[a], 1u); ohi = max(ohi, a); olo = min(olo, a); }
    };
    if (j.aligned16) {
        const int64_t ng = npix >> 4;                            // 16 pixels = 32 bytes per lane and step
        const uint4* v = (const uint4*)base;
        // class plane (Job::u16_plane, one-pass form only): 4 bits per pixel, pixel p in nibble p & 7 of dword p >> 3 --
        // the layout k_mask reads for colour images; 8 bytes per lane and step
        uint2* plane = KIND == 1 ? (uint2*)(j.bins + (((int64_t)img * npix) >> 1)) : nullptr;
        const uint32_t pa = (uint32_t)j.u16_pa;
        const int ps = j.u16_ps;
        // Predicted-threshold mode (Job::pred_mode, as for colour images): with T = x_thr + 1 the mask is `pixel >= T`, and the
        // handle's last call gives a window [pred_lo, pred_hi] in which this call's T are expected.  One bit per pixel, pixel >=
        // pred_hi, 16 pixels = one 16-bit word; a group holding a pixel with pred_lo <= pixel < pred_hi goes on the image's list
        // (k_mask_fix decides those pixels once T is known).  1/8 byte per pixel written instead of 1/2.
        uint16_t* pbits = KIND == 2 ? (uint16_t*)(j.bins + (int64_t)img * (npix >> 3)) : nullptr;
        uint32_t* pcnt = j.edges_i + (int64_t)img * kEdgesI + kEdgesI_PredCnt;
        uint32_t* plist = KIND == 2 ? j.pred_list + (int64_t)img * j.pred_stride : nullptr;
        const uint32_t t_lo = j.pred_lo, t_hi = j.pred_hi, pcap = (uint32_t)j.pred_cap;
        // The next step's 32 bytes are requested before this step's 16 pixels are counted, and a plane's store stays in flight
        // across the loop boundary (loads, store and waits issued by hand, lfa_device.h): left to the compiler the loop asks for
        // its bytes at the top of every step and waits for them with nothing else to do.
        lfa_u32x4 n0, n1;
        if (tid < ng) { vm_request32(v + 2 * tid, n0, n1); vm_wait32<0>(n0, n1); }
        constexpr bool stores = KIND != 0;                            // one store per lane and step
        for (int64_t g = tid; g < ng; g += nth) {
            const uint32_t w[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
            const int64_t gn = g + nth < ng ? g + nth : g;                // (the last step asks for its own bytes again; the copy is not used)
            vm_request32(v + 2 * gn, n0, n1);
#pragma unroll
            for (int k = 0; k < 8; ++k) { count((int)(w[k] & 0xffffu) - xmin); count((int)(w[k] >> 16) - xmin); }
            if (KIND == 2) {
                uint32_t set = 0, und = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t x = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                    set |= (x >= t_hi ? 1u : 0u) << k;
                    und |= (x >= t_lo && x < t_hi) ? 1u : 0u;
                }
                const unsigned long long um = __ballot(und != 0u);    // (over the lanes still in the loop)
                if (um) {                                             // rare: the valley of the histogram
                    const int lane = threadIdx.x & 63, leader = __ffsll(um) - 1;
                    uint32_t slot0 = 0;
                    if (lane == leader) slot0 = atomicAdd(pcnt, (uint32_t)__popcll(um));
                    slot0 = (uint32_t)__shfl((int)slot0, leader);
                    const uint32_t slot = slot0 + (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
                    if (und != 0u && slot < pcap) plist[slot] = (uint32_t)g;
                }
                vm_store2(pbits + g, set);
            }
            if (KIND == 1) {
                uint32_t cls[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t x = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                    const uint32_t c = min((x > pa ? x - pa : 0u) >> ps, 15u);
                    cls[k >> 3] |= c << (4 * (k & 7));
                }
                vm_store8(plane + g, cls[0], cls[1]);
            }
            // the requested bytes are in; issued after them: this step's store (more when the list took something: the wait then
            // is merely earlier), which may still be on its way
            if (stores) vm_wait32<1>(n0, n1); else vm_wait32<0>(n0, n1);
        }
        done = ng << 4;
    }
    for (int64_t p = done + tid; p < npix; p += nth) count((int)base[p] - xmin);
    __syncthreads();
    for (int k = threadIdx.x; k < nlds; k += blockDim.x) {
        uint32_t c = 0;
        for (uint32_t q = 0; q <= cmask; ++q) c += sh[((uint32_t)k << cshift) | ((q + threadIdx.x) & cmask)];   // rotated: spreads the lanes over the banks
        if (c) { atomicAdd(&gh[k], c); ohi = max(ohi, k); olo = min(olo, k); }
    }
    if (j.hist_abs) {                                            // one-pass form: the occupied range falls out of the flush
        olo = wave_min_i32(olo); ohi = wave_max_i32(ohi);
        if ((threadIdx.x & 63) == 0 && ohi >= 0) { atomicMin(&j.info[img].xmin, olo); atomicMax(&j.info[img].xmax, ohi); }
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
    // 16 sub-histograms: lane l adds into copy l & 15, i.e. word (bin << 4) | (l & 15); the 16 lanes with different copies
    // sit on 16 different banks, two lanes with the same copy collide only when their bins have the same parity
    __shared__ uint32_t sh[256 * kHC];
    __shared__ double st[kEdges + 1];
    const int img = blockIdx.y;
    if (j.info[img].all_equal) return;
    if (j.grey_mode && j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Grey]) return;      // k_edges_rgb has built this image's histogram
    for (int k = threadIdx.x; k < 256 * kHC; k += 256) sh[k] = 0;
    uint32_t* shl = sh + (threadIdx.x & (kHC - 1));
    auto hadd = [&](uint32_t k) { atomicAdd(&shl[k * kHC], 1u); };
    const double* te = j.edges_g + (int64_t)img * kEdges;
    __shared__ uint32_t sg[kEdges + 3];
    const uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
    // (both edge lists are fetched before the one barrier: a block's start costs one trip to memory, not two)
    for (int k = threadIdx.x; k < kEdges; k += 256) st[k] = te[k];
    for (int k = threadIdx.x; k < kEdges + 3; k += 256) sg[k] = (k < 256) ? eg[k] : kGInf;   // sentinel above bin 255
    const bool integer_path = eg[kEdgesI_Crit] == 0;
    __syncthreads();
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int64_t npix = (int64_t)j.H * j.W;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    uint32_t amb = 0;
    // float64 path only: bin estimate in float32 (v = g^0.4 by exp2/log2, then the uniform-bin formula), fixed up by
    // compares against the edges
    const float vminf = (float)j.info[img].vmin;
    const float scalef = 256.0f / ((float)j.info[img].vmax - vminf);
    // integer path: one table entry per bucket of 1024 G values (G <= 2 550 000 < 2^22):
    //   .x      the next edge above the bucket's first G (clamped to 2^22-1 = never reached)
    //   .y      bits 0-7 the bin of the bucket's first G; bit 30: a second edge lies inside the bucket as well (only for
    //           nearly black pixels, where bins are narrower than a bucket) -> that pixel walks the edge list;
    //           predicted-threshold mode (Job::pred_mode; the window's ends are multiples of 1024): bit 31 the bucket lies at or
    //           above the window, bit 29 inside it
    // Two words, so that neither has to be taken apart: bin = .y + (G >= .x) is a compare and an add-with-carry; the flag bits
    // ride along in the sum, and the LDS index (bin * 16, 32-bit) drops them by itself.
    __shared__ uint2 tab[2496];
    __shared__ lfa_u32x4 turn[4][192];
    if (integer_path) {
        // A thread takes ten consecutive buckets: one binary search for the first, then the bin only moves forward (a block builds
        // this table before its first pixel and the launch runs eight rounds of blocks per CU: ten searches of eight dependent
        // LDS reads each per thread were 3 us per block, a twentieth of the pass).
        constexpr int kPer = (2496 + 255) / 256;
        {
            const int b0 = threadIdx.x * kPer;
            uint32_t G0 = (uint32_t)b0 << 10;
            int lo = 0, hi = 255;                           // largest k with sg[k] <= G0  (sg[0] = 0)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                const bool ge = sg[mid] <= G0;
                lo = ge ? mid : lo;
                hi = ge ? hi : mid - 1;
            }
            for (int b = b0; b < b0 + kPer && b < 2496; ++b, G0 += 1024u) {
                while (lo < 255 && sg[lo + 1] <= G0) ++lo;
                const uint32_t e1 = sg[lo + 1] < 0x3fffffu ? sg[lo + 1] : 0x3fffffu;
                const uint32_t two = (sg[lo + 2] <= G0 + 1023u) ? 0x40000000u : 0u;
                const uint32_t win = !j.pred_mode ? 0u : (G0 >= j.pred_hi ? 0x80000000u : (G0 >= j.pred_lo ? 0x20000000u : 0u));
                tab[b] = make_uint2(e1, (uint32_t)lo | two | win);
            }
        }
        __syncthreads();
    }
    auto binG = [&](uint32_t G, uint32_t& flags) -> uint32_t {   // bin unless the bucket is flagged (then a lower bound)
        const uint2 ent = tab[G >> 10];                       // one 8-byte LDS read
        flags |= ent.y;
        return ent.y + ((G >= ent.x) ? 1u : 0u);              // bits 29-31 ride along: no user looks above bit 7 without masking
    };
    auto walk = [&](uint32_t G, uint32_t k) -> uint32_t { k &= 0xffu; while (G >= sg[k + 1]) ++k; return k; };   // sg[256..] = inf ends it
    // class plane for k_mask (integer path only): bin >> 4 of every pixel, two pixels per byte (pixel 2q in the low nibble)
    uint8_t* bins = j.bins ? j.bins + (int64_t)img * (npix >> 1) : nullptr;
    auto add = [&](double g) {                             // float64 path (an edge is critical): exact compares + census
        const float vf = __builtin_amdgcn_exp2f(0.4f * __builtin_amdgcn_logf((float)g));
        int k = (int)((vf - vminf) * scalef);
        k = k < 0 ? 0 : (k > 255 ? 255 : k);
        while (k > 0 && g < st[k]) --k;                 // largest k in [0,255] with t[k] <= g  (t[0] = gmin <= g always)
        while (k < 255 && g >= st[k + 1]) ++k;
        // pixels within 2^-40 (relative) of an interior edge: their bin could depend on the last bits of pow()
        const int a = k < 1 ? 1 : k;
        const int b = k + 1 <= 255 ? k + 1 : 255;
        amb += near_rel(g, st[a]) || near_rel(g, st[b]);
        hadd(k);
    };
    int64_t done = 0;
    if (integer_path) {
        if (j.aligned16) {
            const int64_t ng = npix >> 4;
            // The next group's 48 bytes are requested before this group's ~500 instructions start, so a wave never
            // stands still for a memory latency; the class-plane store of a round stays in flight across the loop
            // boundary (lfa_device.h, "vector memory operations issued by hand").  The last round asks for its own
            // group again; the copy is not used.
            // Memory shape as in k_minmax_rgb: a wave takes a round of 64 groups (3 KiB) as three fully coalesced nontemporal
            // 16-byte loads and turns them through its own 3 KiB of LDS into 48 contiguous bytes per lane; block b owns a
            // contiguous run of rounds of its image, its four waves take them in turn.
            const int lane = threadIdx.x & 63, wv4 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the round loop is wave-uniform)
            // (round numbers are 32-bit -- at most 2^20 rounds per image -- and scalar, so the loop control is scalar code)
            const int rounds = (int)((ng + 63) >> 6);
            const int per_block = (rounds + (int)gridDim.x - 1) / (int)gridDim.x;
            const int r_end = min(rounds, (int)(blockIdx.x + 1) * per_block);
            const int64_t n16 = ng * 3;
            auto piece = [&](int r, int k) -> const uint8_t* {         // past the image's end: the round's first piece again
                const int64_t i = (int64_t)r * 192 + 64 * k + lane;
                return base + (i < n16 ? i : (int64_t)r * 192) * 16;
            };
            // kind 0: histogram only; 1: also the 4-bit class plane; 2: also the predicted-threshold bit plane and the list of
            // undecided groups (Job::pred_mode)
            uint8_t* pbits = j.bins + (int64_t)img * (npix >> 3);
            uint32_t* pcnt = j.edges_i + (int64_t)img * kEdgesI + kEdgesI_PredCnt;
            uint32_t* plist = j.pred_list + (int64_t)img * j.pred_stride;
            const uint32_t pcap = (uint32_t)j.pred_cap;
            auto pass = [&](auto kind) {
                constexpr int kKind = decltype(kind)::value;
                constexpr bool kPlane = kKind != 0;
                Raw48 nx;
                int r = (int)blockIdx.x * per_block + wv4;
                if (r >= r_end) return;
                vm_request3nt(piece(r, 0), piece(r, 1), piece(r, 2), nx);
                vm_wait48<0>(nx);
                for (; r < r_end; r += 4) {
                    turn[wv4][lane] = nx.a; turn[wv4][64 + lane] = nx.b; turn[wv4][128 + lane] = nx.c;
                    const int rn = (r + 4 < r_end) ? r + 4 : r;
                    vm_request3nt(piece(rn, 0), piece(rn, 1), piece(rn, 2), nx);
                    const int64_t g = (int64_t)r * 64 + lane;
                    uint32_t w[12];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const lfa_u32x4 v = turn[wv4][3 * lane + k];
                        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
                    }
                    uint32_t G[16], kb[16], flags = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { G[k] = gray_G16(w, k); kb[k] = binG(G[k], flags); }
                    uint32_t set = 0;
                    if (kKind == 2) {
                        // bit k of `set`: pixel k lies at or above the window (set whatever the threshold turns out to be, as long
                        // as it is inside the window): bit 31 of the table word, shifted in by one v_alignbit per pixel
#pragma unroll
                        for (int k = 15; k >= 0; --k) set = __builtin_amdgcn_alignbit(set, kb[k], 31);
                    }
                    if (flags & 0x40000000u) {                        // some pixel of the group is in a two-edge bucket
#pragma unroll
                        for (int k = 0; k < 16; ++k) kb[k] = walk(G[k], kb[k]);
                    }
                    if (kKind == 2) {
                        // a group with a pixel in a bucket inside the window goes on the image's list: k_mask_fix decides its
                        // pixels once the threshold is known
                        const bool und = (flags & 0x20000000u) != 0u && g < ng;
                        const unsigned long long um = __ballot(und);
                        if (um) {                                     // rare: the valley of the histogram
                            const int leader = __ffsll(um) - 1;
                            uint32_t slot0 = 0;
                            if (lane == leader) slot0 = atomicAdd(pcnt, (uint32_t)__popcll(um));
                            slot0 = (uint32_t)__shfl((int)slot0, leader);
                            const uint32_t slot = slot0 + (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
                            if (und && slot < pcap) plist[slot] = (uint32_t)g;
                        }
                        if (g < ng) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) hadd(kb[k]);
                            vm_store2(pbits + g * 2, set);
                        }
                    } else if (g < ng) {                              // only the image's last round has lanes past the end; every round has
                        uint32_t pk[2] = {0u, 0u};                    // lane 0 inside, so the store below is one operation of the wave in
#pragma unroll                                                        // every round -- the hand-counted vmcnt waits rely on that
                        for (int k = 0; k < 16; ++k) { hadd(kb[k]); pk[k >> 3] |= ((kb[k] >> 4) & 15u) << (4 * (k & 7)); }
                        if (kPlane) vm_store8(bins + g * 8, pk[0], pk[1]);
                    }
                    // the requested bytes are in; this round's store -- the newest operation, whatever the list took -- may
                    // still be on its way
                    vm_wait48<kPlane ? 1 : 0>(nx);
                }
            };
            if (j.pred_mode) pass(std::integral_constant<int, 2>{});
            else if (bins) pass(std::integral_constant<int, 1>{});
            else pass(std::integral_constant<int, 0>{});
            done = ng << 4;
        }
        for (int64_t p = done + tid; p < npix; p += nth) {
            uint32_t flags = 0;
            const uint32_t G = gray_G(base[3 * p], base[3 * p + 1], base[3 * p + 2]);
            const uint32_t k = walk(G, binG(G, flags));
            hadd(k);                                                  // (the plane is only used when H*W is a multiple of 16)
        }
    } else {
        if (j.aligned16) {
            const int64_t ng = npix >> 4;
            for (int64_t g = tid; g < ng; g += nth) {
                uint32_t w[12];
                load48(base + g * 48, w);
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    add(gray_u8<V>(byte_of(w, 3 * k), byte_of(w, 3 * k + 1), byte_of(w, 3 * k + 2)));
            }
            done = ng << 4;
        }
        for (int64_t p = done + tid; p < npix; p += nth) add(gray_at<V>(base, p, j.dtype, j.C));
    }
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int c = 0; c < kHC; ++c) tot += sh[threadIdx.x * kHC + ((c + threadIdx.x) & (kHC - 1))];     // rotated: spreads the lanes over the banks
    if (tot) atomicAdd(&j.hist[(int64_t)img * j.hist_stride + threadIdx.x], tot);
    amb = wave_sum_u32(amb);
    if ((threadIdx.x & 63) == 0 && amb) atomicAdd(&j.info[img].ambiguous, amb);
}

// ------------------------------------------------------------------------------------------------------------------
// Otsu, integer path.  One 256-thread block per image.  Up to 2^24 pixels all sums are exact integers (so the
// reference's float32 cumsums are exact too); the only floating-point steps are the two divisions, the float32
// product w1*w2 and the final float64 product, done per candidate in the reference's order.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_otsu_int(Job j) {
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    const uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    const int t = threadIdx.x;
    __shared__ int s_lo, s_hi, s_xthr;                          // s_xthr: x_thr of a uint16 frame on the one-pass histogram (else < 0), for the block-wide tail
    __shared__ unsigned long long s_w[4], s_s[4];
    __shared__ double s_bv[4];
    __shared__ int s_bi[4];
    int xmin, xmax;
    if (t == 0) s_xthr = -1;
    const bool abs_hist = j.dtype == LFA_U8 || j.hist_abs;      // bins are pixel values
    if (j.dtype == LFA_U8) {
        // histogram covers 0..255; find occupied range
        if (t == 0) { s_lo = 0x7fffffff; s_hi = -1; }
        __syncthreads();
        if (gh[t]) { atomicMin(&s_lo, t); atomicMax(&s_hi, t); }
        __syncthreads();
        xmin = s_lo; xmax = s_hi;
        if (t == 0) { inf->xmin = xmin; inf->xmax = xmax; }
    } else {
        xmin = inf->xmin; xmax = inf->xmax;                     // from k_minmax_u16, or from k_hist_u16 in its one-pass form
    }
    if (t == 0 && j.dtype == LFA_U16 && j.plane_stats) { atomicMax(&j.plane_stats[2], (uint32_t)max(xmax, 0)); atomicAdd(&j.plane_stats[3], 1u); }
    const int base_bin = abs_hist ? xmin : 0;                   // two-pass uint16 histogram is stored relative to xmin
    const int nb = xmax - xmin + 1;
    auto lut = [&](int x) -> int { return (j.dtype == LFA_U8) ? (int)j.lut8[x] : (int)j.lut16[x]; };
    uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
    if (nb <= 1) {
        if (t == 0) { inf->all_equal = 1; inf->v_thr = lut(xmin); inf->x_thr = 0x7fffffff; inf->otsu_bin = -1; eg[kEdgesI_Bins] = 0u; }
        return;
    }
    const int seg = (nb + 255) / 256;
    const int b0 = t * seg, b1 = min(nb, b0 + seg);
    double bv = -1.0; int bi = -1;
    if ((int64_t)j.H * j.W > (1 << 24)) {
        // more than 2^24 pixels: the reference's float32 weights round, in bin order (lfa_algo.h otsu_seq_*).  Lane 0 walks
        // the histogram upwards and lane 64 downwards, tile by tile through LDS; then every thread prices the splits of
        // its own segment from the four prefix arrays.
        constexpr int T = 1024;
        __shared__ uint32_t cf[T], cb[T];
        __shared__ int32_t lf[T + 1], lb[T + 1];
        float* w1 = j.otsu_w + (int64_t)img * 2 * j.hist_stride; float* w2 = w1 + j.hist_stride;
        double* s1 = j.otsu_s + (int64_t)img * 2 * j.hist_stride; double* s2 = s1 + j.hist_stride;
        OtsuSeq sf{0.0f, 0.0, 0}, sb{0.0f, 0.0, 0};
        const int ntiles = (nb + T - 1) / T;
        for (int it = 0; it < ntiles; ++it) {
            const int f0 = it * T, fn = min(T, nb - f0);
            const int r0 = (ntiles - 1 - it) * T, rn = min(T, nb - r0);
            for (int k = t; k < fn; k += 256) { cf[k] = gh[base_bin + f0 + k]; lf[k] = lut(xmin + f0 + k); }
            for (int k = t; k < rn; k += 256) { cb[k] = gh[base_bin + r0 + k]; lb[k + 1] = lut(xmin + r0 + k); }
            if (t == 0) { lf[fn] = (f0 + fn < nb) ? lut(xmin + f0 + fn) : -1; lb[0] = (r0 > 0) ? lut(xmin + r0 - 1) : -1; }
            __syncthreads();
            if (t == 0) otsu_seq_fwd(cf, lf, fn, sf, w1 + f0, s1 + f0);
            if (t == 64) otsu_seq_bwd(cb, lb, rn, sb, w2 + r0, s2 + r0);
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        for (int b = b0; b < b1; ++b)
            if (b < nb - 1 && lut(xmin + b + 1) != lut(xmin + b)) {
                const double var = otsu_seq_variance(w1[b], s1[b], w2[b + 1], s2[b + 1]);
                if (bi < 0 || var > bv) { bv = var; bi = b; }
            }
    } else {
    unsigned long long w = 0, s = 0;
    for (int b = b0; b < b1; ++b) {
        const unsigned long long c = gh[base_bin + b];
        w += c; s += c * (unsigned long long)lut(xmin + b);
    }
    // exclusive scan over the 256 segment sums: inclusive scan inside each wave, then the four wave totals
    unsigned long long iw = w, is = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long uw = __shfl_up(iw, o), us = __shfl_up(is, o);
        if ((t & 63) >= o) { iw += uw; is += us; }
    }
    if ((t & 63) == 63) { s_w[t >> 6] = iw; s_s[t >> 6] = is; }
    __syncthreads();
    unsigned long long totw = 0, tots = 0, cw = iw - w, cs = is - s;
    for (int k = 0; k < 4; ++k) {
        const unsigned long long aw = s_w[k], as = s_s[k];
        totw += aw; tots += as;
        if (k < (t >> 6)) { cw += aw; cs += as; }
    }
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
    }
    // first maximum over all candidates: bins ascend with the thread index, so ties go to the lower bin
    if (bi < 0) { bv = -1.0; bi = 0x7fffffff; }                 // (variances are >= 0)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(bv, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
    }
    if ((t & 63) == 0) { s_bv[t >> 6] = bv; s_bi[t >> 6] = bi; }
    __syncthreads();
    if (t == 0) {
        double best = s_bv[0]; int idx = s_bi[0];
        for (int k = 1; k < 4; ++k)
            if (s_bv[k] > best || (s_bv[k] == best && s_bi[k] < idx)) { best = s_bv[k]; idx = s_bi[k]; }
        if (idx == 0x7fffffff) idx = -1;
        eg[kEdgesI_Bins] = 0u;
        if (idx < 0) {                               // gamma table flat over the whole range: all gamma values equal
            inf->all_equal = 1; inf->v_thr = lut(xmin); inf->x_thr = 0x7fffffff; inf->otsu_bin = -1;
        } else {
            const int v = lut(xmin + idx);
            const int x_thr = xmin + idx;
            inf->v_thr = v; inf->x_thr = x_thr; inf->otsu_bin = v - lut(xmin);
            if (j.dtype == LFA_U16 && j.plane_stats) {
                atomicAdd(&j.plane_stats[5], (uint32_t)x_thr); atomicAdd(&j.plane_stats[8], 1u);
                atomicMax(&j.plane_stats[9], (uint32_t)(65535 - x_thr)); atomicMax(&j.plane_stats[10], (uint32_t)x_thr);
            }
            if (j.dtype == LFA_U16 && j.hist_abs && j.u16_pred) {
                // Class plane (k_hist_u16): the classes above the threshold's class q are set, those below are clear, and only
                // the pixels of class q itself are compared in k_mask on the image.  If x_thr is the last value of its
                // class, that class is clear as a whole and nothing is compared (classes one value wide: always).
                // Otherwise worth it when those pixels are few: q is one of the narrow classes and holds at most 1/2048
                // of the pixels (debug flag 4 lifts both conditions).
                const int q = min(max(x_thr - j.u16_pa, 0) >> j.u16_ps, 15);
                const bool settled = q <= 14 && x_thr == j.u16_pa + ((q + 1) << j.u16_ps) - 1;
                bool could = settled;
                if (!settled && q >= 1 && q <= 14) {
                    const int c0 = j.u16_pa + (q << j.u16_ps), c1 = min(c0 + (1 << j.u16_ps), 65536);
                    unsigned long long in_class = 0;
                    for (int x = c0; x < c1; ++x) in_class += gh[x];
                    could = in_class * 2048ull <= (unsigned long long)((int64_t)j.H * j.W);
                }
                if (j.plane_stats) { if (could) atomicAdd(&j.plane_stats[6], 1u); atomicAdd(&j.plane_stats[7], 1u); }
                eg[kEdgesI_Class] = (uint32_t)q | (settled ? 0x100u : 0u);
                eg[kEdgesI_Bins] = (j.u16_plane && (could || (j.debug_flags & 4))) ? 1u : 0u;
            }
            if (j.dtype == LFA_U16 && j.hist_abs) {
                // Predicted-threshold mode.  Every one-pass call reports the range of its T = x_thr + 1 (the next call's window,
                // lfa_api.hip); below, whether the window the handle has in mind would have held this frame.
                const uint32_t T = (uint32_t)x_thr + 1u;
                s_xthr = x_thr;
                if (j.pred_mode) {
                    // the bit plane decides this frame iff T lies inside the window it was written for and the list holds every undecided group
                    const bool held = T >= j.pred_lo && T <= j.pred_hi && eg[kEdgesI_PredCnt] <= (uint32_t)j.pred_cap;
                    eg[kEdgesI_Bins] = held ? 2u : 0u;
                    if (j.plane_stats) { atomicAdd(&j.plane_stats[14], 1u); if (held) atomicAdd(&j.plane_stats[15], 1u); }
                }
            }
        }
    }
    // Would the window the handle has in mind (Job::pred_probe_lo / _hi, from its last completed call) have held this frame: T inside
    // it and at most 1/2048 of the pixels inside it -- a window reaching into the background noise fills the list of undecided
    // groups.  The handle takes the predicted-threshold mode only while at least half of its last call's frames say yes, so a
    // series whose thresholds lie inside the noise never pays for a failed try.  Counted from the histogram by the whole block.
    // Only frames with few pixels near their threshold -- at most 1/1024 within 8 values of it -- shape the next window: one whose
    // threshold fell into the background noise could not use the bits anyway.  The handle gets every such frame's T (lfa_api.hip
    // looks for the tightest cluster: one frame of another scene in the batch must not stretch the window over everything between).
    if (j.dtype == LFA_U16 && j.hist_abs && j.plane_stats) {
        __shared__ unsigned long long s_in, s_near;
        __syncthreads();                                             // (s_xthr)
        const int x_thr = s_xthr;
        if (t == 0) { s_in = 0ull; s_near = 0ull; }
        __syncthreads();
        if (x_thr >= 0) {
            if (t < 16) { const int x = x_thr - 7 + t; if (x >= 0 && x <= 65535) { const uint32_t c = gh[x]; if (c) atomicAdd(&s_near, (unsigned long long)c); } }
            unsigned long long c = 0;
            for (uint32_t x = j.pred_probe_lo + (uint32_t)t; x < j.pred_probe_hi && x < 65536u; x += 256u) c += gh[x];
            if (c) atomicAdd(&s_in, c);
        }
        __syncthreads();
        if (t == 0 && x_thr >= 0) {
            const uint32_t T = (uint32_t)x_thr + 1u;
            const unsigned long long npx = (unsigned long long)((int64_t)j.H * j.W);
            const bool sparse = s_near * 1024ull <= npx;
            if (sparse) { atomicMax(&j.plane_stats[12], 0x7fffffffu - T); atomicMax(&j.plane_stats[13], T); }
            if (j.thr_list) j.thr_list[img] = sparse ? T : 0u;
            if (j.pred_probe_hi > j.pred_probe_lo) {
                const bool would = T >= j.pred_probe_lo && T <= j.pred_probe_hi && s_in * 2048ull <= npx;
                if (would) atomicAdd(&j.plane_stats[16], 1u);
                atomicAdd(&j.plane_stats[17], 1u);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Otsu, colour path: 256 float bins, sequential like the reference (one lane), then the gray-space threshold.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_otsu_rgb(Job j) {
    // Same arithmetic as lfa_algo.h otsu_float256 (checked on the host against skimage's formula): the four cumulative
    // sums stay sequential (their rounding depends on the order) but run on four different waves at once; centres,
    // products, divisions and variances are independent per bin and run one per thread.
    __shared__ float c32[256], w1[256], w2[256];
    __shared__ double ctr[256], prod[256], s1[256], s2[256], var[256];
    const int img = blockIdx.x;
    LfaImageInfo* inf = &j.info[img];
    const int t = threadIdx.x;
    const bool is_grey = j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Grey] != 0u;
    if (t == 0 && is_grey && j.plane_stats) atomicAdd(&j.plane_stats[11], 1u);
    if (inf->all_equal) {
        if (t == 0) {
            inf->v_thr_f = inf->vmin; inf->g_thr = __longlong_as_double(0x7ff0000000000000LL);
            uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
            eg[kEdgesI_MaskThr] = kGInf; eg[kEdgesI_MaskCrit] = 0u; eg[kEdgesI_Bins] = 0u;
        }
        return;
    }
    const Linspace edges(inf->vmin, inf->vmax, kEdges);
    {
        const float c = (float)j.hist[(int64_t)img * j.hist_stride + t];
        const double ce = (edges(t) + edges(t + 1)) / 2.0;
        c32[t] = c; ctr[t] = ce; prod[t] = (double)c * ce;
    }
    __syncthreads();
    if (t == 0)        { float a = 0.0f; for (int i = 0; i < 256; ++i) { a = a + c32[i]; w1[i] = a; } }
    else if (t == 64)  { float a = 0.0f; for (int i = 255; i >= 0; --i) { a = a + c32[i]; w2[i] = a; } }
    else if (t == 128) { double a = 0.0; for (int i = 0; i < 256; ++i) { a = a + prod[i]; s1[i] = a; } }
    else if (t == 192) { double a = 0.0; for (int i = 255; i >= 0; --i) { a = a + prod[i]; s2[i] = a; } }
    __syncthreads();
    if (t < 255) {
        const double m1 = s1[t] / (double)w1[t];
        const double m2 = s2[t + 1] / (double)w2[t + 1];
        const float ww = w1[t] * w2[t + 1];
        const double d = m1 - m2;
        var[t] = (double)ww * (d * d);
    }
    __syncthreads();
    // np.argmax(variance12): the first NaN if there is one, else the first index of the maximum -- as a reduction (the
    // scan by one lane, 255 dependent LDS reads, was 18 us of this kernel's 45)
    __shared__ double s_av[4];
    __shared__ int s_ai[4], s_an[4];
    {
        double v = (t < 255) ? var[t] : -1.0;                      // variances are >= 0 or NaN
        const bool isn = lfa_isnan(v);
        const unsigned long long nb = __ballot(isn);
        if (isn) v = -1.0;
        int idx = t;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(v, o);
            const int i2 = __shfl_xor(idx, o);
            if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
        }
        if ((t & 63) == 0) {
            s_av[t >> 6] = v; s_ai[t >> 6] = idx;
            s_an[t >> 6] = nb ? ((t & ~63) + __ffsll((unsigned long long)nb) - 1) : 0x7fffffff;
        }
    }
    __syncthreads();
    if (t != 0) return;
    int best = s_ai[0], first_nan = s_an[0];
    {
        double bv = s_av[0];
        for (int k = 1; k < 4; ++k) {
            if (s_av[k] > bv) { bv = s_av[k]; best = s_ai[k]; }    // strict: the earlier wave keeps a tie
            first_nan = min(first_nan, s_an[k]);
        }
        if (first_nan != 0x7fffffff) best = first_nan;
    }
    const double centre = ctr[best];
    inf->otsu_bin = best; inf->v_thr_f = centre;
    // mask = pow(gray, 0.4) > centre  <=>  gray >= T  with T the first double whose power exceeds the centre
    const double T = j.host_pow ? j.host_pow[(int64_t)img * kHostPowDoubles + 2 + kEdges + best]
                                : first_true(inf->gmin, inf->gmax, pow(centre, 2.5), [centre](double g) { return gamma_pow(g) > centre; });
    const long long tb = __double_as_longlong(T);
    inf->g_thr = (tb > 0 && tb < 0x7ff0000000000000LL) ? __longlong_as_double(tb - 1) : T;
    // integer form of `gray >= T` for k_mask (critical -> k_mask takes the float64 path)
    uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
    bool crit = false;
    const uint32_t G_thr = g_threshold(T, &crit);
    eg[kEdgesI_MaskThr] = G_thr;
    eg[kEdgesI_MaskCrit] = (crit || (j.debug_flags & 1) || j.dtype != LFA_U8) ? 1u : 0u;
    // k_mask may decide from the class plane alone (class > class of best: set, <: clear, ==: exact compare) iff the
    // histogram took the integer path (the plane was written), the threshold has an exact integer form and sits inside
    // its own bin in integer terms: E[best] <= G_thr <= E[best+1]  (E[256] = +inf)
    const uint32_t e_lo = eg[best], e_hi = (best + 1 <= 255) ? eg[best + 1] : kGInf;
    // ... and iff few pixels share the threshold's class.  Every 16-pixel group holding one is read again from the image
    // by a divergent loop in k_mask; measured, that costs more than the plane saves once ~1.5 % of the groups are
    // touched (a photo whose threshold sits on a populated grey level: 1-99 % of the pixels on the reference's sample
    // images; k_mask 8x slower than comparing the image).  The histogram knows the count: plane only below 1/2048 of
    // the pixels (debug flag 4 lifts the limit, for the tests of the fix-up loop).
    uint32_t in_class = 0;
    for (int i = (best & ~15); i < (best & ~15) + 16; ++i) in_class += j.hist[(int64_t)img * j.hist_stride + i];
    const bool few = (j.debug_flags & 4) || (uint64_t)in_class * 2048u <= (uint64_t)j.H * (uint64_t)j.W;
    const bool could = few && eg[kEdgesI_Crit] == 0 && !crit && !(j.debug_flags & 1) && e_lo <= G_thr && G_thr <= e_hi;
    eg[kEdgesI_Bins] = (j.bins && could && !(j.grey_mode && is_grey)) ? 1u : 0u;      // (grey mode: no histogram pass, no plane)
    if (j.plane_stats) { if (could) atomicAdd(&j.plane_stats[0], 1u); atomicAdd(&j.plane_stats[1], 1u); }
    // Predicted-threshold mode: the bit plane decides this image iff its threshold has an exact integer form inside the window the
    // plane was written for and the list of undecided groups holds them all.  Every call reports the range of its integer
    // thresholds: the next call's window (lfa_api.hip).
    const bool exact = eg[kEdgesI_Crit] == 0 && !crit && !(j.debug_flags & 1) && j.dtype == LFA_U8 && G_thr < 0x400000u;
    if (exact && j.plane_stats) { atomicMax(&j.plane_stats[12], 0x7fffffffu - G_thr); atomicMax(&j.plane_stats[13], G_thr); }
    if (j.pred_mode) {
        // (an equal-channel image of a grey-mode call: k_grey_hist wrote its bits, in grey levels against the unwidened window)
        const uint32_t b_thr = (G_thr + 9999u) / 10000u;
        const bool inside = (j.grey_mode && is_grey) ? (b_thr >= (j.pred_lo_raw + 9999u) / 10000u && b_thr <= (j.pred_hi_raw + 9999u) / 10000u)
                                                     : (G_thr >= j.pred_lo && G_thr <= j.pred_hi);
        const bool held = exact && inside && eg[kEdgesI_PredCnt] <= (uint32_t)j.pred_cap;
        eg[kEdgesI_Bins] = held ? 2u : 0u;
        if (j.plane_stats) { atomicAdd(&j.plane_stats[14], 1u); if (held) atomicAdd(&j.plane_stats[15], 1u); }
    }
}

// ------------------------------------------------------------------------------------------------------------------
static int blocks_per_image(const Job& j, int px_per_block, int launch_blocks = 4096) {
    const int64_t npix = (int64_t)j.H * j.W;
    int64_t need = (npix + px_per_block - 1) / px_per_block;
    int64_t want = (launch_blocks + j.n - 1) / j.n;          // keep the whole launch around 4096 blocks
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
    LFA_LAUNCH("k_init_info", k_init_info, dim3(j.H >= 4096 ? 8 : 2, j.n), dim3(256), 0, s, j);
    if (j.C == 1 && j.dtype == LFA_U16 && j.hist_abs) {
        // one-pass uint16 histogram: bins = pixel values, every bin may be touched.  Cleared BEFORE the wait below: it depends on
        // nothing of the other lane (round 3: behind the wait its 27 us sat between two lanes' histogram passes, in every chunk)
        const hipError_t e = hipMemsetAsync(j.hist, 0, (size_t)j.n * kHistBinsU16 * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    if (j.wait_event) {                                   // lfa_chain_passes: the passes below start when the other handle's are done
        const hipError_t e = hipStreamWaitEvent(s, j.wait_event, 0);
        if (e != hipSuccess) return e;
    }
    if (j.C == 3 || j.dtype == LFA_F64) {                    // the float64-gray path (colour images, float64 images)
        static const int mm_bpi = [] { const char* e = getenv("LFA_MM_BPI"); return e ? atoi(e) : 0; }();
        const dim3 grid(mm_bpi > 0 ? mm_bpi : blocks_per_image(j, 256 * 64), j.n);
        if (j.grey_mode) LFA_LAUNCH("k_grey_hist", k_grey_hist, grid, dim3(256), 0, s, j);
        // 6144 blocks per launch for the min/max pass and (below) for the histogram pass, not 4096 and 3840 (two rounds of what the chip
        // holds of the one, three of the other): measured on the headline step over both grids (tools/r3_bpi_matrix.sh), 3.98 -> 3.84 ms.  The kernel trace says why it is the STEP
        // that moves: the tail kernels of the chunks before get onto a CU only where a pass's block retires without a successor
        // waiting, so they run in the seams between the rounds -- more, shorter rounds let them through earlier and in smaller pieces
        // (one round each, the other extreme: 4.05-4.23 ms).
        const dim3 gridmm(mm_bpi > 0 ? mm_bpi : blocks_per_image(j, 256 * 64, 6144), j.n);
        LFA_DISPATCH_V(k_minmax_rgb, gridmm, dim3(256), s, j);
        LFA_LAUNCH("k_edges_rgb", k_edges_rgb, dim3(j.n), dim3(256), 0, s, j);
        {
            // the histogram blocks live for the whole launch (grid-stride) and the chip holds 1280 of them at once; the grid
            // was three whole rounds of that (15 per image at 256 images) until the sweep above: 6144 blocks, as for the min/max pass
            int per_image = (6144 + j.n / 2) / j.n;
            const int need = (int)(((int64_t)j.H * j.W + 256 * 64 - 1) / (256 * 64));
            if (per_image > need) per_image = need;
            if (per_image < 1) per_image = 1;
            static const int hist_bpi = [] { const char* e = getenv("LFA_HIST_BPI"); return e ? atoi(e) : 0; }();
            if (hist_bpi > 0) per_image = hist_bpi;
            const dim3 gridh(per_image, j.n);
            LFA_DISPATCH_V(k_hist_rgb, gridh, dim3(256), s, j);
        }
        if (j.passes_event) (void)hipEventRecord(j.passes_event, s);
        LFA_LAUNCH("k_otsu_rgb", k_otsu_rgb, dim3(j.n), dim3(256), 0, s, j);
    } else if (j.dtype == LFA_U8) {
        const dim3 grid(blocks_per_image(j, 256 * 128), j.n);
        LFA_LAUNCH("k_hist_u8", k_hist_u8, grid, dim3(256), 0, s, j);
        if (j.passes_event) (void)hipEventRecord(j.passes_event, s);
        LFA_LAUNCH("k_otsu_int", k_otsu_int, dim3(j.n), dim3(256), 0, s, j);
    } else {
        if (!j.hist_abs) {                                // (one pass: bins = pixel values, cleared above; the range falls out of the histogram, k_otsu_int)
            const dim3 grid(blocks_per_image(j, 256 * 64), j.n);
            LFA_LAUNCH("k_minmax_u16", k_minmax_u16, grid, dim3(256), 0, s, j);
            LFA_LAUNCH("k_zero_hist_u16", k_zero_hist_u16, dim3(16, j.n), dim3(256), 0, s, j);
        }
        // (tuning knobs of tools/r3_u16_block_sizes.sh: threads per block -- 256 / 512 / 768 measured within 2 % of 1024 or worse -- and pixels per thread)
        static const int u16_threads = [] { const char* e = getenv("LFA_U16_THREADS"); const int v = e ? atoi(e) : 1024; return (v >= 256 && v <= 1024 && v % 64 == 0) ? v : 1024; }();
        static const int u16_ppt = [] { const char* e = getenv("LFA_U16_PPT"); const int v = e ? atoi(e) : 64; return v >= 16 ? v : 64; }();
        const dim3 gridh(blocks_per_image(j, u16_threads * u16_ppt), j.n);
        if (j.pred_mode && j.bins) LFA_LAUNCH("k_hist_u16", k_hist_u16<2>, gridh, dim3(u16_threads), 0, s, j);
        else if (j.u16_plane && j.bins) LFA_LAUNCH("k_hist_u16", k_hist_u16<1>, gridh, dim3(u16_threads), 0, s, j);
        else LFA_LAUNCH("k_hist_u16", k_hist_u16<0>, gridh, dim3(u16_threads), 0, s, j);
        if (j.passes_event) (void)hipEventRecord(j.passes_event, s);
        LFA_LAUNCH("k_otsu_int", k_otsu_int, dim3(j.n), dim3(256), 0, s, j);
    }
    return hipGetLastError();
}

}  // namespace lfa
