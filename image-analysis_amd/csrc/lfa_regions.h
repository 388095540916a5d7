// lfa_regions.h -- device helpers shared by k_label.hip (which ends with the candidate list) and k_regions.hip
#ifndef LFA_REGIONS_H_
#define LFA_REGIONS_H_

#include "lfa_internal.h"

namespace lfa {

constexpr int kWinRowsLds = 2048;

struct Box { int minr, minc, maxr, maxc;      // expanded, NOT clipped (detection.py:73-74)
             int cr, cc; };                   // clipped ends (numpy slicing)

__device__ inline Box cand_box(const Job& j, int64_t rbase, int root) {
    const int* bb = j.bbox + (rbase + root) * 4;      // folded with L2 atomics, possibly by this very kernel: read past L1
    Box b;
    b.minr = __hip_atomic_load(&bb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.minc = __hip_atomic_load(&bb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.maxr = max(__hip_atomic_load(&bb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b.minr + j.min_maxr);
    b.maxc = max(__hip_atomic_load(&bb[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b.minc + j.min_maxc);
    b.cr = min(b.maxr, j.H); b.cc = min(b.maxc, j.W);
    return b;
}

// Candidate list of one image, run by a whole block of NT threads (NT a multiple of 64): the labels with area >= 100
// (detection.py:81), ascending, compacted in order; then the pool offsets for the window row sums.  K = number of
// labels, label00 = label id at pixel (0,0) != 0; area / bbox / root_of_label must be complete and visible to the block.
template <int NT>
__device__ inline void candidates_phase(const Job& j, int img, int K, int label00, int* wcnt /* shared, NT/64 */, int* base /* shared */) {
    const int64_t rbase = (int64_t)img * j.run_cap;
    const int64_t cbase = (int64_t)img * j.cand_cap;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    if (t == 0) *base = 0;
    __syncthreads();
    for (int l0 = 0; l0 < K; l0 += NT) {
        const int l = l0 + t;
        bool flag = false;
        if (l < K) flag = __hip_atomic_load(&j.area[rbase + __hip_atomic_load(&j.root_of_label[rbase + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)],
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= kAreaMin;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int offs = *base;
        for (int k = 0; k < wv; ++k) offs += wcnt[k];
        if (flag) {
            const int64_t slot = cbase + offs + __popcll(bal & ((1ull << lane) - 1ull));
            j.cand[slot] = l + 1;
            j.cand_sum[slot * 3 + 0] = 0; j.cand_sum[slot * 3 + 1] = 0; j.cand_sum[slot * 3 + 2] = 0;
            j.cand_rs_off[slot] = -1;
        }
        __syncthreads();
        if (t == 0) { int tot = 0; for (int k = 0; k < NT / 64; ++k) tot += wcnt[k]; *base += tot; }
        __syncthreads();
    }
    if (t != 0) return;
    const int nc = *base;
    j.info[img].n_candidates = nc;
    // window row-sum pool: first come first served in label order; needs the dword path of k_box_sums
    const int esz = (j.dtype == LFA_U16) ? 2 : 1;
    const bool fast = j.aligned16 && ((((int64_t)j.W * j.C * esz) & 3) == 0);
    if (!fast || label00 == 0) return;                        // no box pass for this image (see k_box_sums): no pooled rows
    for (int ci = 0, used = 0; ci < nc && ci < 4096; ++ci) {
        const int lab = __hip_atomic_load(&j.cand[cbase + ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Box b = cand_box(j, rbase, __hip_atomic_load(&j.root_of_label[rbase + lab - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int h = b.cr - b.minr, w = b.cc - b.minc;
        const int len = (int)((double)h * .6) - (int)((double)h * .25);
        const int ne = ((int)((double)w * .7) - (int)((double)w * .3)) * j.C;
        if (len <= 0 || len > kWinRowsLds || ne <= 0 || (esz == 2 && ne > 65536)) continue;
        if (used + len > j.rowsum_cap) break;
        j.cand_rs_off[cbase + ci] = used;
        used += len;
    }
}

}  // namespace lfa
#endif
