// k_label.hip -- stages S4b-S6: vertical opening, connected components, label numbering, bbox / area  (gfx950)
//
// Reference (paths under /root/reference):
//   opening   LFA-Image_Analysis/detection.py:60-63 -> skimage/morphology/binary.py:79-88, :138-145 ->
//             scipy.ndimage.binary_erosion(border_value=True) / binary_dilation with a (line_length x 1) footprint
//   label     detection.py:66 -> skimage/measure/_label.py:116-118 -> scipy.ndimage.label, 8-connectivity,
//             ids 1..K in raster order of each component's first pixel
//   bbox/area detection.py:71-73 -> skimage/measure/_regionprops.py:399-411
//
// The opening keeps exactly the vertical runs of the thresholded image with
//     len + (run touches row 0 ? L/2 : 0) + (run touches row H-1 ? (L-1)/2 : 0) >= L          (L = line_length)
// (erosion window [r-L/2, r+(L-1)/2] with True outside, dilation window mirrored with False outside; checked against
// scipy in tests/test_oracle_golden.py).  So the opened image IS a list of vertical runs, and components are labelled
// on runs instead of pixels: two runs in adjacent columns touch (8-connectivity) iff their row ranges, widened by one,
// overlap.  Union-find links every run to the run of its component that holds the raster-first pixel
// (smallest row*W + col), which is also what orders the label ids.  The label image is never materialised on the
// hot path (lfa_debug_fetch paints it for tests).
#include "lfa_internal.h"
#include "lfa_device.h"
#include "lfa_regions.h"

namespace lfa {

__device__ inline bool run_survives(int s, int e, int H, int L) {
    const int len = e - s + 1;
    const int eff = len + (s == 0 ? L / 2 : 0) + (e == H - 1 ? (L - 1) / 2 : 0);
    return eff >= L;
}

// 64 rows of one column: clear the runs that lie strictly inside the word (touching neither its first nor its last
// row, nor -- through `above`, the rows past the image as ones -- the image's last row) and are shorter than L.  Such a
// run fails run_survives whatever the neighbouring words hold, every other run is left whole, and the word's first and
// last bit keep their values, so walking the result gives the same surviving runs.  A vertical opening inside the word,
// bit-parallel: erosion with ones past bit 63, dilation with zeros below bit 0, windows doubled up to L.
__device__ inline uint64_t drop_short_runs(uint64_t w, uint64_t above, int L) {
    const uint64_t x = w | above;
    const uint64_t nx = ~x;
    if (nx == 0) return w;
    uint64_t keep = x & ~(x + 1ull);                                     // the run from bit 0 upwards
    const int lead = __clzll((long long)nx);                             // length of the run ending at bit 63
    if (lead) keep |= ~0ull << (64 - lead);
    if (L < 64) {
        uint64_t e = x;
        for (int have = 1; have < L;) { const int sh = min(have, L - have); e &= (e >> sh) | ~(~0ull >> sh); have += sh; }
        for (int have = 1; have < L;) { const int sh = min(have, L - have); e |= e << sh; have += sh; }
        keep |= e;
    }
    return w & keep;
}

// 64 x 64 bit matrix across the wave: lane i comes in with row i (bit k = column k) and leaves with column i (bit k = row k).  Six
// block swaps between lane pairs (i, i ^ s), s = 32 .. 1: the upper lane of a pair takes its odd s-blocks from the partner's even ones
// and the lower lane the other way round.  A lane needs only half of its partner's bits, so what travels is ONE 32-bit word per
// swap (s = 32: a whole half; below: the wanted blocks of both halves packed side by side), and everything is 32-bit arithmetic.
// The words travel by v_permlane32_swap (s = 32), ds_swizzle (16, 8, 4) and DPP quad permutes (2, 1).  ALL 64 lanes must be here.
template <int S>
__device__ __forceinline__ void transpose_swap32(uint32_t& lo, uint32_t& hi, int lane) {
    constexpr uint32_t m = S == 16 ? 0x0000ffffu : S == 8 ? 0x00ff00ffu : S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
    // The upper lane of a pair (bit S of the lane number set) keeps its odd S-blocks and sends its even ones, the lower lane the other
    // way round.  No branch and no select: with b = S in the upper lane, 0 in the lower, and a = S - b, the word sent is lo's wanted
    // blocks moved down beside hi's moved up, and what arrives is shifted back into the blocks not kept.  (These depend on the lane
    // only: the compiler keeps them in registers across the row blocks.)
    const uint32_t b = (uint32_t)lane & (uint32_t)S, a = (uint32_t)S - b;
    const uint32_t keep = b ? ~m : m;
    const uint32_t send = (m & (lo >> a)) | (~m & (hi << b));
    uint32_t x;
    if (S >= 4) x = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send, (S << 10) | 0x1f);                       // lane ^ S inside 32 lanes, no address register
    else        x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, S == 2 ? 0x4e : 0xb1, 0xf, 0xf, false);   // quad_perm [2,3,0,1] / [1,0,3,2]
    lo = (keep & lo) | (~keep & (x << a));
    hi = (keep & hi) | (~keep & (x >> b));
}
__device__ __forceinline__ uint64_t transpose_bits64(uint64_t w, int lane) {
    // s = 32: the upper half-wave's low words change places with the lower half-wave's high words -- one instruction on gfx950
    const auto sw = __builtin_amdgcn_permlane32_swap((uint32_t)w, (uint32_t)(w >> 32), false, false);
    uint32_t lo = sw[0], hi = sw[1];
    // (one instantiation per swap: left as a loop over s the compiler kept it a loop -- around a switch for the mask -- in the eight-fold
    //  unrolled body of k_runs_count)
    transpose_swap32<16>(lo, hi, lane);
    transpose_swap32<8>(lo, hi, lane);
    transpose_swap32<4>(lo, hi, lane);
    transpose_swap32<2>(lo, hi, lane);
    transpose_swap32<1>(lo, hi, lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Walk one column top to bottom, one 64-row word at a time (step(w, rb), rb ascending, then finish()), and call
// emit(start, end_inclusive) for every surviving run.
template <class Emit>
struct ColumnWalk {
    const Emit& emit;
    const int H, L;
    int start = -1;
    __device__ ColumnWalk(const Emit& e, int H_, int L_) : emit(e), H(H_), L(L_) {}
    __device__ __forceinline__ void step(uint64_t w, int rb) {
        uint64_t t = w ^ ((w << 1) | (start >= 0 ? 1ull : 0ull));       // bit k set: mask changes between row k-1 and k
        if (__popcll((unsigned long long)t) > 8) {                       // a noisy word: most of its runs are too short to survive
            const int rem = H - rb * 64;
            w = drop_short_runs(w, rem < 64 ? ~0ull << rem : 0ull, L);
            t = w ^ ((w << 1) | (start >= 0 ? 1ull : 0ull));
        }
        while (t) {
            const int b = __ffsll((unsigned long long)t) - 1;
            t &= t - 1;
            const int row = rb * 64 + b;
            if (start < 0) start = row;
            else {
                if (run_survives(start, row - 1, H, L)) emit(start, row - 1);
                start = -1;
            }
        }
    }
    __device__ __forceinline__ void finish() { if (start >= 0 && run_survives(start, H - 1, H, L)) emit(start, H - 1); }
};
template <class Word, class Emit>   // word(rb) = the column's 64-row word of row block rb
__device__ inline void column_runs(const Word& word, int HB, int H, int L, const Emit& emit) {
    ColumnWalk<Emit> walk(emit, H, L);
    for (int rb = 0; rb < HB; ++rb) walk.step(word(rb), rb);
    walk.finish();
}

// one thread per column: count the surviving runs and park them in the column's own slots of run_tmp (run_cap is
// runs_per_col * W, and runs_per_col bounds what a column can hold), so that k_runs_fill compacts the run list
// without reading the mask a second time
// Blocks are dealt to the eight XCDs in turn (block b runs on XCD b % 8) and every XCD has an L2 of its own, so the numbering is
// folded: the blocks of one image -- which share every row's cache lines of the bit plane, 32 bytes of each per block -- are
// neighbours on ONE XCD instead of one on each (measured: 4x the plane's bytes from HBM before).
__global__ __launch_bounds__(256) void k_runs_count(Job j, int gx) {
    const int total = gridDim.x;
    const int b = blockIdx.x;
    const int v = (total & 7) ? b : (b & 7) * (total >> 3) + (b >> 3);
    const int img = v / gx;
    const int c = (v - img * gx) * blockDim.x + threadIdx.x;
    if (c >= j.W) return;
    int n = 0;
    uint32_t* tmp = j.run_tmp + (int64_t)img * j.run_cap + (int64_t)c * j.runs_per_col;
    const int per = j.runs_per_col;
    auto emit = [&](int s, int e) {
        if (n < per) tmp[n] = (uint32_t)s | ((uint32_t)e << 16);
        ++n;
    };
    if (j.pred_fused && j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Bins] == 2u) {
        // Predicted-threshold mode, widths that are multiples of 64: the mask is still the row-major bit plane of the histogram pass
        // (k_mask left this image alone, k_mask_fix has set the listed pixels in it).  The wave owns 64 columns; per row block, lane i
        // fetches the 64 bits of row i and the wave turns the 64 x 64 block, so that every lane holds its column's word -- what
        // k_mask would have written and this kernel read back (1/8 byte per pixel each way).  The next block's rows travel meanwhile.
        const int lane = threadIdx.x & 63, c0 = c & ~63, H = j.H;
        const uint64_t* plane = (const uint64_t*)(j.bins + (int64_t)img * (((int64_t)H * j.W) >> 3));
        const int64_t wpr = j.W >> 6;                               // 64-bit words per row
        auto rows = [&](int rb) -> uint64_t { const int r = rb * 64 + lane; return (rb < j.HB && r < H) ? plane[(int64_t)r * wpr + (c0 >> 6)] : 0ull; };
        // Row blocks travel kAhead at a time, one batch ahead of the one being turned: a wave's loads depend on nothing but the
        // block number, and one at a time their latencies -- 24 of them in a 1536-row image, with the image passes of the next
        // chunk keeping HBM busy -- were the kernel's whole duration.
        constexpr int kAhead = 8;
        uint64_t cur[kAhead], nxt[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) nxt[k] = rows(k);
        ColumnWalk<decltype(emit)> walk(emit, H, j.line_length);
        for (int base = 0; base < j.HB; base += kAhead) {
#pragma unroll
            for (int k = 0; k < kAhead; ++k) { cur[k] = nxt[k]; nxt[k] = rows(base + kAhead + k); }
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                if (base + k < j.HB) {                                                          // uniform across the wave
                    // a 64 x 64 block that is all background or all foreground (most of a strip image) is its own transpose
                    // (and nothing happens in it when no column has a run open / every column has)
                    const bool zeros = __ballot(cur[k] != 0ull) == 0ull, ones = __ballot(cur[k] != ~0ull) == 0ull;
                    if (!(zeros || ones)) walk.step(transpose_bits64(cur[k], lane), base + k);
                    else if (__ballot(zeros ? walk.start >= 0 : walk.start < 0) != 0ull) walk.step(cur[k], base + k);
                }
        }
        walk.finish();
    } else {
        const uint64_t* colw = j.maskT + (int64_t)img * j.HB * j.W + c;
        const int W = j.W;
        column_runs([&](int rb) -> uint64_t { return colw[(int64_t)rb * W]; }, j.HB, j.H, j.line_length, emit);
    }
    if (n > per) { n = per; atomicOr(&j.info[img].status, 2); }          // cannot happen: runs_per_col is an upper bound
    j.col_cnt[(int64_t)img * (j.W + 1) + c] = (uint32_t)n;
}

// ---- union-find on runs ---------------------------------------------------------------------------------------------
// Links always point to the run with the smaller key (row_start * W + col), so chains strictly descend, there are no
// cycles, and the root of a finished component is the run holding its raster-first pixel.  parent[] is read with
// agent-scope relaxed atomic loads while other workgroups link (a stale "I am a root" only makes the CAS fail and hand
// back the current parent); the CAS is the only writer.

__device__ inline int uf_find_live(int* parent, int i) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == i) return i;
        i = p;
    }
}

template <class KeyFn>   // key(i) = row_start * W + col of run i
__device__ inline void uf_unite(int* parent, int a, int b, const KeyFn& key) {
    for (;;) {
        a = uf_find_live(parent, a);
        b = uf_find_live(parent, b);
        if (a == b) return;
        if (key(a) < key(b)) { const int t = a; a = b; b = t; }   // a: larger key
        const int old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
        a = old;
    }
}

constexpr int kLdsRows = 4096;       // images of up to this many rows count and rank their roots per row in LDS (2 x 16 KiB)
constexpr int kLdsRuns = 8192;       // parents, row ranges and columns of up to this many runs live in LDS (32 + 32 + 16 KiB)

// Connected components of one image in ONE 1024-thread block: union-find over the runs, flatten + fold area / bbox into
// the roots, number the roots.  Images have a few hundred to a few thousand runs (the strips' columns plus noise that
// survived the opening), so the parent array normally sits in LDS and the CAS chains cost LDS latencies; an image with
// more than kLdsRuns runs works on the global array through the same code.
//   union  : runs in adjacent columns whose row ranges, widened by one, overlap are united (8-connectivity)
//   flatten: every run points at its root; area and bbox are folded into the root with wave-aggregated atomics (runs of
//            one component sit next to each other in the run list, so the lanes of a wave mostly share a root);
//            roots are listed and counted per row of their first pixel
//   rank   : scipy.ndimage.label numbers components in raster order of their first pixel = 1 + rank of the root's
//            (row, col): rows prefix-summed, roots dropped into their row's bucket, rank = roots in earlier rows + roots
//            of the same row further left
__global__ __launch_bounds__(1024) void k_components(Job j) {
    __shared__ int lds_parent[kLdsRuns];
    __shared__ uint32_t lds_se[kLdsRuns];                      // copies of run_se / run_col: every phase below reads them several
    __shared__ uint16_t lds_col[kLdsRuns];                     // times, through dependent chains (binary search, find, ranking)
    __shared__ uint32_t lds_rows[2 * kLdsRows];                // first the columns' offsets into the run list (scan, move, union), then
    uint32_t* const lds_rowcnt = lds_rows;                     // the roots per row of their first pixel -> offsets / fill cursors (rank)
    uint32_t* const lds_rowpos = lds_rows + kLdsRows;
    __shared__ uint32_t wsum[16];
    __shared__ int s_label00;
    __shared__ uint32_t s_nroots;                              // roots found so far (one LDS add per wave and round, not one global add per root)
    __shared__ int s_wcnt[16], s_base;
    const int img = blockIdx.x;
    const int H = j.H, W = j.W;
    uint32_t* off = j.col_cnt + (int64_t)img * (W + 1);
    const int64_t rbase = (int64_t)img * j.run_cap;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

    // ---- run list: exclusive scan of the per-column counts (in place, total at [W]), then the runs k_runs_count parked
    //      per column move to their place in the compact, column-major list
    {
        const int seg = (W + 1023) / 1024;
        const int c0 = min(W, t * seg), c1 = min(W, c0 + seg);
        uint32_t sum = 0;
        for (int c = c0; c < c1; ++c) sum += off[c];
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t run = inc - sum, total = 0;
        for (int q = 0; q < 16; ++q) { const uint32_t v = wsum[q]; total += v; if (q < wv) run += v; }
        const uint32_t cap = (uint32_t)j.run_cap;
        // the offsets also go to LDS for the column search below and for the union phase
        uint32_t* lds_off = lds_rows;
        const bool off_in_lds = W + 1 <= 2 * kLdsRows;
        for (int c = c0; c < c1; ++c) {
            const uint32_t cnt = off[c];
            off[c] = run;
            if (off_in_lds) lds_off[c] = run;
            run += cnt;
        }
        if (t == 0 && off_in_lds) lds_off[W] = total;
        if (t == 0) {
            off[W] = total;
            j.info[img].n_runs = (int)total;
            if (total > cap) atomicOr(&j.info[img].status, 2);
        }
        __threadfence_block();
        __syncthreads();
    }
    // One thread per run of the compact list (not per column: a noisy column parks dozens of runs, and every global load
    // waits out the memory system's latency under the other stream's image passes): the run's column by binary search
    // over the offsets, eight parked runs requested per thread before the first is stored.
    const bool off_in_lds = W + 1 <= 2 * kLdsRows;
    auto OFF = [&](int c) -> uint32_t { return off_in_lds ? lds_rows[c] : __hip_atomic_load(&off[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    {
        const int total = min((int)__hip_atomic_load(&off[W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j.run_cap);
        for (int base = 0; base < total; base += 8 * 1024) {
            uint32_t v[8]; int cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 1024 + t;
                v[u] = 0u; cc[u] = 0;
                if (i < total) {
                    int lo = 1, hi = W;                              // smallest c' with OFF(c') > i; the run's column is c' - 1
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (OFF(mid) > (uint32_t)i) hi = mid; else lo = mid + 1; }
                    cc[u] = lo - 1;
                    v[u] = j.run_tmp[rbase + (int64_t)cc[u] * j.runs_per_col + ((uint32_t)i - OFF(cc[u]))];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 1024 + t;
                if (i < total) {
                    const uint32_t se = v[u];
                    const int c = cc[u], s0 = (int)(se & 0xffffu), e0 = (int)(se >> 16);
                    const int64_t g = rbase + i;
                    j.run_se[g] = se;
                    j.run_col[g] = (uint16_t)c;
                    if (i < kLdsRuns) { lds_se[i] = se; lds_col[i] = (uint16_t)c; }
                    j.parent[g] = i;
                    j.area[g] = e0 - s0 + 1;
                    int4 bb; bb.x = s0; bb.y = c; bb.z = e0 + 1; bb.w = c + 1;
                    *(int4*)(j.bbox + g * 4) = bb;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    const int nr = min((int)__hip_atomic_load(&off[W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j.run_cap);
    const bool in_lds = nr <= kLdsRuns;
    int* parent = in_lds ? lds_parent : (j.parent + rbase);          // k_runs_fill initialised the global array to identity
    const bool rows_in_lds = H <= kLdsRows;                          // else the global arrays (zeroed by k_init_info)
    uint32_t* rowcnt = rows_in_lds ? lds_rowcnt : j.rowcnt + (int64_t)img * (H + 1);
    uint32_t* rowpos = rows_in_lds ? lds_rowpos : j.rowpos + (int64_t)img * (H + 1);
    if (in_lds) for (int i = t; i < nr; i += 1024) lds_parent[i] = i;
    if (t == 0) { s_label00 = 0; s_nroots = 0; }
    __syncthreads();
    auto SE = [&](int i) -> uint32_t { return in_lds ? lds_se[i] : j.run_se[rbase + i]; };
    auto COL = [&](int i) -> int { return in_lds ? (int)lds_col[i] : (int)j.run_col[rbase + i]; };

    // ---- union
    for (int i = t; i < nr; i += 1024) {
        const int c = COL(i);
        if (c + 1 >= j.W) continue;
        const uint32_t se = SE(i);
        const int a = (int)(se & 0xffffu), b = (int)(se >> 16);
        int lo = (int)OFF(c + 1);
        const int end = min((int)OFF(c + 2), j.run_cap);
        int hi = end;
        while (lo < hi) {                                   // first neighbour whose end >= a - 1
            const int mid = (lo + hi) >> 1;
            if ((int)(SE(mid) >> 16) >= a - 1) hi = mid; else lo = mid + 1;
        }
        for (int q = lo; q < end; ++q) {
            if ((int)(SE(q) & 0xffffu) > b + 1) break;
            uf_unite(parent, i, q, [&](int r) { return (SE(r) & 0xffffu) * (uint32_t)j.W + (uint32_t)COL(r); });
        }
    }
    __threadfence_block();
    __syncthreads();

    if (rows_in_lds) {                                               // the offsets' space becomes the row counts
        for (int r = t; r < H; r += 1024) lds_rowcnt[r] = 0u;
        __syncthreads();
    }
    // ---- flatten
    for (int i0 = 0; i0 < nr; i0 += 1024) {
        const int i = i0 + t;
        const bool active = i < nr;
        int r = i, s = 0, e = -1, c = 0;
        if (active) {
            r = uf_find_live(parent, i);
            const uint32_t se = SE(i);
            s = (int)(se & 0xffffu); e = (int)(se >> 16); c = COL(i);
            if (r != i) __hip_atomic_store(&parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // shortens later finds
            if (in_lds) j.parent[rbase + i] = r;                     // the global copy is what lfa_debug_fetch paints from
        }
        {                                                            // roots: their runs start at the components' raster-first pixels
            const bool root = active && r == i;
            const unsigned long long rb = __ballot(root);
            if (rb) {
                uint32_t base = 0;
                if (lane == __ffsll(rb) - 1) base = atomicAdd(&s_nroots, (uint32_t)__popcll(rb));
                base = __shfl(base, __ffsll(rb) - 1);
                if (root) {
                    atomicAdd(&rowcnt[s], 1u);
                    j.rootlist[rbase + base + __popcll(rb & ((1ull << lane) - 1ull))] = i;
                }
            }
        }
        const bool need = active && r != i;
        unsigned long long todo = __ballot(need);
        while (todo) {                                             // wave-uniform loop over the distinct roots
            const int leader = __ffsll(todo) - 1;
            const int lr = __shfl(r, leader);
            const bool mine = need && r == lr;
            const unsigned long long grp = __ballot(mine);
            const int len = (int)wave_sum_u32(mine ? (uint32_t)(e - s + 1) : 0u);
            const int mnr = wave_min_i32(mine ? s : 0x7fffffff), mnc = wave_min_i32(mine ? c : 0x7fffffff);
            const int mxr = wave_max_i32(mine ? e + 1 : -1), mxc = wave_max_i32(mine ? c + 1 : -1);
            if (lane == leader) {
                atomicAdd(&j.area[rbase + lr], len);
                int* bb = j.bbox + (rbase + lr) * 4;
                atomicMin(&bb[0], mnr); atomicMin(&bb[1], mnc); atomicMax(&bb[2], mxr); atomicMax(&bb[3], mxc);
            }
            todo &= ~grp;
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- rank
    const int K = min((int)s_nroots, j.run_cap);
    const int seg = (H + 1023) / 1024;
    const int r0 = min(H, t * seg), r1 = min(H, r0 + seg);
    uint32_t sum = 0;
    for (int r = r0; r < r1; ++r) sum += __hip_atomic_load(&rowcnt[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int q = 0; q < wv; ++q) run += wsum[q];
    for (int r = r0; r < r1; ++r) {                                  // rowcnt becomes the offsets, rowpos the fill cursors
        const uint32_t c = __hip_atomic_load(&rowcnt[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rowcnt[r] = run; rowpos[r] = run; run += c;
    }
    __threadfence_block();
    __syncthreads();
    int* bucket = in_lds ? lds_parent : j.bucket + rbase;            // the parents are flattened and copied out: their space is free
    for (int q = t; q < K; q += 1024) {
        const int i = __hip_atomic_load(&j.rootlist[rbase + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int s = (int)(SE(i) & 0xffffu);
        bucket[atomicAdd(&rowpos[s], 1u)] = i;
    }
    __threadfence_block();
    __syncthreads();
    for (int q = t; q < K; q += 1024) {
        const int i = __hip_atomic_load(&j.rootlist[rbase + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int s = (int)(SE(i) & 0xffffu), c = COL(i);
        const uint32_t lo = __hip_atomic_load(&rowcnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t hi = __hip_atomic_load(&rowpos[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t before = 0;
        for (uint32_t b = lo; b < hi; ++b) {
            const int o = __hip_atomic_load(&bucket[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before += (COL(o) < c) ? 1u : 0u;
        }
        const int lab = (int)(lo + before) + 1;
        j.root_of_label[rbase + lab - 1] = i;
        j.label_of_run[rbase + i] = lab;
        if (s == 0 && c == 0) s_label00 = 1;                // pixel (0,0) is foreground <=> it is the first pixel of label 1
    }
    __threadfence_block();
    __syncthreads();
    if (t == 0) { j.info[img].n_labels = K; j.info[img].label00 = s_label00; }

    // ---- candidate list for the region test (k_regions.hip)
    candidates_phase<1024>(j, img, K, s_label00, s_wcnt, &s_base);
}

hipError_t launch_label(const Job& j, hipStream_t s) {
    const int gx = (j.W + 255) / 256;
    LFA_LAUNCH("k_runs_count", k_runs_count, dim3(gx * j.n), dim3(256), 0, s, j, gx);
    LFA_LAUNCH("k_components", k_components, dim3(j.n), dim3(1024), 0, s, j);
    return hipGetLastError();
}

// ---- diagnostics: paint masks / labels / table for the parity tests ------------------------------------------------
__global__ void k_dbg_mask(Job j, int img, uint8_t* out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)j.H * j.W) return;
    const int r = (int)(p / j.W), c = (int)(p % j.W);
    if (j.pred_fused && j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Bins] == 2u) {      // the mask of this image never left the bit plane
        const uint8_t* plane = j.bins + (int64_t)img * (((int64_t)j.H * j.W) >> 3);
        out[p] = (uint8_t)((plane[p >> 3] >> (p & 7)) & 1u);
        return;
    }
    const uint64_t w = j.maskT[((int64_t)img * j.HB + (r >> 6)) * j.W + c];
    out[p] = (uint8_t)((w >> (r & 63)) & 1ull);
}

__global__ void k_dbg_paint(Job j, int img, int what, void* out) {
    const uint32_t* off = j.col_cnt + (int64_t)img * (j.W + 1);
    const int64_t rbase = (int64_t)img * j.run_cap;
    const int nr = min((int)off[j.W], j.run_cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) {
        const uint32_t se = j.run_se[rbase + i];
        const int s = (int)(se & 0xffffu), e = (int)(se >> 16), c = j.run_col[rbase + i];
        if (what == LFA_DBG_OPENED) {
            for (int r = s; r <= e; ++r) ((uint8_t*)out)[(int64_t)r * j.W + c] = 1;
        } else {
            const int root = j.parent[rbase + i];
            const int lab = j.label_of_run[rbase + root];
            for (int r = s; r <= e; ++r) ((int32_t*)out)[(int64_t)r * j.W + c] = lab;
        }
    }
}

__global__ void k_dbg_table(Job j, int img, int32_t* out, int kmax) {
    const int64_t rbase = (int64_t)img * j.run_cap;
    const int K = min(j.info[img].n_labels, kmax);
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < K; l += gridDim.x * blockDim.x) {
        const int root = j.root_of_label[rbase + l];
        const int* bb = j.bbox + (rbase + root) * 4;
        out[l * 5 + 0] = bb[0]; out[l * 5 + 1] = bb[1]; out[l * 5 + 2] = bb[2]; out[l * 5 + 3] = bb[3];
        out[l * 5 + 4] = j.area[rbase + root];
    }
}

__global__ void k_dbg_hist_abs(Job j, int img, uint32_t* out, int nb) {
    const uint32_t* gh = j.hist + (int64_t)img * j.hist_stride;
    const int xmin = j.info[img].xmin;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x)
        out[k] = (xmin >= 0 && xmin + k < j.hist_stride) ? gh[xmin + k] : 0u;
}

hipError_t launch_debug(const Job& j, int image, int what, void* out, int64_t out_bytes, hipStream_t s) {
    const int64_t npix = (int64_t)j.H * j.W;
    hipError_t e;
    switch (what) {
        case LFA_DBG_MASK:
            if (out_bytes < npix) return hipErrorInvalidValue;
            hipLaunchKernelGGL(k_dbg_mask, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, j, image, (uint8_t*)out);
            break;
        case LFA_DBG_OPENED:
            if (out_bytes < npix) return hipErrorInvalidValue;
            e = hipMemsetAsync(out, 0, npix, s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_dbg_paint, dim3(256), dim3(256), 0, s, j, image, what, out);
            break;
        case LFA_DBG_LABELS:
            if (out_bytes < npix * 4) return hipErrorInvalidValue;
            e = hipMemsetAsync(out, 0, npix * 4, s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_dbg_paint, dim3(256), dim3(256), 0, s, j, image, what, out);
            break;
        case LFA_DBG_TABLE:
            hipLaunchKernelGGL(k_dbg_table, dim3(64), dim3(256), 0, s, j, image, (int32_t*)out, (int)(out_bytes / 20));
            break;
        case LFA_DBG_HIST: {
            const int64_t nb = out_bytes / 4 < j.hist_stride ? out_bytes / 4 : j.hist_stride;
            if (j.hist_abs) {                                        // one-pass uint16 histogram: bins are absolute, hand out [xmin, ...)
                hipLaunchKernelGGL(k_dbg_hist_abs, dim3(64), dim3(256), 0, s, j, image, (uint32_t*)out, (int)nb);
                break;
            }
            e = hipMemcpyAsync(out, j.hist + (int64_t)image * j.hist_stride, nb * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
            break;
        }
        case LFA_DBG_ONEPASS:
            if (out_bytes < 4) return hipErrorInvalidValue;
            e = hipMemsetAsync(out, j.hist_abs ? 1 : 0, 1, s);
            if (e == hipSuccess) e = hipMemsetAsync((char*)out + 1, 0, 3, s);
            if (e != hipSuccess) return e;
            break;
        case LFA_DBG_PLANE:
            if (out_bytes < 4) return hipErrorInvalidValue;
            if ((j.C == 3 || j.u16_plane || j.pred_mode) && j.bins) e = hipMemcpyAsync(out, j.edges_i + (int64_t)image * kEdgesI + kEdgesI_Bins, 4, hipMemcpyDeviceToDevice, s);
            else e = hipMemsetAsync(out, 0, 4, s);
            if (e != hipSuccess) return e;
            break;
        case LFA_DBG_GREY:
            if (out_bytes < 4) return hipErrorInvalidValue;
            if (j.grey_mode) e = hipMemcpyAsync(out, j.edges_i + (int64_t)image * kEdgesI + kEdgesI_Grey, 4, hipMemcpyDeviceToDevice, s);
            else e = hipMemsetAsync(out, 0, 4, s);
            if (e != hipSuccess) return e;
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lfa
