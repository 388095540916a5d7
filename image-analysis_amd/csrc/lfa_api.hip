// lfa_api.hip -- host side of the C ABI declared in include/lfa.h: handle, workspace, argument checks, launch order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "lfa_internal.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define LFA_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return fail(LFA_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));          \
    } while (0)

// makes `device` current for the duration of a call and puts the caller's device back afterwards
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device); else if (err == hipSuccess) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define LFA_ON_DEVICE(dev)                                                                               \
    DeviceGuard guard_(dev);                                                                            \
    if (guard_.err != hipSuccess) return fail(LFA_ERR_HIP, "selecting device %d: %s", (dev), hipGetErrorString(guard_.err))

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

// per-kernel hipEvent brackets (lfa_timing_*); events are pooled and reused after every read
struct EventTimer : lfa::KernelTimer {
    struct Rec { std::string name; hipEvent_t a, b; };
    std::string filter;                // "*" = every kernel, else exact kernel name
    int every = 1, seen = 0;           // "name/N": every N-th launch of that kernel only (two event records per launch are not free)
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    bool open = false;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void begin(const char* name, hipStream_t s) override {
        open = false;
        if (filter != "*" && filter != name) return;
        if (every > 1 && (seen++ % every) != 0) return;
        Rec r{name, get(), get()};
        if (!r.a || !r.b) return;
        if (hipEventRecord(r.a, s) != hipSuccess) return;
        recs.push_back(r);
        open = true;
    }
    void end(hipStream_t s) override {
        if (!open) return;
        (void)hipEventRecord(recs.back().b, s);
        open = false;
    }
    ~EventTimer() override {
        for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

}  // namespace

thread_local lfa::KernelTimer* lfa::g_timer = nullptr;

// Workspace of lfa_analyze_batch_static: device memory that exists without any allocation call.  Every GPU the library
// runs on holds its own copy of this array, at its own address, until the process ends.  In host code the NAME below is a
// host-side stand-in, not a device address: host code uses the address hipGetSymbolAddress returns for the current
// device (static_scratch_ptr, looked up once per device in lfa_create) -- never the name -- as kernel argument and in
// hipMemsetAsync.
static __device__ __align__(256) unsigned char scratch[LFA_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};      // per device: hipGetSymbolAddress(HIP_SYMBOL(scratch)) with that device current

// call with `device` current
int static_scratch_lookup(int device) {
    if (device < 0 || device >= kMaxDevices) return LFA_OK;      // no static workspace beyond kMaxDevices devices (the entry point says so)
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[device]) return LFA_OK;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        g_err = std::string("hipGetSymbolAddress(scratch) on device ") + std::to_string(device) + ": " + hipGetErrorString(e);
        return LFA_ERR_HIP;
    }
    g_scratch_ptr[device] = ptr;
    return LFA_OK;
}
}  // namespace

// Per-call device workspace of the detect path.  A handle owns one in hipMalloc memory (grown by ensure_workspace);
// lfa_analyze_batch_static lays one out inside the library's static device array instead (carve_workspace).
struct Workspace {
    Buf info, hist, otsu_w, otsu_s, edges, edges_i, bins, maskT, col_cnt, run_tmp, run_se, run_col, parent, area, bbox, rowcnt, rowpos, rootlist, bucket, label_of_run, root_of_label, cand,
        cand_sum, cand_rs_off, rowsum, slot_rs_off, rois;
    std::vector<Buf*> all() {
        return {&info, &hist, &otsu_w, &otsu_s, &edges, &edges_i, &bins, &maskT, &col_cnt, &run_tmp, &run_se, &run_col, &parent, &area, &bbox, &rowcnt, &rowpos, &rootlist, &bucket, &label_of_run,
                &root_of_label, &cand, &cand_sum, &cand_rs_off, &rowsum, &slot_rs_off, &rois};
    }
};

struct LfaContext {
    int device = 0;
    int max_h = 0, max_w = 0, max_c = 0, max_batch = 0;
    int gray_variant = LFA_GRAY_FMA_102;
    int debug_flags = 0;
    Workspace ws;
    Buf lut8, lut16;
    Buf roi_idx;                           // max_batch + 2 words: ROI array index of lfa_profile_batch's band kernel
    hipEvent_t done_event = nullptr;       // end of the handle's most recent detect/analyze call (orders calls across streams)
    hipStream_t last_stream = nullptr;
    bool have_done = false;
    lfa::Job last{};
    bool have_last = false;
    EventTimer* timer = nullptr;
    hipEvent_t passes_event = nullptr;     // see lfa_stream_wait_passes
    LfaContext* after = nullptr;           // see lfa_chain_passes
    // Class plane on/off per call (colour images).  k_otsu_rgb counts the images that could use the plane; the two counters
    // are copied to pinned host memory behind the call's kernels and looked at -- without waiting -- when a later call is
    // set up: while fewer than half of the images could use it, the histogram pass does not write the plane at all
    // (photo-like inputs: 0.59 -> 0.42 ms per 256 images).  Results do not depend on the choice.
    Buf plane_stats;
    uint32_t* plane_host = nullptr;        // pinned, kStatsWords words
    hipEvent_t plane_event = nullptr;
    bool plane_pending = false, plane_on = true;
    // Equal-channel ("grey") mode: on while at least half of the colour images of the last completed call had three equal
    // channels.  Outside the mode that count comes from a sampled look (k_minmax_rgb, one round in four), inside it from
    // every pixel (k_grey_hist); when the mode was on and the exact count says "not grey" -- images that are grey where the
    // sample looks and coloured elsewhere -- the handle stays out of the mode for the next four calls.
    bool grey_on = false, pending_grey = false;
    int grey_backoff = 0;
    // Predicted-threshold mode (Job::pred_mode): the window is the range of the integer mask thresholds of the last completed call,
    // widened by 16384 (of G's 2 550 000 steps: a grey level and a half, about one histogram bin in mid-range) on either side.  When fewer than half of the images of a call in
    // the mode could use their bit plane (another scene, or too many pixels inside the window), the next eight calls go without.
    bool pred_have = false, pending_pred = false;
    uint32_t pred_lo = 0, pred_hi = 0, pred_lo_raw = 0, pred_hi_raw = 0;      // (_raw: the range without the margin)
    int pred_backoff = 0;
    // (uint16 frames on the one-pass histogram take the same mode with the pixel value in place of G: the window is the range of T = x_thr + 1
    // of the last completed call widened by max(4, that range) either way, and the handle tries only while at least half of that call's frames
    // had few pixels near their threshold (k_otsu_int).  pred_kind: what the window was measured on -- 2 colour uint8, 1 uint16, 0 none.)
    int pred_kind = 0, pending_pred_kind = 0;
    bool pred_sparse = false;
    Buf thr_list;                          // [max_batch] T of the frames of a uint16 call (Job::thr_list), copied to thr_host behind the counters
    uint32_t* thr_host = nullptr;          // pinned
    int pending_n = 0;
    // uint16 images: the histogram pass needs no min/max pass before it when the bins are the pixel values themselves (one
    // pass less over the image); worth it while the values stay below the LDS table of k_hist_u16 (camera data of 12-14
    // bits).  Chosen from the largest value of the last completed uint16 call, which travels with the counters above.
    bool u16_one_pass = false;
    int u16_max = 65535;                   // largest uint16 value of the last completed uint16 call
    // uint16 class plane (k_hist_u16 / k_mask): sixteen value ranges, narrow around the mean threshold of the last completed
    // uint16 call; written while at least half of that call's frames could use it (judged on the ranges that call had)
    bool u16_have_pred = false, u16_plane_on = true, pending_had_pred = false;
    int u16_thr_mean = 0, u16_thr_spread = 0;   // mean and (largest - smallest) threshold of that call
    int win_hint = 0;                      // longest analysis window of the last completed call (rows), 0 = unknown
    const double* host_pow = nullptr;      // lfa_set_host_pow: records for the next detect / analyze call (device memory), used once
    int host_pow_n = 0;
    // detect_lines workspace (lfa_detect_lines_batch), allocated by the first call that needs it
    bool ln_bbox_dirty = false;
    Buf ln_g, ln_u, ln_binlut, ln_maps, ln_levels, ln_lhist, ln_lval, ln_cls, ln_clsfn, ln_cimg, ln_parent, ln_bbox, ln_cand, ln_info;
    lfa::LinesJob lines_last{};
    bool have_lines = false;
    std::vector<Buf*> all() {
        std::vector<Buf*> v = ws.all();
        v.push_back(&lut8); v.push_back(&lut16); v.push_back(&plane_stats); v.push_back(&roi_idx); v.push_back(&thr_list);
        for (Buf* b : {&ln_g, &ln_u, &ln_binlut, &ln_maps, &ln_levels, &ln_lhist, &ln_lval, &ln_cls, &ln_clsfn, &ln_cimg, &ln_parent, &ln_bbox, &ln_cand, &ln_info})
            v.push_back(b);
        return v;
    }
};

namespace {

struct TimerScope {
    explicit TimerScope(LfaContext* h) { lfa::g_timer = h ? h->timer : nullptr; }
    ~TimerScope() { lfa::g_timer = nullptr; }
};

int ensure(Buf& b, size_t bytes) {
    if (bytes <= b.bytes) return LFA_OK;
    if (b.p) { LFA_HIP(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    bytes = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(LFA_ERR_NOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    b.bytes = bytes;
    return LFA_OK;
}

int run_capacity(int H, int W, int L) {
    // interior runs are >= L long and separated by a gap; at most one shorter run at each border
    const long long per_col = (long long)(H + 1) / (L + 1) + 2;
    long long cap = per_col * W;
    if (cap > 0x7fffffff) cap = 0x7fffffff;
    return (int)cap;
}

// Every workspace buffer of one call (n images of H x W x C of `dtype`, line_length L, max_strips) with its size in bytes.
// plane: with the 4-bit class plane; the float32 Otsu buffers only for 2-D images above 2^24 pixels (k_otsu_int).
struct WsNeeds {
    struct Item { Buf* b; size_t bytes; };
    Item item[32];
    int n = 0;
    void add(Buf& b, size_t bytes) { item[n].b = &b; item[n].bytes = bytes; ++n; }
};

WsNeeds workspace_needs(Workspace& w, int n, int H, int W, int C, int dtype, int L, int max_strips, bool plane) {
    const size_t N = (size_t)n;
    const int HB = (H + 63) / 64;
    const size_t cap = (size_t)run_capacity(H, W, L);
    const size_t hist_stride = (dtype == LFA_U16) ? lfa::kHistBinsU16 : 256;
    WsNeeds q;
    q.add(w.info, N * sizeof(LfaImageInfo));
    q.add(w.hist, N * hist_stride * 4);
    if (C == 1 && (size_t)H * W > ((size_t)1 << 24)) {          // float32-rounded Otsu weights (k_otsu_int)
        q.add(w.otsu_w, N * 2 * hist_stride * 4);
        q.add(w.otsu_s, N * 2 * hist_stride * 8);
    }
    q.add(w.edges, N * lfa::kEdges * 8);
    q.add(w.edges_i, N * lfa::kEdgesI * 4);
    if (plane) q.add(w.bins, N * (((size_t)H * W + 1) / 2));
    q.add(w.maskT, N * HB * W * 8);
    q.add(w.col_cnt, N * (W + 1) * 4);
    q.add(w.run_tmp, N * cap * 4);
    q.add(w.run_se, N * cap * 4);
    q.add(w.run_col, N * cap * 2);
    q.add(w.parent, N * cap * 4);
    q.add(w.area, N * cap * 4);
    q.add(w.bbox, N * cap * 16);
    q.add(w.rowcnt, N * ((size_t)H + 1) * 4);
    q.add(w.rowpos, N * ((size_t)H + 1) * 4);
    q.add(w.rootlist, N * cap * 4);
    q.add(w.bucket, N * cap * 4);
    q.add(w.label_of_run, N * cap * 4);
    q.add(w.root_of_label, N * cap * 4);
    const size_t ccap = (size_t)H * W / lfa::kAreaMin + 1;
    q.add(w.cand, N * ccap * 4);
    q.add(w.cand_sum, N * ccap * 3 * 8);
    q.add(w.cand_rs_off, N * ccap * 4);
    q.add(w.rowsum, N * (size_t)16 * H * 4);
    q.add(w.slot_rs_off, N * (size_t)max_strips * 4);
    q.add(w.rois, N * (size_t)max_strips * sizeof(LfaRoi));
    return q;
}

// the handle's own workspace: hipMalloc memory, grows only (a growing buffer is freed and allocated again -- hipFree waits
// for the device; lfa_create and lfa_reserve size everything up front so that calls within those limits never get here)
int ensure_workspace(LfaContext* h, int n, int H, int W, int C, int dtype, int L, int max_strips) {
    WsNeeds q = workspace_needs(h->ws, n, H, W, C, dtype, L, max_strips, /*plane=*/true);
    for (int k = 0; k < q.n; ++k) {
        int rc = ensure(*q.item[k].b, q.item[k].bytes);
        if (rc) return rc;
    }
    return LFA_OK;
}

int pixel_bytes(int dtype) { return dtype == LFA_U16 ? 2 : (dtype == LFA_F64 ? 8 : 1); }

int check_images(LfaContext* h, const void* images, int dtype, int n, int H, int W, int C, bool allow_f64 = false) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    if (!images) return fail(LFA_ERR_ARG, "null image pointer");
    if (dtype == LFA_F64 && !allow_f64) return fail(LFA_ERR_UNSUPPORTED, "float64 images are taken by lfa_detect_batch only");
    if (dtype != LFA_U8 && dtype != LFA_U16 && dtype != LFA_F64) return fail(LFA_ERR_ARG, "dtype must be LFA_U8, LFA_U16 or LFA_F64, got %d", dtype);
    if (n < 1 || n > h->max_batch) return fail(LFA_ERR_ARG, "n=%d outside [1, max_batch=%d]", n, h->max_batch);
    if (H < 1 || W < 1 || H > h->max_h || W > h->max_w) return fail(LFA_ERR_ARG, "image %dx%d outside the handle's %dx%d", H, W, h->max_h, h->max_w);
    if (C != 1 && C != 3) return fail(LFA_ERR_ARG, "C must be 1 (2-D image) or 3 (colour), got %d (skimage rgb2gray: the input array must have size 3 along channel_axis)", C);
    if (C > h->max_c) return fail(LFA_ERR_ARG, "C=%d exceeds the handle's max_c=%d", C, h->max_c);
    return LFA_OK;
}

lfa::Job make_job(LfaContext* h, const Workspace& w, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                  int L, int max_strips) {
    lfa::Job j{};
    j.images = (const uint8_t*)images;
    j.img_bytes = (int64_t)H * W * C * pixel_bytes(dtype);
    j.n = n; j.H = H; j.W = W; j.C = C; j.dtype = dtype;
    j.min_maxr = min_maxr; j.min_maxc = min_maxc; j.line_length = L;
    j.gray_variant = h->gray_variant;
    j.debug_flags = h->debug_flags;
    j.aligned16 = (((uintptr_t)images & 15) == 0 && (j.img_bytes & 15) == 0) ? 1 : 0;
    if ((C == 3 && dtype == LFA_U16) || dtype == LFA_F64) j.aligned16 = 0;   // uint16 colour, float64: the scalar float64 paths of every kernel (rare inputs, kept simple)
    j.HB = (H + 63) / 64;
    j.run_cap = run_capacity(H, W, L);
    j.max_strips = max_strips;
    j.info = (LfaImageInfo*)w.info.p;
    j.hist = (uint32_t*)w.hist.p;
    j.hist_stride = (dtype == LFA_U16 && C == 1) ? lfa::kHistBinsU16 : 256;      // (uint16 colour: the 256 float bins of the colour path)
    j.otsu_w = (float*)w.otsu_w.p;
    j.otsu_s = (double*)w.otsu_s.p;
    j.edges_g = (double*)w.edges.p;
    j.edges_i = (uint32_t*)w.edges_i.p;
    // class plane: colour images on the vector paths only (16-pixel groups in k_hist_rgb, 16-column groups in k_mask)
    j.passes_event = h->passes_event;
    j.wait_event = h->after ? h->after->passes_event : nullptr;
    j.plane_stats = (uint32_t*)h->plane_stats.p;
    j.win_hint = (h->debug_flags & 32) ? 0 : h->win_hint;
    j.u16_lds_bins = (h->debug_flags & 16) ? lfa::kLdsBinsU16 : h->u16_max + 1;
    j.hist_abs = (dtype == LFA_U16 && !(h->debug_flags & 8) && (h->u16_one_pass || (h->debug_flags & 16))) ? 1 : 0;
    if (dtype == LFA_U16 && C == 1 && j.hist_abs) {
        // class width 16 / 32 / 64 values for frames of up to 12 / 13 / 14 bits, the fourteen narrow classes centred on the
        // mean threshold of the last call
        // -- or, when the thresholds of the last call lay within 10 values of each other (a series of one scene), one value
        // per class: the threshold then is the last value of its class, the plane alone decides the mask (k_otsu_int)
        const int ps = h->u16_max < 4096 ? 4 : (h->u16_max < 8192 ? 5 : 6);
        if (h->u16_have_pred) {
            j.u16_pred = 1;
            j.u16_ps = h->u16_thr_spread <= 10 ? 0 : ps;
            j.u16_pa = h->u16_thr_mean - (j.u16_ps ? (15 << ps) / 2 : 7);
            if (j.u16_pa < 0) j.u16_pa = 0;
        } else if (h->debug_flags & 4) { j.u16_pred = 1; j.u16_ps = ps; j.u16_pa = 0; }
        j.u16_plane = (j.u16_pred && w.bins.bytes >= (size_t)n * (((size_t)H * W + 1) / 2) && j.aligned16 && (W % 16) == 0 && !(h->debug_flags & 2) && (h->u16_plane_on || (h->debug_flags & 4))) ? 1 : 0;
    }
    j.grey_mode = (C == 3 && dtype == LFA_U8 && j.aligned16 && !(h->debug_flags & 2048) && (h->grey_on || (h->debug_flags & 1024))) ? 1 : 0;
    const bool rgb_plane = C == 3 && dtype == LFA_U8 && j.aligned16 && (W % 16) == 0 && !(h->debug_flags & 2) && (h->plane_on || (h->debug_flags & 4));
    // predicted-threshold bit plane instead of the class plane (debug flag 8192: never; 32768: always, with a window that holds every
    // threshold and a list that holds every group -- each pixel then goes through k_mask_fix)
    const bool pred_shape = C == 3 && dtype == LFA_U8 && j.aligned16 && (W % 16) == 0 && !(h->debug_flags & (2 | 8192)) &&
                            w.bins.bytes >= (size_t)n * (((size_t)H * W + 1) / 2);
    const int64_t groups = (int64_t)H * W / 16, room = (int64_t)j.run_cap * 4;     // (the list's home holds 16 bytes per run slot)
    if (pred_shape && (h->debug_flags & 32768)) {
        j.pred_mode = 1; j.pred_lo = 0u; j.pred_hi = 0x400000u; j.pred_lo_raw = 0u; j.pred_hi_raw = 0x400000u;
        j.pred_cap = (int32_t)(groups < room ? groups : room);
    } else if (pred_shape && h->pred_have && h->pred_kind == 2 && h->pred_backoff == 0) {
        j.pred_mode = 1; j.pred_lo = h->pred_lo & ~1023u; j.pred_hi = (h->pred_hi + 1023u) & ~1023u;      // whole buckets of k_hist_rgb's table
        j.pred_cap = (int32_t)(groups / 16 < room ? groups / 16 : room);
        if (j.grey_mode) {
            // Grey mode: the thresholds of equal-channel images move in steps of a grey level (G = 10000 * b) and sit on populated levels,
            // so a margin would put whole grey levels on the list.  The window is the last call's range as it is -- a series with one
            // byte threshold then has no undecided pixel at all (k_grey_hist writes `b >= b_thr` and the mask pass reads 1/8 byte per
            // pixel instead of the byte plane's 1) -- and any other series falls back, at what the byte-plane path costs.
            j.pred_lo = (h->pred_lo_raw) & ~1023u; j.pred_hi = (h->pred_hi_raw + 1023u) & ~1023u;
            j.pred_lo_raw = h->pred_lo_raw; j.pred_hi_raw = h->pred_hi_raw;
        }
    }
    const bool pred_shape_u16 = C == 1 && dtype == LFA_U16 && j.hist_abs && j.aligned16 && (W % 16) == 0 && !(h->debug_flags & (2 | 8192)) &&
                                w.bins.bytes >= (size_t)n * (((size_t)H * W + 1) / 2);
    if (pred_shape_u16 && (h->debug_flags & 32768)) {
        j.pred_mode = 1; j.pred_lo = 0u; j.pred_hi = 65536u;
        j.pred_cap = (int32_t)(groups < room ? groups : room);
    } else if (pred_shape_u16 && h->pred_have && h->pred_kind == 1 && h->pred_backoff == 0 && h->pred_sparse) {
        j.pred_mode = 1; j.pred_lo = h->pred_lo; j.pred_hi = h->pred_hi;
        j.pred_cap = (int32_t)(groups / 16 < room ? groups / 16 : room);
    }
    if (j.pred_mode && dtype == LFA_U16) j.u16_plane = 0;          // (one plane per call: the bits take the class plane's place)
    if (C == 1 && dtype == LFA_U16 && j.hist_abs && h->pred_have && h->pred_kind == 1) { j.pred_probe_lo = h->pred_lo; j.pred_probe_hi = h->pred_hi; }
    j.thr_list = (C == 1 && dtype == LFA_U16 && j.hist_abs) ? (uint32_t*)h->thr_list.p : nullptr;
    j.pred_list = (uint32_t*)w.bbox.p;
    j.pred_stride = room;
    j.pred_fused = (j.pred_mode && (W % 64) == 0 && !(h->debug_flags & 65536)) ? 1 : 0;
    j.bins = (j.u16_plane || rgb_plane || j.pred_mode) ? (uint8_t*)w.bins.p : nullptr;
    j.maskT = (uint64_t*)w.maskT.p;
    j.col_cnt = (uint32_t*)w.col_cnt.p;
    j.run_tmp = (uint32_t*)w.run_tmp.p;
    j.runs_per_col = j.run_cap / W;
    j.run_se = (uint32_t*)w.run_se.p;
    j.run_col = (uint16_t*)w.run_col.p;
    j.parent = (int32_t*)w.parent.p;
    j.area = (int32_t*)w.area.p;
    j.bbox = (int32_t*)w.bbox.p;
    j.rowcnt = (uint32_t*)w.rowcnt.p;
    j.rowpos = (uint32_t*)w.rowpos.p;
    j.rootlist = (int32_t*)w.rootlist.p;
    j.bucket = (int32_t*)w.bucket.p;
    j.label_of_run = (int32_t*)w.label_of_run.p;
    j.root_of_label = (int32_t*)w.root_of_label.p;
    j.cand_cap = (int32_t)((int64_t)H * W / lfa::kAreaMin + 1);
    j.cand = (int32_t*)w.cand.p;
    j.cand_sum = (unsigned long long*)w.cand_sum.p;
    j.cand_rs_off = (int32_t*)w.cand_rs_off.p;
    j.rowsum = (uint32_t*)w.rowsum.p;
    j.rowsum_cap = 16 * H;
    j.slot_rs_off = (int32_t*)w.slot_rs_off.p;
    j.grey_plane_stride = (int64_t)j.run_cap * 16;
    j.grey_plane = (j.grey_mode && !j.pred_mode && j.grey_plane_stride >= (int64_t)H * W && !(h->debug_flags & 4096)) ? (uint8_t*)w.bbox.p : nullptr;   // (pred_mode: the list lives there)
    j.lut8 = (const uint8_t*)h->lut8.p;
    j.lut16 = (const uint16_t*)h->lut16.p;
    return j;
}

// Lay a workspace out inside `base` (capacity bytes, 256-byte aligned pieces).  Returns the bytes it needs; the Bufs are
// set only when that fits.  Allocates nothing.
size_t carve_workspace(Workspace& w, unsigned char* base, size_t capacity, int n, int H, int W, int C, int dtype, int L, int max_strips) {
    WsNeeds q = workspace_needs(w, n, H, W, C, dtype, L, max_strips, /*plane=*/false);
    size_t off = 0;
    for (int k = 0; k < q.n; ++k) off += (q.item[k].bytes + 255) & ~(size_t)255;
    if (off > capacity) return off;
    size_t at = 0;
    for (int k = 0; k < q.n; ++k) {
        q.item[k].b->p = base + at;
        q.item[k].b->bytes = q.item[k].bytes;
        at += (q.item[k].bytes + 255) & ~(size_t)255;
    }
    return off;
}

// static_ws: the workspace comes out of the library's static device array and the call allocates nothing -- no class
// plane, no min/max-free uint16 histogram, no counters travelling back to the handle, no kernel timing (all of which
// either need memory or are choices between paths with identical results).
int detect_impl(LfaContext* h, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                int L, LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                LfaRoi* out_rois, hipStream_t s, lfa::Job* job_out, bool static_ws = false, LfaRoi** ws_rois = nullptr, bool allow_f64 = false) {
    // tables of lfa_set_host_pow are for this call and no other, whether it goes through or not
    const double* host_pow = h ? h->host_pow : nullptr;
    const int host_pow_n = h ? h->host_pow_n : 0;
    if (h) { h->host_pow = nullptr; h->host_pow_n = 0; }
    int rc = check_images(h, images, dtype, n, H, W, C, allow_f64);
    if (rc) return rc;
    if (host_pow && host_pow_n != n)
        return fail(LFA_ERR_ARG, "lfa_set_host_pow gave %d record(s), this call has %d image(s); the tables are withdrawn", host_pow_n, n);
    if (H > 65535 || W > 65535) return fail(LFA_ERR_ARG, "image sides above 65535 are not supported");
    if ((long long)H * W > (1ll << 30)) return fail(LFA_ERR_UNSUPPORTED, "images above 2^30 pixels are not supported");
    if (n > 65535 || (H + 7) / 8 > 65535) return fail(LFA_ERR_UNSUPPORTED, "more than 65535 frames per call, or frames taller than 524280 rows, are not supported");
    if (L < 1) return fail(LFA_ERR_ARG, "line_length must be >= 1");
    if (min_maxr < 0 || min_maxc < 0) return fail(LFA_ERR_ARG, "min_maxr/min_maxc must be >= 0");
    if (!out_strips || !out_counts) return fail(LFA_ERR_ARG, "null output pointer");
    if (max_strips < 1 || max_strips > 1024) return fail(LFA_ERR_ARG, "max_strips must be in [1,1024]");
    LFA_ON_DEVICE(h->device);
    // The counters of the handle's previous call select this call's paths, so the set-up WAITS for them (a 48-byte copy that
    // trails that call's kernels): which path a call takes is then a function of the calls made on the handle, not of how far
    // the device had got when the next call was set up.  The host therefore runs at most one call ahead per handle; callers
    // that pipeline use several handles (bench.py, frames.analyze_stream), and the device still has the other handles' calls
    // queued while this one waits.  LFA_POLL_COUNTERS=1 restores the non-blocking look (hipEventQuery) for comparison.
    static const bool poll = [] { const char* e = getenv("LFA_POLL_COUNTERS"); return e && *e == '1'; }();
    // (not in a static call: it takes fixed paths, and a host wait would not be legal under stream capture)
    if (!static_ws && h->plane_pending && (poll ? hipEventQuery(h->plane_event) == hipSuccess : hipEventSynchronize(h->plane_event) == hipSuccess)) {
        h->plane_pending = false;
        if (h->plane_host[1]) h->plane_on = 2u * h->plane_host[0] >= h->plane_host[1];
        if (h->plane_host[1]) {
            const bool mostly = 2u * h->plane_host[11] >= h->plane_host[1];
            if (h->pending_grey && !mostly) h->grey_backoff = 4;
            else if (h->grey_backoff > 0) --h->grey_backoff;
            h->grey_on = mostly && h->grey_backoff == 0;
        }
        {
            const uint32_t tmin = 0x7fffffffu - h->plane_host[12], tmax = h->plane_host[13];
            h->pred_have = h->plane_host[12] != 0u && tmin <= tmax;
            h->pred_kind = h->pred_have ? h->pending_pred_kind : 0;
            if (h->pred_have && h->pred_kind == 1) {
                // uint16: the tightest cluster of the call's T values -- the largest set within 32 values of each other --, provided it
                // holds at least half of the frames; widened by max(4, its own range) either way
                std::vector<uint32_t> T;
                for (int k = 0; k < h->pending_n; ++k) if (h->thr_host[k]) T.push_back(h->thr_host[k]);
                std::sort(T.begin(), T.end());
                size_t best_i = 0, best_c = 0;
                for (size_t i = 0, e = 0; i < T.size(); ++i) {
                    while (e < T.size() && T[e] - T[i] <= 32u) ++e;
                    if (e - i > best_c) { best_c = e - i; best_i = i; }
                }
                h->pred_have = best_c > 0 && 2 * best_c >= (size_t)h->pending_n;
                if (h->pred_have) {
                    const uint32_t c_lo = T[best_i], c_hi = T[best_i + best_c - 1];
                    const uint32_t margin = (h->debug_flags & 16384) ? 0u : (c_hi - c_lo > 4u ? c_hi - c_lo : 4u);
                    h->pred_lo = c_lo > margin ? c_lo - margin : 0u;
                    h->pred_hi = c_hi + margin;
                } else h->pred_kind = 0;
            } else if (h->pred_have) {
                const uint32_t margin = (h->debug_flags & 16384) ? 0u : 16384u;
                h->pred_lo = tmin > margin ? tmin - margin : 0u;
                h->pred_hi = tmax + margin;
                h->pred_lo_raw = tmin; h->pred_hi_raw = tmax;
            }
            h->pred_sparse = h->plane_host[17] != 0u && 2u * h->plane_host[16] >= h->plane_host[17];
            if (h->pending_pred && h->plane_host[14] && 2u * h->plane_host[15] < h->plane_host[14]) h->pred_backoff = 8;
            else if (h->pred_backoff > 0) --h->pred_backoff;
        }
        if (h->plane_host[3]) { h->u16_max = (int)h->plane_host[2]; h->u16_one_pass = h->plane_host[2] < (uint32_t)lfa::kLdsBinsU16; }
        if (h->plane_host[4]) h->win_hint = (int)h->plane_host[4];
        if (h->plane_host[8]) {
            h->u16_thr_mean = (int)(h->plane_host[5] / h->plane_host[8]); h->u16_have_pred = true;
            h->u16_thr_spread = (int)h->plane_host[10] - (65535 - (int)h->plane_host[9]);
        }
        if (h->plane_host[7]) h->u16_plane_on = !h->pending_had_pred || 2u * h->plane_host[6] >= h->plane_host[7];
    }
    lfa::Job j;
    if (static_ws) {
        unsigned char* base = (h->device < kMaxDevices) ? (unsigned char*)g_scratch_ptr[h->device] : nullptr;
        if (!base) return fail(LFA_ERR_UNSUPPORTED, "no static workspace on device %d (devices 0..%d have one)", h->device, kMaxDevices - 1);
        Workspace w;
        const size_t need = carve_workspace(w, base, (size_t)LFA_STATIC_SCRATCH_BYTES, n, H, W, C, dtype, L, max_strips);
        if (need > (size_t)LFA_STATIC_SCRATCH_BYTES)
            return fail(LFA_ERR_ARG, "lfa_analyze_batch_static: %d image(s) of %dx%dx%d with line_length %d and max_strips %d need %zu bytes of workspace, "
                        "the static workspace holds LFA_STATIC_SCRATCH_BYTES = %zu; pass fewer or smaller images, or use lfa_analyze_batch",
                        n, H, W, C, L, max_strips, need, (size_t)LFA_STATIC_SCRATCH_BYTES);
        j = make_job(h, w, images, dtype, n, H, W, C, min_maxr, min_maxc, L, max_strips);
        // no events either: nothing is recorded for, or awaited from, other handles (lfa_stream_wait_passes /
        // lfa_chain_passes do not see static calls), so the call is kernels and memsets on `s` only and can be captured
        j.passes_event = nullptr; j.wait_event = nullptr;
        j.pred_mode = 0; j.pred_fused = 0; j.pred_list = nullptr; j.thr_list = nullptr; j.pred_probe_lo = j.pred_probe_hi = 0u;
        j.plane_stats = nullptr; j.bins = nullptr; j.u16_plane = 0; j.hist_abs = 0; j.win_hint = 0; j.grey_mode = 0; j.grey_plane = nullptr;
        j.u16_lds_bins = lfa::kLdsBinsU16; j.u16_pred = 0; j.u16_pa = 0; j.u16_ps = 0;
        if (ws_rois) *ws_rois = (LfaRoi*)w.rois.p;
    } else {
        // (after the counters: the workspace depends on the paths they select)
        h->have_last = false;                        // lfa_debug_fetch must not follow pointers a failed call may have freed
        if ((rc = ensure_workspace(h, n, H, W, C, dtype, L, max_strips))) return rc;
        j = make_job(h, h->ws, images, dtype, n, H, W, C, min_maxr, min_maxc, L, max_strips);
        // the handle's workspace is shared by its calls: a call on another stream than the last one waits for that one
        if (h->have_done && s != h->last_stream) LFA_HIP(hipStreamWaitEvent(s, h->done_event, 0));
        if (ws_rois) *ws_rois = (LfaRoi*)h->ws.rois.p;
    }
    j.host_pow = (C == 3 || dtype == LFA_F64) ? host_pow : nullptr;
    if (!out_rois && ws_rois) out_rois = *ws_rois;          // lfa_analyze_batch without a caller's ROI buffer
    lfa::OutPtrs o{out_strips, out_counts, out_info, out_rois};
    TimerScope ts(static_ws ? nullptr : h);
    LFA_HIP(lfa::launch_threshold(j, s));
    LFA_HIP(lfa::launch_mask(j, s));
    LFA_HIP(lfa::launch_label(j, s));
    LFA_HIP(lfa::launch_regions(j, o, s));
    if (j.plane_stats && !h->plane_pending) {
        LFA_HIP(hipMemcpyAsync(h->plane_host, h->plane_stats.p, lfa::kStatsWords * 4, hipMemcpyDeviceToHost, s));
        h->pending_had_pred = h->u16_have_pred;
        h->pending_grey = j.grey_mode != 0;
        h->pending_pred = j.pred_mode != 0;
        h->pending_pred_kind = (C == 3 && dtype == LFA_U8) ? 2 : ((C == 1 && dtype == LFA_U16) ? 1 : 0);
        if (j.thr_list) LFA_HIP(hipMemcpyAsync(h->thr_host, h->thr_list.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        h->pending_n = j.thr_list ? n : 0;
        LFA_HIP(hipEventRecord(h->plane_event, s));
        h->plane_pending = true;
    }
    h->last = j; h->have_last = true;
    if (job_out) *job_out = j;
    return LFA_OK;
}

}  // namespace

extern "C" {

int lfa_version(void) { return LFA_VERSION; }

const char* lfa_last_error(void) { return g_err.c_str(); }

int lfa_create(int device, int max_h, int max_w, int max_c, int max_batch, LfaHandle* out) {
    if (!out) return fail(LFA_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (max_h < 1 || max_w < 1 || max_batch < 1 || max_batch > 65535 || (max_c != 1 && max_c != 3))
        return fail(LFA_ERR_ARG, "lfa_create: bad limits (max_h=%d max_w=%d max_c=%d max_batch=%d)", max_h, max_w, max_c, max_batch);
    int ndev = 0;
    LFA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(LFA_ERR_ARG, "device %d not in [0,%d)", device, ndev);
    LFA_ON_DEVICE(device);
    LfaContext* h = new (std::nothrow) LfaContext();
    if (!h) return fail(LFA_ERR_NOMEM, "out of host memory");
    h->device = device; h->max_h = max_h; h->max_w = max_w; h->max_c = max_c; h->max_batch = max_batch;
    if (const char* e = getenv("LFA_DEBUG_FLAGS")) h->debug_flags = atoi(e);      // A/B runs of unmodified programs (lfa_set_debug_flags overrides)
    int rc;
    if ((rc = ensure(h->lut8, 256)) || (rc = ensure(h->lut16, 65536 * 2))) { lfa_destroy(h); return rc; }
    // default tables from the C library's pow(); the Python binding overrides them with numpy's
    std::vector<uint8_t> l8(256);
    for (int k = 0; k < 256; ++k) {
        const double x = (k == 255) ? 1.0 : (double)k * (1.0 / 255.0) + 0.0;       // np.linspace(0, 1, 256)
        double v = 255.0 * std::pow(x, 0.4);
        v = std::nearbyint(v);
        l8[k] = (uint8_t)(v < 255.0 ? v : 255.0);
    }
    std::vector<uint16_t> l16(65536);
    for (int k = 0; k < 65536; ++k) l16[k] = (uint16_t)(std::pow((double)k / 65535.0, 0.4) * 65535.0 * 1);
    if (hipMemcpy(h->lut8.p, l8.data(), 256, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->lut16.p, l16.data(), 65536 * 2, hipMemcpyHostToDevice) != hipSuccess) {
        lfa_destroy(h);
        return fail(LFA_ERR_HIP, "uploading gamma tables failed");
    }
    if ((rc = static_scratch_lookup(device))) { lfa_destroy(h); return rc; }      // set-up of lfa_analyze_batch_static, once per device
    // the whole workspace for calls within the handle's limits at the defaults (line_length >= 10, max_strips <= 64), class
    // plane and big-image Otsu buffers included; lfa_reserve widens those two limits
    if ((rc = ensure_workspace(h, max_batch, max_h, max_w, max_c, LFA_U16, 10, 64))) { lfa_destroy(h); return rc; }
    if (hipEventCreateWithFlags(&h->done_event, hipEventDisableTiming) != hipSuccess) {
        h->done_event = nullptr; lfa_destroy(h);
        return fail(LFA_ERR_HIP, "hipEventCreate failed");
    }
    if ((rc = ensure(h->plane_stats, lfa::kStatsWords * 4))) { lfa_destroy(h); return rc; }
    if ((rc = ensure(h->roi_idx, ((size_t)max_batch + 2) * 4))) { lfa_destroy(h); return rc; }
    if (hipHostMalloc((void**)&h->plane_host, lfa::kStatsWords * 4, hipHostMallocDefault) != hipSuccess) {
        h->plane_host = nullptr; lfa_destroy(h);
        return fail(LFA_ERR_NOMEM, "hipHostMalloc failed");
    }
    for (int k = 0; k < lfa::kStatsWords; ++k) h->plane_host[k] = 0u;
    if ((rc = ensure(h->thr_list, (size_t)max_batch * 4))) { lfa_destroy(h); return rc; }
    if (hipHostMalloc((void**)&h->thr_host, (size_t)max_batch * 4, hipHostMallocDefault) != hipSuccess) {
        h->thr_host = nullptr; lfa_destroy(h);
        return fail(LFA_ERR_NOMEM, "hipHostMalloc failed");
    }
    if (hipEventCreateWithFlags(&h->plane_event, hipEventDisableTiming) != hipSuccess) {
        h->plane_event = nullptr; lfa_destroy(h);
        return fail(LFA_ERR_HIP, "hipEventCreate failed");
    }
    if (hipEventCreateWithFlags(&h->passes_event, hipEventDisableTiming) != hipSuccess) {
        h->passes_event = nullptr; lfa_destroy(h);
        return fail(LFA_ERR_HIP, "hipEventCreate failed");
    }
    *out = h;
    return LFA_OK;
}

int lfa_destroy(LfaHandle h) {
    if (!h) return LFA_OK;
    DeviceGuard guard_(h->device);
    if (h->plane_event && h->plane_pending) (void)hipEventSynchronize(h->plane_event);   // the counters' copy reads plane_stats and writes plane_host
    for (Buf* b : h->all()) if (b->p) (void)hipFree(b->p);
    if (h->passes_event) (void)hipEventDestroy(h->passes_event);
    if (h->done_event) (void)hipEventDestroy(h->done_event);
    if (h->plane_event) (void)hipEventDestroy(h->plane_event);
    if (h->plane_host) (void)hipHostFree(h->plane_host);
    if (h->thr_host) (void)hipHostFree(h->thr_host);
    delete h->timer;
    delete h;
    return LFA_OK;
}

int lfa_set_gamma_lut(LfaHandle h, int dtype, const void* lut, int n) {
    if (!h || !lut) return fail(LFA_ERR_ARG, "null argument");
    LFA_ON_DEVICE(h->device);
    if (dtype == LFA_U8) {
        if (n != 256) return fail(LFA_ERR_ARG, "uint8 table needs 256 entries, got %d", n);
        const uint8_t* t = (const uint8_t*)lut;
        for (int k = 1; k < 256; ++k) if (t[k] < t[k - 1]) return fail(LFA_ERR_ARG, "gamma table must be non-decreasing");
        LFA_HIP(hipMemcpy(h->lut8.p, lut, 256, hipMemcpyHostToDevice));
    } else if (dtype == LFA_U16) {
        if (n != 65536) return fail(LFA_ERR_ARG, "uint16 table needs 65536 entries, got %d", n);
        const uint16_t* t = (const uint16_t*)lut;
        for (int k = 1; k < 65536; ++k) if (t[k] < t[k - 1]) return fail(LFA_ERR_ARG, "gamma table must be non-decreasing");
        LFA_HIP(hipMemcpy(h->lut16.p, lut, 65536 * 2, hipMemcpyHostToDevice));
    } else {
        return fail(LFA_ERR_ARG, "bad dtype %d", dtype);
    }
    return LFA_OK;
}

int lfa_set_gray_variant(LfaHandle h, int variant) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    if (variant < 0 || variant >= LFA_GRAY_NVARIANTS) return fail(LFA_ERR_ARG, "gray variant %d not in [0,%d)", variant, LFA_GRAY_NVARIANTS);
    h->gray_variant = variant;
    return LFA_OK;
}

int lfa_set_host_pow(LfaHandle h, const LfaHostPow* tables, int n) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    static_assert(sizeof(LfaHostPow) == lfa::kHostPowDoubles * 8, "LfaHostPow layout");
    if (!tables) { h->host_pow = nullptr; h->host_pow_n = 0; return LFA_OK; }
    if (n < 1 || n > h->max_batch) return fail(LFA_ERR_ARG, "n=%d outside [1, max_batch=%d]", n, h->max_batch);
    if ((uintptr_t)tables & 7) return fail(LFA_ERR_ARG, "tables must be 8-byte aligned");
    h->host_pow = (const double*)tables; h->host_pow_n = n;
    return LFA_OK;
}

int lfa_set_debug_flags(LfaHandle h, int flags) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    h->debug_flags = flags;
    return LFA_OK;
}

int lfa_detect_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                     int line_length, LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                     void* stream) {
    int rc = detect_impl(h, images, dtype, n, H, W, C, min_maxr, min_maxc, line_length, out_strips, max_strips, out_counts,
                         out_info, nullptr, (hipStream_t)stream, nullptr, false, nullptr, /*allow_f64=*/true);
    if (rc) return rc;
    LFA_ON_DEVICE(h->device);
    LFA_HIP(hipEventRecord(h->done_event, (hipStream_t)stream));
    h->last_stream = (hipStream_t)stream; h->have_done = true;
    return LFA_OK;
}

static int profile_impl(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, const LfaRoi* rois,
                        int n_rois, const void* luts, int n_luts, const int32_t* roi_lut, bool with_lut, int inverse, int range,
                        double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream) {
    int rc = check_images(h, images, dtype, n, H, W, C);
    if (rc) return rc;
    if (n_rois < 0) return fail(LFA_ERR_ARG, "n_rois < 0");
    if (with_lut && (!luts || n_luts < 1)) return fail(LFA_ERR_ARG, "luts must hold at least one table");
    if (n_rois == 0) return LFA_OK;
    if (!rois || !out_profiles || !out_peaks) return fail(LFA_ERR_ARG, "null pointer");
    if (profile_stride < 1) return fail(LFA_ERR_ARG, "profile_stride must be >= 1");
    LFA_ON_DEVICE(h->device);
    lfa::Job j{};
    j.images = (const uint8_t*)images;
    j.img_bytes = (int64_t)H * W * C * (dtype == LFA_U16 ? 2 : 1);
    j.n = n; j.H = H; j.W = W; j.C = C; j.dtype = dtype;
    j.aligned16 = (((uintptr_t)images & 15) == 0 && (j.img_bytes & 15) == 0) ? 1 : 0;
    // the band kernel's ROI index lives in the handle: ordered with the handle's other calls like the detect workspace
    if (h->have_done && (hipStream_t)stream != h->last_stream) LFA_HIP(hipStreamWaitEvent((hipStream_t)stream, h->done_event, 0));
    TimerScope ts(h);
    LFA_HIP(lfa::launch_profiles(j, rois, n_rois, nullptr, 0, nullptr, inverse, range, out_profiles, profile_stride, out_peaks,
                                 with_lut ? (const uint8_t*)luts : nullptr, n_luts, with_lut ? roi_lut : nullptr,
                                 (hipStream_t)stream, (int32_t*)h->roi_idx.p));
    LFA_HIP(hipEventRecord(h->done_event, (hipStream_t)stream));
    h->last_stream = (hipStream_t)stream; h->have_done = true;
    return LFA_OK;
}

int lfa_profile_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, const LfaRoi* rois,
                      int n_rois, int inverse, int range, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                      void* stream) {
    return profile_impl(h, images, dtype, n, H, W, C, rois, n_rois, nullptr, 0, nullptr, false, inverse, range, out_profiles,
                        profile_stride, out_peaks, stream);
}

int lfa_profile_lut_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, const LfaRoi* rois,
                          int n_rois, const void* luts, int n_luts, const int32_t* roi_lut, int inverse, int range,
                          double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream) {
    return profile_impl(h, images, dtype, n, H, W, C, rois, n_rois, luts, n_luts, roi_lut, true, inverse, range, out_profiles,
                        profile_stride, out_peaks, stream);
}

int lfa_profile_any_batch(LfaHandle h, const void* regions, int elem_type, int n, int H, int W, int C, const LfaRoi* rois,
                          int n_rois, int inverse, int range, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                          void* stream) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    if (elem_type == LFA_U8 || elem_type == LFA_U16)
        return lfa_profile_batch(h, regions, elem_type, n, H, W, C, rois, n_rois, inverse, range, out_profiles, profile_stride, out_peaks, stream);
    if (elem_type < LFA_E_I8 || elem_type > LFA_E_F64) return fail(LFA_ERR_ARG, "unknown element type %d", elem_type);
    if (!regions || !rois || !out_profiles || !out_peaks) return fail(LFA_ERR_ARG, "null pointer");
    if (n < 1 || H < 1 || W < 1 || (C != 1 && C != 3 && C != 4 && C != 2)) return fail(LFA_ERR_ARG, "bad shape n=%d %dx%dx%d", n, H, W, C);
    if (n_rois < 0 || profile_stride < 1) return fail(LFA_ERR_ARG, "bad n_rois / profile_stride");
    LFA_ON_DEVICE(h->device);
    TimerScope ts(h);
    hipError_t e = lfa::launch_profiles_any(regions, elem_type, n, H, W, C, rois, n_rois, inverse, range, out_profiles, profile_stride,
                                            out_peaks, (hipStream_t)stream);
    if (e != hipSuccess) return fail(LFA_ERR_HIP, "lfa_profile_any_batch: %s", hipGetErrorString(e));
    return LFA_OK;
}

int lfa_peaks_batch(LfaHandle h, const double* profiles, const int32_t* lengths, int n_profiles, int profile_stride,
                    int inverse, int window_size, int two_peaks, int range, LfaPeaks* out_peaks, void* stream) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    if (n_profiles < 0) return fail(LFA_ERR_ARG, "n_profiles < 0");
    if (n_profiles == 0) return LFA_OK;
    if (!profiles || !lengths || !out_peaks) return fail(LFA_ERR_ARG, "null pointer");
    if (profile_stride < 1 || window_size < 0) return fail(LFA_ERR_ARG, "bad stride / window");
    LFA_ON_DEVICE(h->device);
    TimerScope ts(h);
    LFA_HIP(lfa::launch_peaks(profiles, lengths, n_profiles, profile_stride, inverse, window_size, two_peaks, range,
                              out_peaks, (hipStream_t)stream));
    return LFA_OK;
}

static int analyze_impl(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                        int line_length, LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                        LfaRoi* out_rois, double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream, bool static_ws) {
    if (!out_profiles || !out_peaks) return fail(LFA_ERR_ARG, "null output pointer");
    if (profile_stride < 1) return fail(LFA_ERR_ARG, "profile_stride must be >= 1");
    lfa::Job j{};
    LfaRoi* ws_rois = nullptr;                  // the workspace's own ROI buffer, used when the caller passes none
    int rc = detect_impl(h, images, dtype, n, H, W, C, min_maxr, min_maxc, line_length, out_strips, max_strips, out_counts,
                         out_info, out_rois, (hipStream_t)stream, &j, static_ws, &ws_rois);
    if (rc) return rc;
    if (!out_rois) out_rois = ws_rois;
    LFA_ON_DEVICE(h->device);
    TimerScope ts(static_ws ? nullptr : h);
    LFA_HIP(lfa::launch_profiles(j, out_rois, n * max_strips, out_counts, max_strips, j.slot_rs_off, 0, 15, out_profiles, profile_stride,
                                 out_peaks, nullptr, 0, nullptr, (hipStream_t)stream));
    if (!static_ws) {
        LFA_HIP(hipEventRecord(h->done_event, (hipStream_t)stream));
        h->last_stream = (hipStream_t)stream; h->have_done = true;
    }
    return LFA_OK;
}

int lfa_analyze_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                      int line_length, LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                      LfaRoi* out_rois, double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream) {
    return analyze_impl(h, images, dtype, n, H, W, C, min_maxr, min_maxc, line_length, out_strips, max_strips, out_counts, out_info,
                        out_rois, out_profiles, profile_stride, out_peaks, stream, false);
}

int lfa_analyze_batch_static(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, int min_maxr, int min_maxc,
                             int line_length, LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                             LfaRoi* out_rois, double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream) {
    return analyze_impl(h, images, dtype, n, H, W, C, min_maxr, min_maxc, line_length, out_strips, max_strips, out_counts, out_info,
                        out_rois, out_profiles, profile_stride, out_peaks, stream, true);
}

int64_t lfa_static_workspace_bytes(void) { return (int64_t)LFA_STATIC_SCRATCH_BYTES; }

int lfa_reserve(LfaHandle h, int max_strips, int min_line_length) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    if (max_strips < 1 || max_strips > 1024) return fail(LFA_ERR_ARG, "max_strips must be in [1,1024]");
    if (min_line_length < 1) return fail(LFA_ERR_ARG, "min_line_length must be >= 1");
    LFA_ON_DEVICE(h->device);
    int rc = ensure_workspace(h, h->max_batch, h->max_h, h->max_w, h->max_c, LFA_U16, min_line_length, max_strips);
    if (rc) return rc;
    return ensure_workspace(h, h->max_batch, h->max_h, h->max_w, h->max_c, LFA_U8, min_line_length, max_strips);
}

int64_t lfa_workspace_bytes(LfaHandle h) {
    if (!h) return 0;
    int64_t total = 0;
    for (Buf* b : h->all()) total += (int64_t)b->bytes;
    return total;
}

int lfa_gaussian_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C, double sigma,
                       const double* host_weights, int radius, void* out, int out_f32, void* stream) {
    int rc = check_images(h, images, dtype, n, H, W, C);
    if (rc) return rc;
    if (!out) return fail(LFA_ERR_ARG, "null output pointer");
    double w[2 * lfa::kGaussMaxRadius + 1];
    if (host_weights) {
        if (radius < 1 || radius > lfa::kGaussMaxRadius) return fail(LFA_ERR_UNSUPPORTED, "radius %d outside [1,%d]", radius, lfa::kGaussMaxRadius);
        for (int k = 0; k < 2 * radius + 1; ++k) w[k] = host_weights[k];
        for (int k = 0; k < radius; ++k)
            if (w[k] != w[2 * radius - k]) return fail(LFA_ERR_UNSUPPORTED, "kernel taps must be symmetric (scipy's correlate1d symmetric path)");
    } else {
        if (!(sigma > 0.0)) return fail(LFA_ERR_ARG, "sigma must be > 0");
        radius = (int)(4.0 * sigma + 0.5);
        if (radius < 1 || radius > lfa::kGaussMaxRadius) return fail(LFA_ERR_UNSUPPORTED, "sigma %.3f needs radius %d outside [1,%d]", sigma, radius, lfa::kGaussMaxRadius);
        // scipy.ndimage._filters._gaussian_kernel1d: exp(-0.5 / sigma^2 * x^2), normalised by the plain sum
        const double sigma2 = sigma * sigma;
        double sum = 0.0;
        for (int k = -radius; k <= radius; ++k) { w[k + radius] = std::exp(-0.5 / sigma2 * (double)(k * k)); }
        for (int k = 0; k < 2 * radius + 1; ++k) sum += w[k];
        for (int k = 0; k < 2 * radius + 1; ++k) w[k] /= sum;
    }
    LFA_ON_DEVICE(h->device);
    lfa::Job j{};
    j.images = (const uint8_t*)images;
    j.img_bytes = (int64_t)H * W * C * (dtype == LFA_U16 ? 2 : 1);
    j.n = n; j.H = H; j.W = W; j.C = C; j.dtype = dtype;
    j.aligned16 = (((uintptr_t)images & 15) == 0 && (j.img_bytes & 15) == 0) ? 1 : 0;
    TimerScope ts(h);
    LFA_HIP(lfa::launch_gaussian_w(j, w, radius, out, out_f32, (hipStream_t)stream));
    return LFA_OK;
}

int lfa_detect_lines_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W,
                           const double* w_sigma1, const double* w_local,
                           LfaLineBox* out_boxes, int max_boxes, int32_t* out_counts, LfaLinesInfo* out_info, void* stream) {
    int rc = check_images(h, images, dtype, n, H, W, 1);
    if (rc) return rc;
    if (!out_boxes || !out_counts) return fail(LFA_ERR_ARG, "null output pointer");
    if (max_boxes < 1 || max_boxes > 65536) return fail(LFA_ERR_ARG, "max_boxes must be in [1,65536]");
    if ((long long)H * W > (1ll << 30)) return fail(LFA_ERR_UNSUPPORTED, "images above 2^30 pixels are not supported");
    if (n > 65535 || (H + 7) / 8 > 65535) return fail(LFA_ERR_UNSUPPORTED, "more than 65535 frames per call, or frames taller than 524280 rows, are not supported");
    // scipy.ndimage._filters._gaussian_kernel1d: exp(-0.5 / sigma^2 * x^2), normalised by the plain sum
    auto taps = [](double sigma, int radius, double* w) {
        const double sigma2 = sigma * sigma;
        double sum = 0.0;
        for (int k = -radius; k <= radius; ++k) w[k + radius] = std::exp(-0.5 / sigma2 * (double)(k * k));
        for (int k = 0; k < 2 * radius + 1; ++k) sum += w[k];
        for (int k = 0; k < 2 * radius + 1; ++k) w[k] /= sum;
    };
    double w9[9], w3[3];
    if (w_sigma1) { for (int k = 0; k < 9; ++k) w9[k] = w_sigma1[k]; } else taps(1.0, 4, w9);
    if (w_local) { for (int k = 0; k < 3; ++k) w3[k] = w_local[k]; } else taps((3 - 1) / 6.0, 1, w3);
    for (int k = 0; k < 4; ++k) if (w9[k] != w9[8 - k]) return fail(LFA_ERR_UNSUPPORTED, "kernel taps must be symmetric (scipy's correlate1d symmetric path)");
    if (w3[0] != w3[2]) return fail(LFA_ERR_UNSUPPORTED, "kernel taps must be symmetric (scipy's correlate1d symmetric path)");
    LFA_ON_DEVICE(h->device);
    hipStream_t s = (hipStream_t)stream;
    lfa::LinesJob j{};
    j.n = n; j.H = H; j.W = W;
    j.kh = H / 8 > 1 ? H / 8 : 1; j.kw = W / 8 > 1 ? W / 8 : 1;                  // _adapthist.py:78-79
    j.sy = j.kh / 2; j.sx = j.kw / 2;                                             // :123
    const int ey = (j.kh - H % j.kh) % j.kh + (j.kh + 1) / 2, ex = (j.kw - W % j.kw) % j.kw + (j.kw + 1) / 2;   // :125-126
    j.nty = (j.sy + H + ey) / j.kh - 1; j.ntx = (j.sx + W + ex) / j.kw - 1;       // :141
    if (j.nty < 1 || j.ntx < 1) return fail(LFA_ERR_UNSUPPORTED, "image %dx%d too small for CLAHE tiles", H, W);
    {
        const double c = 0.05 * (double)((long long)j.kh * j.kw);                // :155
        j.clim = c < 1.0 ? 1 : (int)c;
    }
    j.cap = max_boxes;
    j.w0 = w3[1]; j.w1 = w3[0];
    const size_t N = (size_t)n, npix = (size_t)H * W;
    const bool keep_gauss = (h->debug_flags & 64) != 0;                           // lfa_lines_debug_fetch(LFA_LDBG_GAUSS) needs it
    h->have_lines = false;                                                        // lfa_lines_debug_fetch must not follow pointers a failed call may have freed
    if ((keep_gauss && (rc = ensure(h->ln_g, N * npix * 8))) || (rc = ensure(h->ln_u, N * npix * 2)) || (rc = ensure(h->ln_binlut, N * 65536)) ||
        (rc = ensure(h->ln_maps, N * (size_t)j.nty * j.ntx * 256 * 4)) || (rc = ensure(h->ln_levels, N * npix * 2)) ||
        (rc = ensure(h->ln_lhist, N * 16384 * 4)) || (rc = ensure(h->ln_lval, N * 16384 * 8)) || (rc = ensure(h->ln_cls, N * 16384 * 2)) || (rc = ensure(h->ln_clsfn, N * 16)) || (rc = ensure(h->ln_cimg, N * npix * 2)) ||
        (rc = ensure(h->ln_parent, N * npix * 4)) ||
        (rc = ensure(h->ln_cand, N * (size_t)j.cap * 12)) || (rc = ensure(h->ln_info, N * sizeof(LfaLinesInfo))))
        return rc;
    // the box extents table is all zero between calls (k_ccl_boxes / k_ccl_collect, k_lines.hip): cleared when it is
    // allocated, and again after a call whose launches did not all go out
    const size_t bbox_was = h->ln_bbox.bytes;                                     // (not the pointer: a larger block may start where the old one did)
    if ((rc = ensure(h->ln_bbox, N * npix * 16))) return rc;
    const bool clear_bbox = h->ln_bbox.bytes != bbox_was || h->ln_bbox_dirty;
    j.g = keep_gauss ? (double*)h->ln_g.p : nullptr; j.u = (uint16_t*)h->ln_u.p; j.binlut = (uint8_t*)h->ln_binlut.p; j.maps = (int32_t*)h->ln_maps.p;
    j.levels = (uint16_t*)h->ln_levels.p; j.lhist = (uint32_t*)h->ln_lhist.p; j.lval = (double*)h->ln_lval.p; j.cls = (uint16_t*)h->ln_cls.p; j.clsfn = (int32_t*)h->ln_clsfn.p; j.force_table = (h->debug_flags & 128) ? 1 : 0; j.cimg = (uint16_t*)h->ln_cimg.p;
    j.parent = (int32_t*)h->ln_parent.p; j.bbox = (int32_t*)h->ln_bbox.p; j.cand = (int32_t*)h->ln_cand.p; j.info = (LfaLinesInfo*)h->ln_info.p;
    j.images = (const uint8_t*)images; j.img_bytes = (int64_t)H * W * (dtype == LFA_U16 ? 2 : 1); j.dtype = dtype;
    for (int k = 0; k < 5; ++k) j.wg[k] = w9[k];
    if (h->have_done && s != h->last_stream) LFA_HIP(hipStreamWaitEvent(s, h->done_event, 0));
    if (clear_bbox) { LFA_HIP(hipMemsetAsync(h->ln_bbox.p, 0, h->ln_bbox.bytes, s)); h->ln_bbox_dirty = false; }
    TimerScope ts(h);
    h->ln_bbox_dirty = true;
    LFA_HIP(lfa::launch_lines(j, out_boxes, out_counts, max_boxes, s));
    h->ln_bbox_dirty = false;
    if (out_info) LFA_HIP(hipMemcpyAsync(out_info, j.info, N * sizeof(LfaLinesInfo), hipMemcpyDeviceToDevice, s));
    LFA_HIP(hipEventRecord(h->done_event, s));
    h->last_stream = s; h->have_done = true;
    h->lines_last = j; h->have_lines = true;
    return LFA_OK;
}

int lfa_lines_debug_fetch(LfaHandle h, int image_index, int what, void* out, int64_t out_bytes, void* stream) {
    if (!h || !out) return fail(LFA_ERR_ARG, "null argument");
    if (!h->have_lines) return fail(LFA_ERR_ARG, "no lfa_detect_lines_batch call has run on this handle yet");
    const lfa::LinesJob& j = h->lines_last;
    if (image_index < 0 || image_index >= j.n) return fail(LFA_ERR_ARG, "image_index %d outside the last batch of %d", image_index, j.n);
    const int64_t npix = (int64_t)j.H * j.W;
    const void* src = nullptr;
    int64_t bytes = 0;
    switch (what) {
        case LFA_LDBG_GAUSS:
            if (!j.g) return fail(LFA_ERR_ARG, "the Gaussian image is kept only when debug flag 64 was set before the call (lfa_set_debug_flags)");
            src = j.g + image_index * npix; bytes = npix * 8; break;
        case LFA_LDBG_U16:    src = j.u + image_index * npix; bytes = npix * 2; break;
        case LFA_LDBG_LEVELS: src = j.levels + image_index * npix; bytes = npix * 2; break;
        case LFA_LDBG_ROOTS:  src = j.parent + image_index * npix; bytes = npix * 4; break;
        case LFA_LDBG_LVAL:   src = j.lval + (int64_t)image_index * 16384; bytes = 16384 * 8; break;
        default: return fail(LFA_ERR_ARG, "lfa_lines_debug_fetch: unknown item %d", what);
    }
    if (out_bytes < bytes) return fail(LFA_ERR_ARG, "lfa_lines_debug_fetch(what=%d) needs %lld bytes, got %lld", what, (long long)bytes, (long long)out_bytes);
    LFA_ON_DEVICE(h->device);
    LFA_HIP(hipMemcpyAsync(out, src, (size_t)bytes, hipMemcpyDefault, (hipStream_t)stream));
    return LFA_OK;
}

int lfa_debug_fetch(LfaHandle h, int image_index, int what, void* out, int64_t out_bytes, void* stream) {
    if (!h || !out) return fail(LFA_ERR_ARG, "null argument");
    if (!h->have_last) return fail(LFA_ERR_ARG, "no detect/analyze call has run on this handle yet");
    if (image_index < 0 || image_index >= h->last.n) return fail(LFA_ERR_ARG, "image_index %d outside the last batch of %d", image_index, h->last.n);
    LFA_ON_DEVICE(h->device);
    hipError_t e = lfa::launch_debug(h->last, image_index, what, out, out_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? LFA_ERR_ARG : LFA_ERR_HIP, "lfa_debug_fetch(what=%d): %s", what, hipGetErrorString(e));
    return LFA_OK;
}

int lfa_timing_enable(LfaHandle h, const char* kernel_filter) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    LFA_ON_DEVICE(h->device);
    if (!kernel_filter || !*kernel_filter) {
        if (h->timer) { LFA_HIP(hipDeviceSynchronize()); delete h->timer; h->timer = nullptr; }
        return LFA_OK;
    }
    if (!h->timer) h->timer = new (std::nothrow) EventTimer();
    if (!h->timer) return fail(LFA_ERR_NOMEM, "out of host memory");
    h->timer->filter = kernel_filter;
    h->timer->every = 1; h->timer->seen = 0;
    const size_t slash = h->timer->filter.find('/');
    if (slash != std::string::npos) {
        const int n = std::atoi(h->timer->filter.c_str() + slash + 1);
        if (n < 1) return fail(LFA_ERR_ARG, "lfa_timing_enable: \"%s\": the launch stride after '/' must be >= 1", kernel_filter);
        h->timer->every = n;
        h->timer->filter.resize(slash);
    }
    return LFA_OK;
}

int lfa_timing_read(LfaHandle h, int max_entries, char* names, double* total_ms, int32_t* launches, int32_t* n_out) {
    if (!h || !names || !total_ms || !launches || !n_out) return fail(LFA_ERR_ARG, "null argument");
    *n_out = 0;
    if (!h->timer) return LFA_OK;
    LFA_ON_DEVICE(h->device);
    LFA_HIP(hipDeviceSynchronize());
    EventTimer* t = h->timer;
    int n = 0;
    for (auto& r : t->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        int k = 0;
        for (; k < n; ++k) if (std::strncmp(names + 32 * k, r.name.c_str(), 31) == 0) break;
        if (k == n) {
            if (n >= max_entries) { t->pool.push_back(r.a); t->pool.push_back(r.b); continue; }
            std::memset(names + 32 * n, 0, 32);
            std::strncpy(names + 32 * n, r.name.c_str(), 31);
            total_ms[n] = 0.0; launches[n] = 0; ++n;
        }
        total_ms[k] += ms; launches[k] += 1;
        t->pool.push_back(r.a); t->pool.push_back(r.b);
    }
    t->recs.clear();
    *n_out = n;
    return LFA_OK;
}

int lfa_stream_wait_passes(LfaHandle producer, void* stream) {
    if (!producer) return fail(LFA_ERR_ARG, "null handle");
    LFA_ON_DEVICE(producer->device);
    LFA_HIP(hipStreamWaitEvent((hipStream_t)stream, producer->passes_event, 0));
    return LFA_OK;
}

int lfa_chain_passes(LfaHandle consumer, LfaHandle producer) {
    if (!consumer) return fail(LFA_ERR_ARG, "null handle");
    if (producer == consumer) return fail(LFA_ERR_ARG, "a handle cannot follow itself (its calls are ordered already)");
    if (producer && producer->device != consumer->device) return fail(LFA_ERR_ARG, "handles on devices %d and %d: chaining is per device", producer->device, consumer->device);
    consumer->after = producer;
    return LFA_OK;
}

int lfa_sync(LfaHandle h, void* stream) {
    if (!h) return fail(LFA_ERR_ARG, "null handle");
    LFA_ON_DEVICE(h->device);
    LFA_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LFA_OK;
}

}  // extern "C"
