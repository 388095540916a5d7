// lfa_internal.h -- workspace layout and kernel launchers shared by the translation units of liblfa_hip.so
#ifndef LFA_INTERNAL_H_
#define LFA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lfa.h"

namespace lfa {

constexpr int kHistBinsU16 = 65536;
constexpr int kLdsBinsU16 = 16384;        // k_hist_u16 counts this many consecutive values in LDS, the rest with global atomics
constexpr int kEdges = 257;              // 256 float bins -> 257 edges
constexpr int kEdgesI = 264;             // integer edge record per image: 257 thresholds + the three words below
constexpr int kEdgesI_Crit = 257;        // != 0: some histogram edge is critical -> float64 histogram path
constexpr int kEdgesI_MaskThr = 258;     // integer form of the mask threshold
constexpr int kEdgesI_MaskCrit = 259;    // != 0: mask threshold critical -> float64 mask path
constexpr int kEdgesI_Bins = 260;        // != 0: the image's plane (Job::bins) is valid and k_mask may use it: 1 = class plane, 2 = predicted-threshold
                                         // bit plane (Job::pred_mode)
constexpr int kEdgesI_Class = 261;       // uint16 class plane: the class q of the image's threshold x_thr; bit 8: x_thr is the last value
                                         // of class q, so that class is clear as a whole and k_mask never looks at the frame (k_otsu_int)
constexpr int kEdgesI_Grey = 262;        // colour uint8 images on the vector path: 1 while no pixel with unequal channel bytes has been seen
                                         // (k_init_info sets it, k_minmax_rgb / k_grey_hist clear it)
constexpr int kEdgesI_PredCnt = 263;     // predicted-threshold mode: entries k_hist_rgb asked for in the image's list of undecided pixel groups
                                         // (k_init_info clears it; above Job::pred_cap = the list overflowed)
constexpr int kHostPowDoubles = 2 + 257 + 256;   // sizeof(LfaHostPow) / 8
constexpr int kStatsWords = 18;          // Job::plane_stats
constexpr int kAreaMin = 100;            // detection.py:81
constexpr int kMaxCand = 1 << 20;

// Device-side description of one batch call.  Passed BY VALUE to every kernel (fits the kernarg segment).
struct Job {
    // input
    const uint8_t* images;   // batch-major
    int64_t img_bytes;       // bytes per image
    int32_t n, H, W, C;
    int32_t dtype;           // LFA_U8 / LFA_U16
    int32_t min_maxr, min_maxc, line_length;
    int32_t gray_variant;
    int32_t debug_flags;     // bit0: force the float64 colour path (as if every threshold were critical)
    int32_t aligned16;       // images base and per-image / per-row strides allow 16-byte vector loads
    // derived
    int32_t HB;              // row blocks of 64 rows
    int32_t run_cap;         // run capacity per image
    int32_t max_strips;
    // workspace (all per image, image i at i*stride)
    LfaImageInfo* info;      // [n]
    uint32_t* hist;          // [n][65536]      (uint16)  or [n][256]
    int32_t hist_stride;
    float* otsu_w;           // [n][2][hist_stride]  float32 weight1 / weight2 prefix arrays, images above 2^24 pixels only
    double* otsu_s;          // [n][2][hist_stride]  the matching cumsum(counts*centers) arrays
    double* edges_g;         // [n][257]  gray-space bin edges (colour path)
    uint32_t* edges_i;       // [n][264]  the same as integer thresholds on G = 2125*b0 + 7154*b1 + 721*b2 (k_threshold.hip)
    hipEvent_t passes_event; // recorded on the launch stream right after the histogram pass (lfa_stream_wait_passes); may be null
    hipEvent_t wait_event;   // the image passes of this call start after it (lfa_chain_passes); may be null
    uint32_t* plane_stats;   // [kStatsWords]  [12] 0x7fffffff - smallest and [13] largest integer mask threshold of the call's images (colour uint8: G_thr,
                             //      k_otsu_rgb; uint16 frames on the one-pass histogram: T = x_thr + 1 of the frames with few pixels near it, k_otsu_int),
                             //      [14] images the predicted-threshold plane was written for, [15] those whose mask was decided from it; [16] uint16 frames the
                             //      handle's candidate window (pred_probe_lo / _hi) would have held, [17] frames that was looked at; [11] colour images of this call with three equal channels throughout; [4] longest analysis window of the call in rows (k_finalize); [0] colour images of this call whose mask could be decided from the class plane, [1] colour images
                             //      seen (k_otsu_rgb counts), [2] largest pixel value of the call's uint16 images, [3] uint16 images
                             //      seen (k_otsu_int); [5] sum, [8] number, [9] 65535 - smallest and [10] largest of the uint16 thresholds x_thr, [6] uint16 images whose
                             //      mask could be decided from the class plane around u16_pa, [7] images that test was made on;
                             //      the host reads them later to choose the next call's paths
    int32_t win_hint;        // > 0: longest analysis window (rows) of the handle's last completed call -- sizes the LDS profile slots of
                             //      k_profile / k_peaks below the worst case 0.35*H (longer windows take the global-memory path)
    int32_t u16_lds_bins;    // one-pass uint16 histogram: values below this are counted in LDS (largest value of the last call + 1)
    int32_t hist_abs;        // uint16: the histogram is indexed by the pixel value itself and k_otsu_int finds the occupied range
                             //         from it -- no min/max pass (values >= kLdsBinsU16 go to global atomics)
    int32_t u16_pa, u16_ps;  // uint16 class plane: class(v) = min(max(v - u16_pa, 0) >> u16_ps, 15) -- sixteen value ranges, narrow
                             //         around the mean threshold of the handle's last call
    int32_t u16_pred;        // u16_pa / u16_ps are set (a call has completed before, or debug flag 4)
    int32_t u16_plane;       // k_hist_u16 (one-pass form) writes that class of every pixel into `bins`
    uint8_t* bins;           // [n][H*W/2]  colour path: class plane, (histogram bin >> 4) of every pixel in 4 bits, written by
                             //             k_hist_rgb and read by k_mask instead of the image (0.5 byte per pixel instead of 3);
                             //             null = not used
    uint64_t* maskT;         // [n][HB][W]      vertical 64-row words, thresholded image
    uint32_t* col_cnt;       // [n][W+1]        runs per column -> exclusive offsets
    uint32_t* run_tmp;       // [n][run_cap]    runs parked per column (runs_per_col slots each) between counting and compaction
    int32_t runs_per_col;    // run_cap / W
    uint32_t* run_se;        // [n][run_cap]    start | (end_inclusive << 16)
    uint16_t* run_col;       // [n][run_cap]
    int32_t* parent;         // [n][run_cap]
    int32_t* area;           // [n][run_cap]    valid at roots
    int32_t* bbox;           // [n][run_cap][4] minr, minc, maxr, maxc (half-open) valid at roots
    uint32_t* rowcnt;        // [n][H+1]        roots (components) whose first pixel lies in row r; element H = number of roots
    uint32_t* rowpos;        // [n][H+1]        exclusive prefix of rowcnt, then the fill cursor of every row's bucket
    int32_t* rootlist;       // [n][run_cap]    the root runs in discovery order
    int32_t* bucket;         // [n][run_cap]    the same grouped by row of their first pixel
    int32_t* label_of_run;   // [n][run_cap]    label id, written for root runs only
    int32_t* root_of_label;  // [n][run_cap]
    int32_t cand_cap;        // candidates per image: H*W/100 + 1 (disjoint components of >= 100 pixels)
    int32_t* cand;           // [n][cand_cap]   labels (1-based) with area >= 100, ascending
    unsigned long long* cand_sum;  // [n][cand_cap][3] integer sums of the original image over each candidate's box
    int32_t* cand_rs_off;    // [n][cand_cap]   offset of the candidate's window row sums in `rowsum`, or -1
    uint32_t* rowsum;        // [n][rowsum_cap] integer row sums of analysis windows (written by k_box_sums)
    int32_t rowsum_cap;      // 16 * H
    int32_t* slot_rs_off;    // [n][max_strips] the same offset per output strip slot (written by k_finalize)
    int32_t grey_mode;       // colour uint8 images are expected to have three equal channels (what cv2.imread makes of a grey file):
                             //         k_grey_hist takes their byte histogram in ONE pass and k_minmax_rgb / k_hist_rgb skip them
    uint8_t* grey_plane;     // grey mode: [n][grey_plane_stride] channel 0 of every pixel, written by k_grey_hist and compared by k_mask instead of
    int64_t grey_plane_stride;   //         the three channels (1 byte per pixel instead of 3).  Lives in the `bbox` buffer, which nothing uses before
                             //         the labelling stage; null when that buffer is smaller than H*W bytes per image (line_length > 15)
    // (uint16 frames, one-pass histogram: the same with the pixel value in place of G and T = x_thr + 1 in place of G_thr -- the mask is
    // `pixel >= T` --, any window ends; k_hist_u16 writes the bits and the list, k_otsu_int decides.)
    // Predicted-threshold mode (colour uint8 images, vector path): the handle's last completed call gives a window [pred_lo, pred_hi] -- both
    // multiples of 1024, the bucket size of k_hist_rgb's table -- in which this call's integer mask thresholds G_thr are expected.  k_hist_rgb
    // then writes ONE bit per pixel, G >= pred_hi, into `bins` instead of the 4-bit class, and appends the 16-pixel groups holding a pixel with
    // pred_lo <= G < pred_hi to the image's list.  An image whose G_thr falls inside the window and whose list did not overflow gets its mask
    // from those bits (k_mask) plus an exact compare of the listed groups' pixels (k_mask_fix); any other image is compared on the image
    // itself.  1/8 byte per pixel written and read instead of 1/2.
    int32_t pred_mode;
    uint32_t pred_lo, pred_hi;
    uint32_t* pred_list;     // [n][pred_stride] indices of the listed groups; lives in the `bbox` buffer like grey_plane
    int64_t pred_stride;
    int32_t pred_cap;        // entries an image may use
    int32_t pred_fused;      // pred_mode and W % 64 == 0: an image decided from the bit plane keeps its mask THERE -- k_mask skips it, k_mask_fix sets
                             //         the listed pixels in the plane, k_runs_count turns 64 x 64 bit blocks in the wave (debug flag 65536: off)
    uint32_t pred_lo_raw, pred_hi_raw;   // grey mode: the window before it was widened to whole buckets -- k_grey_hist works in grey levels
                             //         (b_lo = ceil(pred_lo_raw / 10000), b_hi likewise), and a bucket edge must not pull a whole level onto the list
    uint32_t* thr_list;      // [n] uint16, one-pass histogram: T = x_thr + 1 of every frame with few pixels near it, else 0 (k_otsu_int) -> the handle's next window
    uint32_t pred_probe_lo, pred_probe_hi;   // uint16, one-pass histogram: the window the handle would use next -- k_otsu_int reports how many
                             //         frames of this call it would have held (plane_stats[16], [17]); hi <= lo: none
    const double* host_pow;  // [n] LfaHostPow records (lfa_set_host_pow) or null: vmin, vmax, 257 gray-space edges, 256 gray-space thresholds
    const uint8_t* lut8;     // [256]
    const uint16_t* lut16;   // [65536]
};

struct OutPtrs {
    LfaStrip* strips;        // [n][max_strips]
    int32_t* counts;         // [n]
    LfaImageInfo* info;      // [n] or null
    LfaRoi* rois;            // [n][max_strips] or null
};

// Optional per-kernel timing (diagnostics for bench.py): when the calling thread has a timer installed, every launch
// made through LFA_LAUNCH is bracketed by two hipEventRecord calls on the launch stream.
struct KernelTimer {
    virtual void begin(const char* name, hipStream_t s) = 0;
    virtual void end(hipStream_t s) = 0;
    virtual ~KernelTimer() {}
};
extern thread_local KernelTimer* g_timer;

#define LFA_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                         \
    do {                                                                                \
        if (lfa::g_timer) lfa::g_timer->begin(NAME, STREAM);                            \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);              \
        if (lfa::g_timer) lfa::g_timer->end(STREAM);                                    \
    } while (0)

// --- launchers (each enqueues on `s` and returns the launch status) -----------------------------------------------
hipError_t launch_threshold(const Job& j, hipStream_t s);                 // min/max, histogram, Otsu -> info
hipError_t launch_mask(const Job& j, hipStream_t s);                      // thresholded image -> maskT
hipError_t launch_label(const Job& j, hipStream_t s);                     // opening + runs + union-find + ranks
hipError_t launch_regions(const Job& j, const OutPtrs& o, hipStream_t s); // candidates, box means, accept, sort
hipError_t launch_profiles(const Job& j, const LfaRoi* rois, int n_rois, const int32_t* counts_or_null,
                           int max_strips_or_0, const int32_t* rs_off_or_null, int inverse, int rng, double* profiles,
                           int stride, LfaPeaks* peaks, const uint8_t* luts_or_null, int n_luts, const int32_t* roi_lut_or_null,
                           hipStream_t s, int32_t* roi_idx_or_null = nullptr);   // roi_idx: n + 2 words of scratch -> band kernel allowed
hipError_t launch_peaks(const double* profiles, const int32_t* lengths, int n, int stride, int inverse, int window,
                        int two, int rng, LfaPeaks* out, hipStream_t s);
hipError_t launch_profiles_any(const void* data, int elem, int n, int H, int W, int C, const LfaRoi* rois, int n_rois, int inverse,
                               int rng, double* profiles, int stride, LfaPeaks* peaks, hipStream_t s);
hipError_t launch_gaussian_w(const Job& j, const double* w, int radius, void* out, int out_f32, hipStream_t s);
constexpr int kGaussMaxRadius = 8;
hipError_t launch_debug(const Job& j, int image, int what, void* out, int64_t out_bytes, hipStream_t s);

// detect_lines (k_lines.hip): one batch call, passed by value
struct LinesJob {
    int32_t n, H, W;
    int32_t kh, kw;          // CLAHE tile: max(H // 8, 1) x max(W // 8, 1)
    int32_t sy, sx;          // rows / columns padded in front: kh // 2, kw // 2
    int32_t nty, ntx;        // tiles with a histogram: padded size / tile - 1
    int32_t clim;            // clip limit in counts: int(clip(0.05 * kh * kw, 1, None))
    int32_t cap;             // candidate boxes kept per image
    double w0, w1;           // centre and side tap of the 3-tap Gaussian of threshold_local
    double wg[5];            // taps 0..4 of the 9-tap sigma-1 Gaussian (wg[4] = centre; symmetric)
    const uint8_t* images;   // [n] frames, uint8 or uint16 (dtype), C-contiguous
    int64_t img_bytes;
    int32_t dtype;
    double* g;               // [n][H*W]     gaussian(image), kept only for lfa_lines_debug_fetch (null otherwise)
    uint16_t* u;             // [n][H*W]     img_as_uint of the gamma-adjusted image
    uint8_t* binlut;         // [n][65536]   uint16 value -> CLAHE bin
    int32_t* maps;           // [n][nty*ntx][256] tile mappings
    uint16_t* levels;        // [n][H*W]     _clahe's result
    uint32_t* lhist;         // [n][16384]   histogram of levels, then its inclusive prefix
    double* lval;            // [n][16384]   final float value per level
    uint16_t* cls;           // [n][16384]   class per level (0 = background)
    int32_t* clsfn;          // [n][4]       a, m, valid: class = min(max(level - a, 0), m) where the table has that form
    int32_t force_table;     // debug flag 128: always look the class up
    uint16_t* cimg;          // [n][H*W]     class per pixel (k_ccl_init)
    int32_t* parent;         // [n][H*W]     union-find over pixels
    int32_t* bbox;           // [n][H*W][4]  at roots: rows down, columns left, columns right of the root (all zero between calls)
    int32_t* cand;           // [n][cap][3]  root, minr, maxr of the boxes that pass the filter
    LfaLinesInfo* info;      // [n]
};
hipError_t launch_lines(const LinesJob& j, LfaLineBox* out, int32_t* counts, int max_boxes, hipStream_t s);

}  // namespace lfa
#endif
