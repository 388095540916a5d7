/*
 * lfa.h -- C ABI of the MI355X-native LFA (lateral-flow-assay) analysis hot path.
 *
 * The reference (bwasden2003/Image-Analysis) has no FFI layer: its boundary for this path is the set of
 * module-level Python functions in LFA-Image_Analysis/detection.py that the GUI calls.  This header is what
 * a binding for those functions talks to (ctypes stub: image-analysis_amd/_capi.py, and INTEGRATION.md).
 * Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 on success or a negative LFA_ERR_* code,
 *     lfa_last_error() returns a thread-local message for the last failing call on this thread;
 *   - image / ROI / output pointers are DEVICE pointers (hipMalloc memory, e.g. torch.Tensor.data_ptr())
 *     unless a parameter is documented as host memory; images are C-contiguous, batch-major:
 *     image i starts at byte i * H*W*C*sizeof(pixel);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls enqueue work and
 *     return without waiting unless documented otherwise -- one documented wait: a detect / analyze call first waits for
 *     the 48-byte counter block of the handle's PREVIOUS detect / analyze call (it selects between code paths with
 *     identical results), i.e. the host runs at most one such call ahead per handle; pipeline over several handles;
 *     one handle per device and per host thread; a handle's
 *     detect / analyze calls share its workspace, so they run in the order they were made even when they name
 *     different streams (a call on another stream than the handle's previous call first waits for that call's event);
 *     a call makes the handle's device current while it runs and leaves the caller's current device as it found it;
 *   - nothing here reads or writes host memory behind the caller's back except the documented
 *     host-pointer parameters.
 */
#ifndef LFA_H_
#define LFA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFA_VERSION 1

/* pixel types */
#define LFA_U8  0
#define LFA_U16 1

/* error codes */
#define LFA_OK               0
#define LFA_ERR_ARG         -1   /* bad argument (shape, dtype, null pointer, size beyond the handle's limits) */
#define LFA_ERR_HIP         -2   /* a HIP runtime call failed; message holds hipGetErrorString */
#define LFA_ERR_UNSUPPORTED -3   /* valid in the reference but not implemented here (documented in DESIGN.md) */
#define LFA_ERR_NOMEM       -4

/* which rounding order `rgb @ [0.2125, 0.7154, 0.0721]` has on the host whose numpy is the yardstick
 * (skimage/color/colorconv.py:876-877; the order is a property of the host BLAS, SURVEY.md section 7 hard part 1).
 * x_k = fl(pixel_k * fl(1/255)); c = coefficients. */
#define LFA_GRAY_FMA_102   0     /* fma(x2,c2, fma(x0,c0, x1*c1))   -- numpy 2.2.6 / OpenBLAS, x86-64 (probed) */
#define LFA_GRAY_SEQ       1     /* (x0*c0 + x1*c1) + x2*c2 */
#define LFA_GRAY_FMA_012   2     /* fma(x2,c2, fma(x1,c1, x0*c0)) */
#define LFA_GRAY_FMA_210   3     /* fma(x0,c0, fma(x1,c1, x2*c2)) */
#define LFA_GRAY_PAIR      4     /* x0*c0 + (x1*c1 + x2*c2) */
#define LFA_GRAY_FMA01_ADD 5     /* fma(x1,c1, x0*c0) + x2*c2 */
#define LFA_GRAY_NVARIANTS 6

typedef struct LfaContext* LfaHandle;

/* One accepted strip = one element of the list detect_lateral_flow_tests returns (detection.py:85-93).
 * Coordinates are the reference's [minr, maxr, minc, maxc] AFTER the min_maxr/min_maxc expansion
 * (detection.py:74) and BEFORE numpy's slice clipping, i.e. maxr/maxc may exceed H/W; crop = image[minr:maxr, minc:maxc]. */
typedef struct {
    int32_t minr, maxr, minc, maxc;
    int32_t area;      /* regionprops area of the component (pixel count), detection.py:81 */
    int32_t label;     /* its label id (1-based, scipy.ndimage.label numbering) */
} LfaStrip;

/* A rectangle of rows/cols of image `image` whose row-mean profile is wanted: rows [r0,r1), cols [c0,c1),
 * already clipped to the image.  Equals the window image_analysis takes (detection.py:215-218). */
typedef struct {
    int32_t image;
    int32_t r0, r1, c0, c1;
} LfaRoi;

/* Result of image_analysis / inverse_image_analysis for one ROI (detection.py:241-245, :277-281). */
typedef struct {
    int32_t n_peaks;        /* 0 (empty profile: the reference raises ValueError), 1 or 2 */
    int32_t length;         /* profile length = r1 - r0 */
    int32_t start[2];
    int32_t end[2];
    double  area[2];
} LfaPeaks;

/* Per-image intermediates of the detect path, for parity checks and diagnostics. */
typedef struct {
    /* integer path (2-D uint8 / uint16): everything in integers */
    int32_t xmin, xmax;       /* min / max pixel value */
    int32_t v_thr;            /* threshold_otsu(adjust_gamma(image, 0.4)) -- the reference's `threshold` (detection.py:38) */
    int32_t x_thr;            /* mask == (pixel > x_thr): largest input value whose gamma value is <= v_thr */
    /* colour path ((H,W,3) uint8): gray is float64 */
    double  gmin, gmax;       /* min / max of rgb2gray(image) */
    double  vmin, vmax;       /* pow(gmin,0.4), pow(gmax,0.4) evaluated on the device */
    double  v_thr_f;          /* bin centre chosen by Otsu, from the device's vmin/vmax */
    double  g_thr;            /* mask == (gray > g_thr) */
    int32_t otsu_bin;         /* argmax index into variance12 (thresholding.py:378) */
    int32_t all_equal;        /* 1: every gamma value identical -> threshold = that value, empty mask (thresholding.py:359-362) */
    uint32_t ambiguous;       /* colour path: pixels closer than 2^-40 (relative) to a histogram edge or to the threshold,
                                 i.e. pixels whose side could depend on the last bits of the host's pow(); 0 in practice */
    /* labelling */
    int32_t n_runs;           /* vertical runs that survive the opening */
    int32_t n_labels;         /* K of label() (detection.py:66) */
    int32_t label00;          /* labeled[0,0] -- the reference's `background_color` (detection.py:67) */
    int32_t n_candidates;     /* labels with area >= 100 */
    int32_t n_accepted;       /* after the mean test and check_duplicate (detection.py:81-85) */
    int32_t n_strips;         /* after remove_outliers (detection.py:87); may exceed max_strips -> see status */
    int32_t status;           /* bit0: n_strips/n_accepted exceeded max_strips (outputs truncated);
                                 bit1: run capacity exceeded;
                                 bit2 (informational, not an error): the image was decided from the caller's host-pow
                                 tables (lfa_set_host_pow) */
} LfaImageInfo;

/* ---- lifetime ---------------------------------------------------------------------------------------------------- */

/* Create a context on `device`.  max_h/max_w/max_c bound the images later calls may pass; max_batch bounds n per call
 * (a call beyond them fails with LFA_ERR_ARG).  Allocates here the whole device workspace of calls within those limits
 * that keep line_length >= 10 and max_strips <= 64 (class plane and big-image Otsu buffers included), so such calls
 * allocate nothing.  A call with a smaller line_length or a larger max_strips GROWS the workspace on the spot: the
 * buffers concerned are released and allocated again with hipFree / hipMalloc, and hipFree waits for the whole device --
 * call lfa_reserve first where that matters (a pipeline of several handles, a latency-bound loop).
 * Builds the default gamma LUTs with the C library's pow().
 * Replaces: module import + the three globals of detection.py:25-28 (state lives in the handle instead). */
int lfa_create(int device, int max_h, int max_w, int max_c, int max_batch, LfaHandle* out);
/* Size the handle's workspace now for calls with max_strips up to `max_strips` and line_length down to
 * `min_line_length` (at the handle's image and batch limits): after it, no detect / analyze call within those limits
 * allocates or frees.  No counterpart in the reference. */
int lfa_reserve(LfaHandle h, int max_strips, int min_line_length);
/* Device bytes the handle holds right now (workspace + tables); a test hook: unchanged across a call = that call did not
 * allocate.  No counterpart in the reference. */
int64_t lfa_workspace_bytes(LfaHandle h);
int lfa_destroy(LfaHandle h);
const char* lfa_last_error(void);
int lfa_version(void);

/* Gamma look-up tables of skimage.exposure.adjust_gamma(image, 0.4) (exposure.py:611-617 for uint8, :674-679 for
 * uint16).  `lut` is HOST memory: 256 uint8 or 65536 uint16 values, must be non-decreasing.  The Python binding passes
 * tables built with the host's numpy so that they carry that numpy's pow() bits. */
int lfa_set_gamma_lut(LfaHandle h, int dtype, const void* lut, int n);

/* Select the rounding order of rgb2gray's matmul (LFA_GRAY_*), colorconv.py:876-877.  numpy evaluates `rgb @ coeffs` row by
 * row of the image -- a BLAS matrix-vector product, and a dot product when the image is one pixel wide -- so the order
 * the reference gets can depend on the image width: the Python binding probes the host's numpy per width
 * (_capi.detect_gray_variant) and calls this before a detect / analyze call whose images need another order.  The
 * value is read when a call is made; calls already enqueued keep theirs. */
int lfa_set_gray_variant(LfaHandle h, int variant);

/* Colour / float64 images whose LfaImageInfo.ambiguous came back non-zero: decide them again with the HOST's pow().
 * The reference bins pow(gray, 0.4) with its host's libm / numpy (skimage/exposure/exposure.py:679,
 * numpy/lib/histograms.py:800-859, skimage/filters/thresholding.py:364-381); the device evaluates the 257 histogram edges and
 * the threshold with ITS pow() and counts in `ambiguous` the pixels within 2^-40 (relative) of one of them -- the only
 * pixels another pow() could put on the other side.  For such an image the caller builds one LfaHostPow from the image's
 * gmin / gmax with the host's pow (the Python binding: _capi.host_pow_tables, numpy) and runs the image again:
 *   tables: DEVICE memory, n records; they are used by the handle's NEXT detect / analyze call (plain or static), which must
 *   pass exactly n images (else LFA_ERR_ARG), and by that call only; the memory must stay valid until that call's kernels
 *   have run.  NULL (with any n) withdraws tables not yet used.
 * In that call the device evaluates no pow() at all: vmin / vmax, the gray-space edges and the gray-space threshold of
 * every Otsu bin come from the record, everything else (min/max, histogram, Otsu scan, mask, ...) runs as always, and
 * LfaImageInfo.status has bit2 set.  2-D integer images ignore the tables (their path has no pow()). */
typedef struct {
    double vmin, vmax;        /* pow(gmin, 0.4), pow(gmax, 0.4) */
    double edges_g[257];      /* edges_g[i] = smallest double g in [gmin, gmax] with pow(g, 0.4) >= linspace(vmin, vmax, 257)[i];
                                 +inf where no such g exists; vmin == vmax: edges_g[0] = gmin, the rest +inf */
    double thr_g[256];        /* thr_g[b] = smallest double g in [gmin, gmax] with pow(g, 0.4) > (edge[b] + edge[b+1]) / 2 in
                                 gamma space, +inf where none: the mask of an image whose Otsu bin is b is gray >= thr_g[b] */
} LfaHostPow;
int lfa_set_host_pow(LfaHandle h, const LfaHostPow* tables, int n);

/* Test hook (no counterpart in the reference).  bit0: colour images take the float64 compare path even when the exact
 * integer form of the thresholds is available (that path otherwise runs only for "critical" thresholds).
 * bit1: images do not use the per-pixel class plane between the histogram and the mask pass (the mask pass then
 * re-reads the image, as it does for unaligned images).  Colour images and, with the one-pass histogram, uint16 frames
 * have such a plane.
 * bit2: the class plane is written and used however many pixels share the threshold's class (by default an image uses
 * it only below 1/2048 of its pixels, because each of them sends the mask pass back to the image, and a handle stops
 * writing the plane while fewer than half of the images of its last completed call could use it; uint16 frames also
 * need a completed call before, whose mean threshold centres the classes -- with bit2 they start at value 0 without;
 * a frame whose threshold is the last value of its class needs no compare at all and always uses the plane).
 * bit3: uint16 images always take the min/max pass before the histogram pass; bit4: never (by default a handle drops the
 * min/max pass once its last completed uint16 call held no value of 16384 or more).
 * bit5: the LDS profile slots of lfa_analyze_batch are sized for the worst case (by default for the longest analysis window
 * of the handle's last completed call; longer windows take the global-memory path).
 * bit6 (64): lfa_detect_lines_batch keeps the float64 Gaussian image (8 more bytes per pixel) for
 * lfa_lines_debug_fetch(LFA_LDBG_GAUSS); bit7 (128): its labelling looks every pixel's class up in the class table (by
 * default the class is computed from the level where the table is "0, then one class per level, then constant").
 * bit10 (1024): colour uint8 calls take the equal-channel ("grey") mode whatever the handle's last call held; bit11 (2048):
 * never.  By default a handle takes it once at least half of the colour images of its last completed call had three equal
 * channels in every pixel -- what cv2.imread hands the reference for a grey file (detection.py:454): one pass then takes the
 * byte histogram and the min/max and histogram passes skip the image; an image with a single unequal pixel takes the
 * ordinary passes.  bit12 (4096): the equal-channel pass writes no byte plane (the mask pass then compares the three
 * channels, as it does for any image without a class plane).  Results are identical either way.
 * bit13 (8192): colour uint8 calls never take the predicted-threshold mode; bit15 (32768): always, with a window that holds
 * every threshold, so that every pixel is decided by the exact compare of the fix-up kernel; bit14 (16384): the window is
 * the bare range of the last call's thresholds, without a margin.  By default a handle that has completed a colour uint8
 * call expects the next call's integer mask thresholds inside the range the last call's had (widened by 16384 of G's
 * 2 550 000 steps, i.e. a grey level and a half, either way): the histogram pass then writes one bit per pixel -- at or above the window
 * -- instead of the 4-bit class, and lists the 16-pixel groups holding a pixel inside the window; an image whose threshold
 * falls inside the window, with at most 1/16 of its groups listed, takes its mask from those bits plus an exact compare of
 * the listed pixels, any other image is compared on the image itself; a handle leaves the mode for eight calls when fewer
 * than half of a call's images could use their bits.  uint16 frames on the one-pass histogram take the same mode with the
 * pixel value in place of G (T = x_thr + 1, the mask is `pixel >= T`): the window is the tightest cluster of the last call's
 * T values -- the largest set within 32 values of each other, among the frames with few pixels near their threshold --
 * widened by max(4, its range) either way, and the handle tries it only after a call has reported that the window would
 * have held at least half of its frames (T inside it, at most 1/2048 of the pixels inside it); bit13 / bit14 / bit15 act
 * on it as on the colour mode.  Equal-channel images in grey mode (bit10 / the handle's own finding): the window is the
 * last call's range of thresholds as it is, in grey levels -- no margin: a series with one byte threshold has no undecided
 * pixel, any other falls back per image.  bit16 (65536): the bits of an image they decide are always turned into vertical words
 * by the mask pass (default for widths that are multiples of 64: the run extraction turns them itself).  Results are identical either way.
 * bit8 (256) / bit9 (512): k_edges_rgb moves every interior gray-space histogram edge it computed one double up / down --
 * what a pow() whose last bit differs would do -- so that the tests can show an image with a pixel exactly on an edge
 * going wrong without lfa_set_host_pow and right with it (tables from lfa_set_host_pow are never touched). */
/* (The environment variable LFA_DEBUG_FLAGS, read by lfa_create, gives a new handle its initial flags.) */
int lfa_set_debug_flags(LfaHandle h, int flags);

/* ---- detect: detect_lateral_flow_tests(image, line_length) -- detection.py:31-94 ----------------------------------- */

/* images: n images of H x W x C pixels of `dtype` (C==1: 2-D input; C==3: colour, uint8 only).
 * min_maxr/min_maxc: the module globals of detection.py:26 the GUI sets at :457-460.
 * out_strips: device, n * max_strips records; image i's strips start at i*max_strips, sorted by (minc, maxc) (detection.py:89).
 * out_counts: device, n int32 = number of strips of image i (<= max_strips unless info.status bit0).
 * out_info:   device, n LfaImageInfo, or NULL.
 * The canny / Hough / rotate stages (detection.py:51-57, :84) do not influence the return value and are not run. */
int lfa_detect_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                     int min_maxr, int min_maxc, int line_length,
                     LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                     void* stream);

/* ---- profile + peaks: image_analysis / inverse_image_analysis / analyze_peaks -- detection.py:205-316 ------------- */

/* For each ROI: profile[k] = mean over cols [c0,c1) and channels of row r0+k (np.nanmean of an integer array,
 * detection.py:219), then the one- or two-peak logic (detection.py:221-245; `inverse` != 0: :257-281).
 * `range` is the reference's parameter of the same name (15 for image_analysis, 1 for inverse_image_analysis).
 * rois: device, n_rois records.  out_profiles: device, n_rois * profile_stride float64; ROI j's profile starts at
 * j*profile_stride; every ROI must have r1-r0 <= profile_stride (else LFA_ERR_ARG is reported through out_peaks[j].n_peaks = -1).
 * out_peaks: device, n_rois records. */
int lfa_profile_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                      const LfaRoi* rois, int n_rois, int inverse, int range,
                      double* out_profiles, int profile_stride, LfaPeaks* out_peaks, void* stream);

/* The same after the per-region point operations MyGUI.analyze_plot applies first (detection.py:535-539):
 *   region = adjust_gamma(region, gamma_level); rescale_intensity(region, in_range=(low, high)); util.invert(region)
 * (skimage/exposure/exposure.py:620-681, :485-600; skimage/util/_invert.py:62-74).  On uint8 / uint16 pixels each of them,
 * and any chain of them, is a value -> value map of the same dtype, i.e. a table; the caller builds the table with the
 * host's numpy (bit-exact by construction) and every pixel of a ROI goes through it before the row mean.
 * luts: device, n_luts tables of 256 uint8 (LFA_U8) or 65536 uint16 (LFA_U16) entries, back to back.
 * roi_lut: device int32[n_rois], the table of each ROI, or NULL = table 0 for all; an index outside [0, n_luts) is
 * reported through out_peaks[j].n_peaks = -1 like any other bad ROI. */
int lfa_profile_lut_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                          const LfaRoi* rois, int n_rois, const void* luts, int n_luts, const int32_t* roi_lut,
                          int inverse, int range, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                          void* stream);

/* image_analysis / inverse_image_analysis on regions of the other element types np.nanmean takes (detection.py:205-281 with
 * `region` an int8 / int16 / int32 / uint32 or float64 array; bool regions are uint8 0 / 1 and go through lfa_profile_batch).
 * float64: NaN pixels count as 0 and are left out of the row's divisor (np.nanmean), the row sum is numpy's pairwise sum.
 * regions: device, n arrays of H*W*C elements; everything else as lfa_profile_batch.  Single regions handed over by a
 * caller, not a batched hot path: one wave per ROI. */
#define LFA_E_I8  2
#define LFA_E_I16 3
#define LFA_E_I32 4
#define LFA_E_U32 5
#define LFA_E_F64 6
/* float64 IMAGES (2-D or colour, non-negative values) are taken by lfa_detect_batch only, as pixel dtype LFA_F64: gray is the
 * pixel itself (2-D) or rgb @ coeffs without the 1/255 scaling (colour), everything downstream is the colour path's float64
 * arithmetic (skimage: img_as_float leaves floats alone, adjust_gamma uses scale 1).  When pixel (0,0) belongs to a component
 * the reference compares float box means: those sums follow numpy's order (2-D: pairwise over C-order buffers of 8192 elements;
 * colour: one running sum per channel), k_regions.hip. */
#define LFA_F64 LFA_E_F64
int lfa_profile_any_batch(LfaHandle h, const void* regions, int elem_type, int n, int H, int W, int C, const LfaRoi* rois,
                          int n_rois, int inverse, int range, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                          void* stream);

/* analyze_peaks(profile, inverse, window_size) on caller-supplied float64 profiles (detection.py:283-316).
 * profiles: device, n_profiles * profile_stride float64; lengths: device int32[n_profiles].
 * out: device LfaPeaks[n_profiles] with n_peaks = 1 (start[0], end[0], area[0]); when `two_peaks` != 0 the full
 * image_analysis peak logic (detection.py:221-245 / :257-281) runs instead, with `range`. */
int lfa_peaks_batch(LfaHandle h, const double* profiles, const int32_t* lengths, int n_profiles, int profile_stride,
                    int inverse, int window_size, int two_peaks, int range, LfaPeaks* out_peaks, void* stream);

/* ---- the whole job: detect, then image_analysis(strip, selected=False) of every strip ----------------------------- */

/* = for image in batch: for strip in detect_lateral_flow_tests(image): image_analysis(strip, False)
 *   (GUI sequence detection.py:462 then :543 with default gamma/exposure), with no host round trip.
 * out_rois (device, n*max_strips, may be NULL) receives the window of each strip (detection.py:216);
 * out_profiles: device, n*max_strips*profile_stride float64; out_peaks: device, n*max_strips.
 * Slots beyond out_counts[i] are left untouched. */
int lfa_analyze_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                      int min_maxr, int min_maxc, int line_length,
                      LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                      LfaRoi* out_rois, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                      void* stream);

/* lfa_analyze_batch for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results, but the per-call workspace lies in a
 * file-scope device array of LFA_STATIC_SCRATCH_BYTES inside the library (each GPU the library runs on holds its own copy
 * until the process ends; its address is looked up once per device in lfa_create), and the call makes no hipMalloc /
 * hipFree / allocating library call, does no kernel timing and feeds nothing back to the handle's path choices (it always
 * takes the plane-less, min/max-first paths, whose results are identical to the others').  It consists of kernel launches
 * and hipMemsetAsync on `stream` and nothing else -- no host wait (not even for the previous call's counters), no event
 * recorded or awaited: lfa_stream_wait_passes / lfa_chain_passes do not see static calls, and the call can be captured
 * into a HIP graph and replayed (tests/test_zz_no_alloc.py does that under hipStreamCaptureModeGlobal).
 *   - a batch whose workspace does not fit fails with LFA_ERR_ARG and a message naming the bytes needed; at the default
 *     line_length 10 and max_strips 64 the array holds e.g. 4 images of 1536x2048x3, 3 frames of 2048x2048 uint16 or one
 *     image of 3072x4096x3;
 *     lfa_static_workspace_bytes() returns the capacity;
 *   - every call on a GPU uses that GPU's one array: calls that overlap in time on one GPU (two streams, two handles, two
 *     threads) RACE -- serialise them; the array keeps the last call's contents, each call clears what it needs zeroed;
 *   - lfa_debug_fetch after it reads the static workspace (valid until the next static call on that GPU). */
#define LFA_STATIC_SCRATCH_BYTES (64u << 20)
int lfa_analyze_batch_static(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                             int min_maxr, int min_maxc, int line_length,
                             LfaStrip* out_strips, int max_strips, int32_t* out_counts, LfaImageInfo* out_info,
                             LfaRoi* out_rois, double* out_profiles, int profile_stride, LfaPeaks* out_peaks,
                             void* stream);
int64_t lfa_static_workspace_bytes(void);

/* ---- gaussian: skimage.filters.gaussian(image, sigma) -- skimage/_shared/filters.py:35-166, caller detection.py:166 ---- */

/* Separable Gaussian, mode='nearest', truncate=4.0, on img_as_float(image) (uint8: x*(1/255), uint16: x*(1/65535)),
 * axis 0 then axis 1, channels (C==3) filtered independently.  out: device, n*H*W*C float64 (out_f32 == 0) or float32.
 * host_weights: HOST memory, 2*radius+1 kernel taps as scipy.ndimage._filters._gaussian_kernel1d builds them, or NULL
 * to have them computed here from sigma with the C library's exp() (radius = int(4*sigma + 0.5), at most 8). */
int lfa_gaussian_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W, int C,
                       double sigma, const double* host_weights, int radius, void* out, int out_f32, void* stream);

/* ---- detect_lines: LFA-Image_Analysis/detection.py:156-201 (defined by the reference, never called) ------ */

typedef struct {
    int32_t minr, maxr, minc, maxc;   /* [minr, maxr, minc, maxc] as detection.py:200 appends them (minc = 0, maxc = W always) */
} LfaLineBox;

typedef struct {
    int32_t umin, umax;               /* extremes of the uint16 image img_as_uint makes inside equalize_adapthist (_adapthist.py:73) */
    int32_t cmin, cmax;               /* extremes of _clahe's uint16 result (_adapthist.py:87) */
    double p80, p99;                  /* np.percentile(enhanced_image, (80, 99)), detection.py:179 */
    int32_t n_labels;                 /* labeled_image.max(), detection.py:184 */
    int32_t n_boxes;                  /* regions that pass the filter of detection.py:191-199 (may exceed max_boxes) */
    int32_t status;                   /* bit 0: more boxes than max_boxes; bit 1: a value above 1 reached img_as_uint (the
                                         reference raises "Images of type float must be between -1 and 1.") */
    int32_t pad;
} LfaLinesInfo;

/* = for image in batch: detect_lines(image) for 2-D uint8 / uint16 images (C == 1; the reference's function fails on
 *   colour input at its `minr, minc, maxr, maxc = region.bbox`).  Stages: gaussian(sigma 1) -> threshold_local(block 3,
 *   'gaussian') -> image * (threshold > 0) -> adjust_gamma(0.5) -> equalize_adapthist(clip_limit 0.05) -> percentile
 *   rescale (80, 99) -> label (skimage labels an np.intp copy of a non-boolean image: of this float image in [0, 1] the
 *   pixels equal to 1.0, 8-connected) -> bounding boxes -> area / shape filter, all on the device.
 * w_sigma1: HOST, 9 taps of scipy's Gaussian kernel for sigma 1; w_local: HOST, 3 taps for sigma (3-1)/6; NULL = computed
 *   here with the C library's exp().
 * out_boxes: device, n*max_boxes, in the order the reference appends them; out_counts: device, n; out_info: device, n, or NULL.
 *   Cost of max_boxes: the boxes that pass the filter (at most max_boxes are kept) are ranked by one 256-thread block per image
 *   with an O(m^2) compare loop -- nothing for the handful a real frame yields, 4e9 compares at the limit of 65536; the dense
 *   ROI path likewise indexes its ROI array with one block (k_roi_index, linear in n_rois).  Keep max_boxes near what the
 *   frames can yield (the Python binding starts at 256 and grows on overflow).
 * The workspace of this path (about 26 bytes per pixel) is allocated by the first call that needs it and grows with later
 * ones: unlike detect/analyze, this call MAY allocate (and then waits for the device). */
int lfa_detect_lines_batch(LfaHandle h, const void* images, int dtype, int n, int H, int W,
                           const double* w_sigma1, const double* w_local,
                           LfaLineBox* out_boxes, int max_boxes, int32_t* out_counts, LfaLinesInfo* out_info, void* stream);

#define LFA_LDBG_GAUSS   0   /* float64 H*W : sk.filters.gaussian(image)                        (detection.py:166); kept only when
                                 lfa_set_debug_flags(h, 64) was in force at the call: the product path never writes it */
#define LFA_LDBG_U16     1   /* uint16  H*W : img_as_uint(adjust_gamma(...)) inside equalize_adapthist (_adapthist.py:73) */
#define LFA_LDBG_LEVELS  2   /* uint16  H*W : _clahe's result                                   (_adapthist.py:87) */
#define LFA_LDBG_ROOTS   3   /* int32   H*W : index of the first pixel of the pixel's component, -1 = background (detection.py:184) */
#define LFA_LDBG_LVAL    4   /* float64 16384 : value of rescale_intensity(enhanced_image, ...) per level (detection.py:180) */
/* Copy intermediate `what` of image `image_index` of the handle's last lfa_detect_lines_batch call to `out` (device or host). */
int lfa_lines_debug_fetch(LfaHandle h, int image_index, int what, void* out, int64_t out_bytes, void* stream);

/* ---- diagnostics for parity tests (valid until the next detect/analyze call on this handle) ----------------------- */

#define LFA_DBG_MASK    0   /* uint8 H*W : image_copy > threshold            (detection.py:39,47) */
#define LFA_DBG_OPENED  1   /* uint8 H*W : binary_dilation(binary_erosion())  (detection.py:60-63) */
#define LFA_DBG_LABELS  2   /* int32 H*W : label(dilated)                     (detection.py:66) */
#define LFA_DBG_TABLE   3   /* int32 K*5 : minr,minc,maxr,maxc,area per label (regionprops, detection.py:71-73) */
#define LFA_DBG_HIST    4   /* uint32 nb : histogram counts (nb = 256 colour / uint8, xmax-xmin+1 uint16; input domain) */
#define LFA_DBG_PLANE   5   /* int32 1   : 1 if the mask of this image was decided from the class plane, 2 if from the
                                            predicted-threshold bit plane, else 0                                              */
#define LFA_DBG_GREY    7   /* int32 1   : 1 if this colour image went through the equal-channel path (byte histogram, no min/max
                                              and no histogram pass), else 0 */
#define LFA_DBG_ONEPASS 6   /* int32 1   : 1 if the last call took the one-pass uint16 histogram (no min/max pass), else 0 */

/* Rasterise intermediate `what` of image `image_index` of the LAST detect/analyze call into `out` (device). */
int lfa_debug_fetch(LfaHandle h, int image_index, int what, void* out, int64_t out_bytes, void* stream);

/* ---- per-kernel timing for bench.py (no counterpart in the reference) ---------------------------------------------- */

/* kernel_filter: "*" brackets every kernel launch made through this handle with a pair of hipEvents on the launch
 * stream, "<kernel name>" only that kernel, "<kernel name>/N" only every N-th launch of it (the event records sit between
 * the kernels of a stream: bracketing every launch of the headline's dominant kernel cost 2 % of the step), NULL or ""
 * switches timing off.  lfa_timing_read waits for the device,
 * returns the summed milliseconds and launch counts per kernel name since the last read (names: max_entries slots of
 * 32 bytes, HOST memory) and forgets them. */
int lfa_timing_enable(LfaHandle h, const char* kernel_filter);
int lfa_timing_read(LfaHandle h, int max_entries, char* names, double* total_ms, int32_t* launches, int32_t* n_out);

/* Chunk pipelining over two handles (no counterpart in the reference).  Every detect/analyze call records an event on
 * its stream right after its histogram pass, i.e. when two of its three whole-image passes are done and what is left
 * is the (short) mask pass and the latency-bound tail (labelling, regions, profiles, peaks).  lfa_stream_wait_passes
 * makes `stream` wait for that point of `producer`'s most recent call (no-op before its first call), so a caller that
 * feeds consecutive batches to two handles on two streams can start batch k+1's passes when batch k stops needing
 * the memory bandwidth, and batch k's tail runs in their shadow. */
int lfa_stream_wait_passes(LfaHandle producer, void* stream);

/* The same ordering as a standing rule (no counterpart in the reference): from now on the whole-image passes of every
 * detect/analyze call on `consumer` start when the passes (through the histogram) of `producer`'s most recent call are
 * done -- the wait is enqueued inside the call, after its set-up kernels and right before its first image pass, so
 * nothing but the passes themselves is held back.  With handles h0 .. h(k-1) taking consecutive batches in turn and
 * h(i) chained after h(i-1) (h0 after h(k-1)), at most one batch is in its min/max or histogram pass at any time: those
 * passes run at their stand-alone speed, one behind the other, and the tails of the batches before run beside them.
 * (On a chained handle the profile kernel of lfa_analyze_batch asks for 60 KiB of LDS it does not use, so that it and the
 * peak search behind it run after the histogram pass of the next batch instead of beside it; results do not depend on it.)
 * producer NULL ends the rule.  Both handles must be on the same device; end the rule before destroying `producer`. */
int lfa_chain_passes(LfaHandle consumer, LfaHandle producer);

/* Block until everything enqueued on `stream` through this handle has finished. */
int lfa_sync(LfaHandle h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LFA_H_ */
