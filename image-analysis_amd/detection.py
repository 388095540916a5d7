"""Drop-in for the hot path of the reference's LFA-Image_Analysis/detection.py (functions at :31-316).

Same names, signatures, defaults, module globals, return types and exceptions; the GUI class (:319-789) is not
part of this module and importing it opens no window.  Every numeric step runs in liblfa_hip.so on the MI355X
(through image-analysis_amd/_capi.py); this module only marshals arrays and keeps the reference's host-side list
logic.  There is no CPU fallback: without the HIP library or without a GPU the calls raise.

    import image_analysis_amd.detection as detection
    detection.min_maxc = int(width / 40); detection.min_maxr = int(detection.min_maxc * 20)   # detection.py:457-460
    crops = detection.detect_lateral_flow_tests(image)                                         # detection.py:462
    profile, peaks = detection.image_analysis(crops[0], False)                                 # detection.py:543
"""
from __future__ import annotations

import numpy as np

from . import _capi

# module globals the reference's callers assign with `global` (detection.py:25-28)
image_obj = None
min_maxr, min_maxc = 0, 0
angle_threshold = 0
file_path = ""

_DEVICE = 0


def set_device(index: int) -> None:
    """Select the GPU the module-level functions run on (new; the reference has no devices)."""
    global _DEVICE
    _DEVICE = int(index)


def _engine(H, W):
    return _capi.shared_engine(_DEVICE, H, W)


def _check_pixels(image: np.ndarray, allow_float64=False) -> np.ndarray:
    if image.dtype == np.uint8 or image.dtype == np.uint16:
        return image
    if image.dtype.kind in "if" and np.any(image < 0):
        # skimage.exposure._assert_non_negative (exposure.py:603-608)
        raise ValueError("Image Correction methods work correctly only on images with non-negative values. "
                         "Use skimage.exposure.rescale_intensity.")
    if allow_float64 and image.dtype == np.float64:
        return image
    raise NotImplementedError(f"pixel dtype {image.dtype}: the HIP path takes uint8 and uint16 images "
                              "(what cv2.imread / tifffile hand the reference, detection.py:454)")


def detect_lateral_flow_tests(image, line_length=10):
    """detection.py:31-94.  Returns views `image[minr:maxr, minc:maxc]` sorted by (minc, maxc)."""
    image = np.asanyarray(image)
    if image.ndim == 2:
        pass
    elif image.ndim == 3:
        if image.shape[-1] != 3:
            # skimage.color.colorconv._prepare_colorarray (colorconv.py:137-140)
            raise ValueError(f"the input array must have size 3 along `channel_axis`, got {image.shape}")
    else:
        raise ValueError("Invalid image dimensions. Only 2D and 3D images are supported.")
    if image.size == 0:
        raise ValueError("empty image")
    _check_pixels(image, allow_float64=True)
    H, W = image.shape[0], image.shape[1]
    eng = _engine(H, W)
    strips, _info = eng.detect(np.ascontiguousarray(image)[None], int(min_maxr), int(min_maxc), int(line_length))
    return [image[int(s["minr"]):int(s["maxr"]), int(s["minc"]):int(s["maxc"])] for s in strips[0]]


def straighten_region(copy_image, region, angles):
    """detection.py:97-115 rotates a patch of a private copy that is never read again (:69, :86): no effect on any
    result, so nothing is computed here."""
    return None


def check_duplicate(minr, minc, maxr, maxc, lst) -> bool:
    """True when the box rows [minr, maxr] x cols [minc, maxc] touches none of the regions accepted so far, each of them
    grown to at least min_maxr x min_maxc; bounds are inclusive (detection.py:118-127).  Host list logic: the batched
    path runs the same test on the device (k_finalize)."""
    def grown(reg):
        r0, r1, c0, c1 = reg[0], reg[1], reg[2], reg[3]
        return r0, max(r1, r0 + min_maxr), c0, max(c1, c0 + min_maxc)

    return not any(minr <= r1 and maxr >= r0 and minc <= c1 and maxc >= c0 for r0, r1, c0, c1 in map(grown, lst))


def remove_outliers(lst):
    """detection.py:138-153."""
    widths = [max(r[3], r[2] + min_maxc) - r[2] for r in lst]
    if not widths:
        return []
    narrowest = min(widths)
    return [r for r, w in zip(lst, widths) if w <= narrowest * 2]


def detect_lines(image) -> list:
    """detection.py:156-201 -> [[minr, maxr, minc, maxc], ...] (the reference defines this function and never calls it).

    Gaussian, local threshold, gamma 0.5, CLAHE, percentile rescale, labelling (of the image cast to intp, as skimage's
    label does: the pixels at 1.0) and the box filter all run
    on the device (`lfa_detect_lines_batch`); the matplotlib figure of :162-164 is not drawn.  Like the reference it needs
    a 2-D image: on colour input the reference fails at `minr, minc, maxr, maxc = region.bbox` (:189)."""
    image = np.asanyarray(image)
    if image.ndim != 2:
        raise ValueError("too many values to unpack (expected 4)" if image.ndim > 2 else "detect_lines needs a 2-D image")
    if image.size == 0:
        raise ValueError("empty image")
    _check_pixels(image)
    boxes, info = _engine(image.shape[0], image.shape[1]).detect_lines(image[None])
    if info["status"][0] & 2:
        # skimage.util.dtype._convert inside equalize_adapthist (_adapthist.py:73)
        raise ValueError("Images of type float must be between -1 and 1.")
    return [[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in boxes[0]]


def gaussian(image, sigma=1):
    """skimage.filters.gaussian(image, sigma) as detect_lines calls it (detection.py:166): float64 result."""
    image = np.asanyarray(image)
    _check_pixels(image)
    if image.ndim == 3 and image.shape[-1] != 3:
        raise NotImplementedError("only (H,W) and (H,W,3) images")
    eng = _engine(image.shape[0], image.shape[1])
    return eng.gaussian(image[None], sigma=float(sigma))[0]


def _as_2d_elements(region: np.ndarray):
    """(h,w) or (h,w,c) -> contiguous (h, w*c) plus c: the row mean over columns and channels is the mean over w*c elements."""
    if region.ndim == 3:
        h, w, c = region.shape
        return np.ascontiguousarray(region).reshape(h, w * c), c
    if region.ndim == 2:
        return np.ascontiguousarray(region), 1
    raise ValueError(f"not enough values to unpack (expected 2, got {region.ndim})")


def _peaks_to_list(pk):
    out = []
    for i in range(int(pk["n_peaks"])):
        out.append((int(pk["start"][i]), int(pk["end"][i]), np.float64(pk["area"][i])))
    return out


_OTHER_REGION_DTYPES = ("int8", "int16", "int32", "uint32", "float64")       # what lfa_profile_any_batch takes


def _analysis(region, selected, rng, inverse, lut=None):
    region = np.asanyarray(region)
    other = False
    if region.dtype == np.bool_:
        region = region.view(np.uint8)                        # np.mean of a bool array is the mean of its 0 / 1 bytes
    elif region.dtype.name in _OTHER_REGION_DTYPES and lut is None:
        other = True
    elif region.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError(f"region dtype {region.dtype}: the HIP path takes bool, uint8, uint16, int8, int16, int32, uint32 "
                                  "and float64 regions")
    flat, c = _as_2d_elements(region)
    h, we = flat.shape
    w = we // c
    if selected:
        r0, r1, c0, c1 = 0, h, 0, w
    else:
        r0, r1, c0, c1 = int(h * .25), int(h * .6), int(w * .3), int(w * .7)
    if r1 - r0 <= 0:
        raise ValueError("attempt to get argmax of an empty sequence" if not inverse else
                         "attempt to get argmin of an empty sequence")
    if h == 0 or we == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    roi = np.zeros(1, dtype=_capi.ROI_DT)
    roi["image"], roi["r0"], roi["r1"], roi["c0"], roi["c1"] = 0, r0, r1, c0 * c, c1 * c
    eng = _engine(h, we)
    if other:
        profiles, peaks = eng.profiles_any(flat[None], roi, inverse=inverse, rng=int(rng))
    else:
        profiles, peaks = eng.profiles(flat[None], roi, inverse=inverse, rng=int(rng), luts=lut)
    return [profiles[0], _peaks_to_list(peaks[0])]


def image_analysis(region, selected, range=15):
    """detection.py:205-245 -> [profile float64 (h',), [(start, end, area)[, (start2, end2, area2)]]]."""
    return _analysis(region, selected, range, inverse=False)


def inverse_image_analysis(region, selected, range=1):
    """detection.py:247-281 (valleys).  Like the reference it needs a 3-D region (:249)."""
    region = np.asanyarray(region)
    if region.ndim != 3:
        raise ValueError(f"not enough values to unpack (expected 3, got {region.ndim})")
    return _analysis(region, selected, range, inverse=True)


def analyze_plot(region, selected, gamma_level=1, low_exposure=None, high_exposure=None, invert=False):
    """The data path of MyGUI.analyze_plot (detection.py:535-543): adjust_gamma(region, gamma_level), then
    rescale_intensity(region, in_range=(low_exposure, high_exposure)) when either is set, then either
    inverse_image_analysis(util.invert(region), selected) or image_analysis(region, selected).

    The three point operations become one table (`_capi.point_op_lut`, built with the host's numpy) that the profile
    kernel applies to every pixel it reads; the pre-processed region is never materialised."""
    region = np.asanyarray(region)
    if region.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError("analyze_plot pre-processing is built for uint8 and uint16 regions")
    lut = _capi.point_op_lut(region.dtype, gamma_level, low_exposure, high_exposure, bool(invert))
    if invert:
        if region.ndim != 3:
            raise ValueError(f"not enough values to unpack (expected 3, got {region.ndim})")
        return _analysis(region, selected, 1, inverse=True, lut=lut)
    return _analysis(region, selected, 15, inverse=False, lut=lut)


def analyze_peaks(profile, inverse=False, window_size=25):
    """detection.py:283-316 -> (start_index, end_index, area)."""
    profile = np.ascontiguousarray(profile, dtype=np.float64)
    if profile.ndim != 1:
        raise ValueError("profile must be one-dimensional")
    if profile.size == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    pk = _capi.shared_engine(_DEVICE).peaks([profile], inverse=inverse, window_size=int(window_size))[0]
    return int(pk["start"][0]), int(pk["end"][0]), np.float64(pk["area"][0])


# ---- batch entry points (new: SURVEY.md section 3.4; the reference handles one image and one strip at a time) --------
def analyze_batch(images, line_length=10, max_batch=16, device=None, max_strips=64, devices=None):
    """detect + image_analysis(strip, False) for every strip of every image of a (n,H,W[,3]) batch.

    Uses the module globals min_maxr / min_maxc exactly like detect_lateral_flow_tests.  Returns one entry per image:
    a list of (crop_view, profile, peaks) in the order detect_lateral_flow_tests returns the crops.  max_strips is the
    number of result slots per image to start with; an image with more accepted strips re-runs its batch with 4x the
    slots (up to 1024), so the value only matters for speed and memory.

    devices=[d0, d1, ...]: the batch is cut into contiguous blocks, one per entry (`shard.shard_range`; image i goes to
    entry i*len(devices)//n), and every block runs on its device with its own handle, its own pair of HIP streams and its
    own host thread (`frames.analyze_stream`: pinned, double-buffered uploads, batches of max_batch); the per-image results
    are joined on the host in image order.  Images are independent, so no device talks to another.  The
    same device may be named more than once (two handles, two stream pairs on one GPU)."""
    images = np.asanyarray(images)
    if images.ndim not in (3, 4) or (images.ndim == 4 and images.shape[-1] != 3):
        raise ValueError("expected a batch (n,H,W) or (n,H,W,3)")
    if images.dtype == np.float64:
        # float64 images: the batched C call takes integer pixels; detect and the profiles run per image / per crop
        # (lfa_detect_batch with LFA_F64, lfa_profile_any_batch) -- same results, no batching across images
        _check_pixels(images, allow_float64=True)
        out = []
        for img in images:
            crops = detect_lateral_flow_tests(img, line_length)
            out.append([(c, *image_analysis(c, False)) for c in crops])
        return out
    _check_pixels(images)
    if devices is not None:
        res = _analyze_sharded(images, [int(d) for d in devices], int(line_length), int(max_batch), int(max_strips))
    else:
        dev = _DEVICE if device is None else int(device)
        eng = _capi.Engine(device=dev, max_h=images.shape[1], max_w=images.shape[2], max_c=3, max_batch=max_batch)
        try:
            res, _info = eng.analyze(images, int(min_maxr), int(min_maxc), int(line_length), max_strips=int(max_strips))
        finally:
            eng.close()
    out = []
    for i, (strips, profiles, peaks) in enumerate(res):
        img = images[i]
        out.append([(img[int(s["minr"]):int(s["maxr"]), int(s["minc"]):int(s["maxc"])], profiles[k], _peaks_to_list(peaks[k]))
                    for k, s in enumerate(strips)])
    return out


def _analyze_sharded(images, devices, line_length, max_batch, max_strips):
    """One host thread per entry of `devices`, each streaming its contiguous block of the batch through its own handle."""
    from concurrent.futures import ThreadPoolExecutor

    from . import frames, shard
    if not devices:
        raise ValueError("devices must name at least one GPU")
    n, world = len(images), len(devices)
    mm_r, mm_c = int(min_maxr), int(min_maxc)
    pack = max(1, 16 // world)

    def run(k):
        lo, hi = shard.shard_range(n, k, world)
        got = list(frames.analyze_stream((images[i] for i in range(lo, hi)), batch=max_batch, device=devices[k],
                                         line_length=line_length, max_strips=max_strips, min_maxr=mm_r, min_maxc=mm_c,
                                         pack_threads=pack))
        assert [g[0] for g in got] == list(range(hi - lo))
        return [g[1:] for g in got]

    with ThreadPoolExecutor(max_workers=world) as ex:
        blocks = list(ex.map(run, range(world)))                   # in shard order = image order
    return [r for blk in blocks for r in blk]
