"""CPU oracle for the LFA hot path (TEST INFRASTRUCTURE ONLY).

This file is a numpy + scipy.ndimage restatement of the arithmetic the reference executes for
strip detection, per-strip / per-ROI profiles and top-2 peaks.  It is the checker for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``image-analysis_amd/`` imports it, and the product path fails loudly without its HIP library.

What it follows (paths relative to /root/reference; SP = image-analysis/lib/python3.10/site-packages):

* detect path          LFA-Image_Analysis/detection.py:31-94, :118-153
* rgb2gray             SP/skimage/color/colorconv.py:131-147, :834-877 ; SP/skimage/util/dtype.py:320-329
* adjust_gamma         SP/skimage/exposure/exposure.py:603-681
* threshold_otsu       SP/skimage/filters/thresholding.py:256-381 ; SP/skimage/exposure/exposure.py:35-83, :271-305
* binary opening       SP/skimage/morphology/binary.py:79-88, :138-145 ; footprints.py:230-231
* label / regionprops  SP/skimage/measure/_label.py:116-118 ; _regionprops.py:399-411, :1321-1333
* profile + peaks      LFA-Image_Analysis/detection.py:205-316
* gaussian             SP/skimage/_shared/filters.py:35-166

skimage itself delegates every output-determining step to numpy / scipy.ndimage, which are installed
here, so this is the reference arithmetic and not an approximation of it.  The canny / Hough / rotate
stages (detection.py:51-57, :84, :97-115) write only to a copy that is never read again and are left out.

Pinning: tests/test_oracle_golden.py checks this file against
  (i)  the known answer the reference's own export holds (images/excel2.xlsx: frame, crop box, profile);
  (ii) vectors produced by executing the reference's numpy-only source lines (detection.py:118-153,
       :205-316) in the build container (tests/golden/make_golden.py; outputs in tests/golden/*.npz).
tests/test_skimage_vectors.py checks it against
  (iii) the whole of detect_lateral_flow_tests (detection.py:31-153) executed over the reference's vendored
       scikit-image 0.20.0 Python sources on 24 small inputs plus the excel2 frame (tests/golden/vendored_skimage.py,
       make_golden.py; tests/golden/ref_skimage_detect.npz): threshold bits, SHA-256 of gray / gamma / mask / opened mask /
       label image, the bbox + area table and the returned boxes;
  (iv) expected values the vendored skimage's own test-suite holds for adjust_gamma, threshold_otsu and rgb2gray.
Still unpinned: the numpy / scipy underneath are this container's 2.2.6 / 1.15.3, not the reference's 1.24.3 / 1.10.1
(whose C extensions exist only as darwin-arm64 binaries), and the file decoders (cv2 / tifffile are absent).
"""
from __future__ import annotations

import warnings

import numpy as np
from scipy import ndimage as ndi

GRAY_COEFFS = (0.2125, 0.7154, 0.0721)   # colorconv.py:876
GAMMA = 0.4                               # detection.py:35,43


# --------------------------------------------------------------------------------------------------
# D1  rgb2gray  (colorconv.py:834-877)
# --------------------------------------------------------------------------------------------------
def rgb2gray(image: np.ndarray) -> np.ndarray:
    """uint8/uint16 (H,W,3) -> float64 (H,W); same numpy calls as skimage so the host BLAS rounding applies."""
    image = np.asanyarray(image)
    if image.shape[-1] != 3:
        raise ValueError(f"the input array must have size 3 along `channel_axis`, got {image.shape}")
    if image.dtype.kind == "u":
        imax = np.iinfo(image.dtype).max
        rgb = np.multiply(image, 1.0 / imax, dtype=np.float64)      # dtype.py:328
    elif image.dtype.kind == "f":
        rgb = image.astype(np.float64)
    else:
        raise ValueError("oracle supports unsigned and float colour input only")
    coeffs = np.array(GRAY_COEFFS, dtype=rgb.dtype)
    return rgb @ coeffs


# --------------------------------------------------------------------------------------------------
# D2  adjust_gamma  (exposure.py:611-681)
# --------------------------------------------------------------------------------------------------
def gamma_lut_u8(gamma: float = GAMMA) -> np.ndarray:
    lut = 255 * 1 * (np.linspace(0, 1, 256) ** gamma)
    return np.minimum(np.rint(lut), 255).astype("uint8")


def gamma_lut_u16(gamma: float = GAMMA) -> np.ndarray:
    x = np.arange(65536, dtype=np.uint16)
    return (((x / 65535.0) ** gamma) * 65535.0 * 1).astype(np.uint16)


def adjust_gamma(image: np.ndarray, gamma: float = 1.0) -> np.ndarray:
    if gamma < 0:
        raise ValueError("Gamma should be a non-negative real number.")
    if image.dtype == np.uint8:
        return gamma_lut_u8(gamma)[image]
    if np.any(image < 0):
        raise ValueError("Image Correction methods work correctly only on images with non-negative values. "
                         "Use skimage.exposure.rescale_intensity.")
    if image.dtype.kind == "u":
        scale = float(np.iinfo(image.dtype).max)
    elif image.dtype.kind == "i":
        scale = float(np.iinfo(image.dtype).max)          # dtype_limits(clip_negative=True) -> (0, max)
    else:
        scale = 1.0
    return (((image / scale) ** gamma) * scale * 1).astype(image.dtype)


# --------------------------------------------------------------------------------------------------
# D3  threshold_otsu  (thresholding.py:312-381)
# --------------------------------------------------------------------------------------------------
def histogram_image_range(image: np.ndarray, nbins: int = 256):
    """exposure.histogram(image.reshape(-1), nbins, source_range='image') -> (counts, bin_centers)."""
    flat = image.reshape(-1)
    if np.issubdtype(flat.dtype, np.integer):
        lo = int(flat.min())
        hi = int(flat.max())
        centers = np.arange(lo, hi + 1)
        if lo < 0:
            counts = np.bincount((flat.astype(np.int64) - lo).ravel(), minlength=hi - lo + 1)
        else:
            counts = np.bincount(flat.ravel(), minlength=hi + 1)[lo:]
        return counts, centers
    counts, edges = np.histogram(flat, bins=nbins, range=None)
    return counts, (edges[:-1] + edges[1:]) / 2.0


def otsu_from_hist(counts, centers):
    """thresholding.py:365-381 on a histogram (counts, bin_centers): the threshold and the index of its bin."""
    counts = np.asarray(counts).astype("float32", copy=False)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = int(np.argmax(variance12))
    return centers[idx], idx


def threshold_otsu(image: np.ndarray, nbins: int = 256):
    first = image.reshape(-1)[0]
    if np.all(image == first):
        return first
    counts, centers = histogram_image_range(image, nbins)
    return otsu_from_hist(counts, centers)[0]


# --------------------------------------------------------------------------------------------------
# D6  vertical opening with rectangle(line_length, 1)  (binary.py:79-88, :138-145)
# --------------------------------------------------------------------------------------------------
def vertical_opening(mask: np.ndarray, line_length: int = 10) -> np.ndarray:
    fp = np.ones((line_length, 1), dtype=np.uint8)
    eroded = np.empty(mask.shape, dtype=bool)
    ndi.binary_erosion(mask, structure=fp, output=eroded, border_value=True)
    dilated = np.empty(mask.shape, dtype=bool)
    ndi.binary_dilation(eroded, structure=fp, output=dilated)
    return dilated


# --------------------------------------------------------------------------------------------------
# D7  label (8-connectivity, raster order of first pixel)  (_label.py:116-118)
# --------------------------------------------------------------------------------------------------
def label8(mask: np.ndarray):
    labels, k = ndi.label(mask, structure=np.ones((3, 3), dtype=int))
    return labels.astype(np.int32, copy=False), int(k)


def region_table(labels: np.ndarray, k: int) -> np.ndarray:
    """(k,5) int64: minr, minc, maxr, maxc (half-open), area -- regionprops bbox/area, label order."""
    out = np.zeros((k, 5), dtype=np.int64)
    if k == 0:
        return out
    slices = ndi.find_objects(labels)
    areas = np.bincount(labels.ravel(), minlength=k + 1)
    for i, sl in enumerate(slices):
        out[i] = (sl[0].start, sl[1].start, sl[0].stop, sl[1].stop, areas[i + 1])
    return out


# --------------------------------------------------------------------------------------------------
# D9-D11  region test, duplicate rejection, outlier removal, sort  (detection.py:67-93, :118-153)
# --------------------------------------------------------------------------------------------------
def check_duplicate(minr, minc, maxr, maxc, lst, min_maxr, min_maxc) -> bool:
    for a, b, c, d in lst:
        b = max(b, a + min_maxr)
        d = max(d, c + min_maxc)
        if minr <= b and maxr >= a and minc <= d and maxc >= c:
            return False
    return True


def remove_outliers(lst, min_maxc):
    widths = [max(d, c + min_maxc) - c for _, _, c, d in lst]
    if not widths:
        return []
    wmin = min(widths)
    return [r for r, w in zip(lst, widths) if w <= wmin * 2]


def detect_debug(image: np.ndarray, min_maxr: int, min_maxc: int, line_length: int = 10) -> dict:
    """Every intermediate of detection.py:31-94 that determines the return value."""
    image = np.asarray(image)
    if image.ndim == 2:
        gray = image
    elif image.ndim == 3:
        gray = rgb2gray(image)
    else:
        raise ValueError("Invalid image dimensions. Only 2D and 3D images are supported.")
    v = adjust_gamma(gray, GAMMA)
    thr = threshold_otsu(v)
    mask = v > thr
    opened = vertical_opening(mask, line_length)
    labels, k = label8(opened)
    bg = float(np.mean(labels[0:1, 0:1], axis=(0, 1)))
    table = region_table(labels, k)
    accepted = []
    accepted_labels = []
    for lab in range(1, k + 1):
        minr, minc, maxr, maxc, area = (int(t) for t in table[lab - 1])
        maxr, maxc = max(maxr, minr + min_maxr), max(maxc, minc + min_maxc)
        if area < 100:
            continue                      # `and region.area >= 100` (detection.py:81); skipping the mean is equivalent
        if image.ndim == 2:
            mean_color = np.mean(image[minr:maxr, minc:maxc])
        else:
            mean_color = np.mean(image[minr:maxr, minc:maxc, :], axis=(0, 1))
        if np.mean(mean_color) > bg:
            if check_duplicate(minr, minc, maxr, maxc, accepted, min_maxr, min_maxc):
                accepted.append([minr, maxr, minc, maxc])
                accepted_labels.append((lab, area))
    kept = remove_outliers(accepted, min_maxc)
    kept_meta = [accepted_labels[accepted.index(r)] for r in kept]
    order = sorted(range(len(kept)), key=lambda i: (kept[i][2], kept[i][3]))
    strips = [kept[i] for i in order]
    meta = [kept_meta[i] for i in order]
    return dict(gamma=v, threshold=thr, mask=mask, opened=opened, labels=labels, n_labels=k, background=bg,
                table=table, accepted=accepted, strips=strips,
                strip_labels=[m[0] for m in meta], strip_areas=[m[1] for m in meta])


def detect_lateral_flow_tests(image: np.ndarray, min_maxr: int, min_maxc: int, line_length: int = 10):
    d = detect_debug(image, min_maxr, min_maxc, line_length)
    return [image[a:b, c:e] for a, b, c, e in d["strips"]]


# --------------------------------------------------------------------------------------------------
# profile + peaks  (detection.py:205-316)
# --------------------------------------------------------------------------------------------------
def _trapz(y):
    fn = getattr(np, "trapezoid", None) or np.trapz
    return fn(y)


def analyze_peaks(profile, inverse: bool = False, window_size: int = 25):
    n = len(profile)
    peak = int(np.argmin(profile) if inverse else np.argmax(profile))
    lo = max(0, peak - window_size)
    hi = min(n, peak + window_size + 1)
    thr = np.mean(profile[lo:hi])
    s = peak
    e = peak
    if inverse:
        while s > 0 and (profile[s] <= profile[s - 1] or profile[s] < thr):
            s -= 1
        while e < n - 1 and (profile[e] <= profile[e + 1] or profile[e] < thr):
            e += 1
    else:
        while s > 0 and (profile[s] >= profile[s - 1] or profile[s] > thr):
            s -= 1
        while e < n - 1 and (profile[e] >= profile[e + 1] or profile[e] > thr):
            e += 1
    base = np.linspace(profile[s], profile[e], e - s + 1)
    seg = profile[s:e + 1]
    area = (_trapz(base) - _trapz(seg)) if inverse else (_trapz(seg) - _trapz(base))
    return s, e, area


def _window(region: np.ndarray, selected: bool):
    h, w = region.shape[0], region.shape[1]
    if selected:
        return region
    return region[int(h * .25):int(h * .6), int(w * .3):int(w * .7)]


def _two_peaks(profile, inverse: bool, rng: int, fallback: bool):
    n = len(profile)
    s, e, a = analyze_peaks(profile, inverse=inverse)
    first = [(s, e, a)]
    if not (s > 1 and e < n - 2):
        return first
    if fallback:
        left = profile[:s - rng] if s - rng > 0 else profile[:s - 1]
        right = profile[e + rng:] if e + rng < n else profile[e + 1:]
    else:
        left = profile[:s - rng]
        right = profile[e + rng:]
    if not (len(left) > 1 and len(right) > 1):
        return first
    if inverse:
        take_left = right[np.argmin(right)] > left[np.argmin(left)]
    else:
        take_left = right[np.argmax(right)] < left[np.argmax(left)]
    side = left if take_left else right
    s2, e2, a2 = analyze_peaks(side, inverse=inverse)
    if not take_left:
        off = e + rng if fallback else e + 1          # detection.py:238-239 adds `range`, :274-275 the literal 1
        s2 += off
        e2 += off
    if (s2 <= s and e2 > s) or (s2 <= e and e2 > e) or (s2 >= s and e2 <= e):
        return first
    return first + [(s2, e2, a2)]


def image_analysis(region: np.ndarray, selected: bool, range: int = 15):
    """detection.py:205-245."""
    win = _window(np.asarray(region), selected)
    axis = (1, 2) if win.ndim == 3 else 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        profile = np.nanmean(win, axis=axis)
    return [profile, _two_peaks(profile, False, range, True)]


def inverse_image_analysis(region: np.ndarray, selected: bool, range: int = 1):
    """detection.py:247-281 (3-D input only: the reference unpacks three dimensions)."""
    region = np.asarray(region)
    h, w, ch = region.shape
    win = region if selected else region[int(h * .25):int(h * .6), int(w * .3):int(w * .7), :]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        profile = np.nanmean(win, axis=(1, 2))
    return [profile, _two_peaks(profile, True, range, False)]


# --------------------------------------------------------------------------------------------------
# analyze_plot pre-processing  (detection.py:535-543)
# --------------------------------------------------------------------------------------------------
def rescale_intensity(image: np.ndarray, in_range) -> np.ndarray:
    """skimage.exposure.rescale_intensity(image, in_range=(lo, hi)) with out_range='dtype' (exposure.py:577-600)
    for the unsigned integer images of this path: the result has the input's dtype."""
    image = np.asarray(image)
    if image.dtype.kind != "u":
        raise NotImplementedError("oracle restates rescale_intensity for unsigned integer images only")
    imin, imax = map(float, in_range)                                  # :582 (None -> TypeError, like the reference)
    omin, omax = 0.0, float(np.iinfo(image.dtype).max)                  # :583 intensity_range(image, 'dtype')
    out = np.clip(image, imin, imax)                                   # :594 (float64: Python-float bounds)
    if imin != imax:
        out = (out - imin) / (imax - imin)                             # :597
        return np.asarray(out * (omax - omin) + omin, dtype=image.dtype)   # :598
    return np.clip(out, omin, omax).astype(image.dtype)                # :600


def invert(image: np.ndarray) -> np.ndarray:
    """skimage.util.invert for unsigned integers (util/_invert.py:64-66): dtype max - image."""
    image = np.asarray(image)
    if image.dtype.kind != "u":
        raise NotImplementedError("oracle restates invert for unsigned integer images only")
    return np.subtract(np.iinfo(image.dtype).max, image, dtype=image.dtype)


def analyze_plot(region: np.ndarray, selected: bool, gamma_level: float = 1, low_exposure=None, high_exposure=None,
                 invert_image: bool = False):
    """MyGUI.analyze_plot's data path (detection.py:535-543): gamma, optional rescale, optional invert, then
    image_analysis / inverse_image_analysis."""
    region = adjust_gamma(np.asarray(region), gamma_level)
    if low_exposure is not None or high_exposure is not None:
        region = rescale_intensity(region, (low_exposure, high_exposure))
    if invert_image:
        return inverse_image_analysis(invert(region), selected)
    return image_analysis(region, selected)


# --------------------------------------------------------------------------------------------------
# gaussian  (_shared/filters.py:35-166 ; only caller detection.py:166)
# --------------------------------------------------------------------------------------------------
def gaussian(image: np.ndarray, sigma: float = 1.0) -> np.ndarray:
    image = np.asarray(image)
    if image.dtype.kind == "u":
        f = np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)
    else:
        f = image.astype(np.float64)
    if image.ndim == 3 and image.shape[-1] == 3:
        sig = [sigma, sigma, 0]
    else:
        sig = sigma
    return ndi.gaussian_filter(f, sig, mode="nearest", cval=0, truncate=4.0)


# --------------------------------------------------------------------------------------------------
# end-to-end helper used by the CPU baseline
# --------------------------------------------------------------------------------------------------
def analyze_image(image: np.ndarray, min_maxr: int | None = None, min_maxc: int | None = None):
    """detect + profile + peaks for one image with the GUI's parameter rule (detection.py:457-460)."""
    if min_maxc is None:
        min_maxc = int(image.shape[1] / 40)
    if min_maxr is None:
        min_maxr = int(min_maxc * 20)
    strips = detect_lateral_flow_tests(image, min_maxr, min_maxc)
    return [image_analysis(s, False) for s in strips]
