"""CPU oracle for `detect_lines` (TEST INFRASTRUCTURE ONLY; same rules as oracle/lfa_oracle.py: imported by tests/ only,
never by the product path).

numpy + scipy restatement of LFA-Image_Analysis/detection.py:156-201, a function the reference defines and never calls
(SURVEY.md section 8f.3).  Paths below are relative to /root/reference; SP = image-analysis/lib/python3.10/site-packages.

    detection.py:166   gaussian                 SP/skimage/_shared/filters.py:35-166 (sigma 1, mode 'nearest', truncate 4)
    detection.py:167   threshold_local          SP/skimage/filters/thresholding.py:157-253 (gaussian, sigma (3-1)/6, 'reflect')
    detection.py:173   adjust_gamma(float, .5)  SP/skimage/exposure/exposure.py:611-681 (scale 1 for floats: image ** 0.5)
    detection.py:175   equalize_adapthist       SP/skimage/exposure/_adapthist.py:21-309 ; util/dtype.py (img_as_uint)
    detection.py:179   np.percentile            numpy, method 'linear'
    detection.py:180   rescale_intensity        SP/skimage/exposure/exposure.py:485-600
    detection.py:184   label (non-bool input)   SP/skimage/measure/_label.py:30-120 -> _ccomp (Cython, darwin binary only)
    detection.py:187   regionprops .bbox        SP/skimage/measure/_regionprops.py

CLAHE is written here pixel by pixel (which tile maps a pixel sees, with which weights, in which order the four float32
terms are added) instead of through skimage's block reshapes, i.e. the way the HIP kernels compute it.

Pinning: tests/golden/make_golden.py runs the reference's own lines :156-201 over the vendored skimage sources for every
stage up to and including the percentile rescale (tests/golden/ref_detect_lines.npz; tests/test_lines_oracle.py holds
this file to them bit for bit).  `label` of a non-boolean image is Cython in skimage (`_ccomp`, shipped as a
darwin-arm64 binary only): its rule -- neighbours (8-connected in 2-D) with EQUAL value belong together, value 0 is
background, labels number the components in raster order of their first pixel -- is restated in `label_equal` and pinned by
the known-answer arrays the vendored test-suite holds (SP/skimage/measure/tests/test_ccomp.py:14-24, :58-62, :76-99), not by
a run of the reference on full frames: "parity unpinned" for that one step beyond those vectors.

The image detect_lines hands to label() is FLOAT64 in [0, 1].  `_ccomp.label_cython` keeps its data in a typed buffer
`cnp.ndarray[DTYPE_t]` with DTYPE = np.intp (the binary holds the names DTYPE, intp, array, order and the message "Buffer
dtype mismatch"; `_label.py:40-51` documents "ndarray of dtype int"), so the image is converted to intp first: every value
below 1.0 becomes background 0 and only the pixels whose rescaled value is exactly 1.0 -- at or above the 99th percentile --
are foreground, all of one class.  `label_as_skimage` restates that (round 3; before, every distinct float level was taken
as a class of its own, which no integer vector of the vendored suite could tell apart from the truncating rule).  The
conversion itself cannot be run here (darwin binary): it stays "parity unpinned".
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

NR_OF_GRAY = 2 ** 14          # _adapthist.py:17
NBINS = 256                   # equalize_adapthist default
CLIP_LIMIT = 0.05             # detection.py:176


def gaussian_float(image: np.ndarray) -> np.ndarray:
    """sk.filters.gaussian(image) of a 2-D uint8 / uint16 / float image (detection.py:166): float64, sigma 1, 'nearest'."""
    image = np.asanyarray(image)
    if image.dtype.kind == "u":
        x = np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)     # img_as_float, dtype.py:328
    elif image.dtype.kind == "f":
        x = image.astype(np.float64)
    else:
        raise ValueError("oracle: unsigned or float images")
    return ndi.gaussian_filter(x, 1, mode="nearest", cval=0, truncate=4.0)


def local_threshold(g: np.ndarray) -> np.ndarray:
    """threshold_local(g, block_size=3, method='gaussian', mode='reflect') (thresholding.py:236-243): sigma = 2/6 per axis."""
    return ndi.gaussian_filter(g, ((3 - 1) / 6.0,) * g.ndim, mode="reflect", cval=0, truncate=4.0) - 0


def to_uint16(x: np.ndarray) -> np.ndarray:
    """img_as_uint of a float64 image in [0, 1] (dtype.py _convert: multiply by 65535, rint, clip, cast)."""
    if x.min() < -1.0 or x.max() > 1.0:
        raise ValueError("Images of type float must be between -1 and 1.")
    y = np.multiply(x, 65535.0, dtype=np.float64)
    np.rint(y, out=y)
    np.clip(y, 0, 65535, out=y)
    return y.astype(np.uint16)


def clip_histogram(hist: np.ndarray, clip_limit: int) -> np.ndarray:
    """_adapthist.py:229-280, same steps (int64 counts)."""
    hist = hist.astype(np.int64).copy()
    over = hist > clip_limit
    n_excess = int(hist[over].sum() - over.sum() * clip_limit)
    hist[over] = clip_limit
    bin_incr = n_excess // hist.size
    upper = clip_limit - bin_incr
    low = hist < upper
    n_excess -= int(low.sum()) * bin_incr
    hist[low] += bin_incr
    mid = (hist >= upper) & (hist < clip_limit)
    n_excess += int(hist[mid].sum() - mid.sum() * clip_limit)
    hist[mid] = clip_limit
    while n_excess > 0:
        before = n_excess
        for index in range(hist.size):
            under = hist < clip_limit
            step = max(1, int(np.count_nonzero(under)) // n_excess)
            sel = np.zeros(hist.size, bool)
            sel[index::step] = under[index::step]
            hist[sel] += 1
            n_excess -= int(sel.sum())
            if n_excess <= 0:
                break
        if before == n_excess:
            break
    return hist


def clahe_geometry(H: int, W: int):
    """Tile size, padding and tile counts of _clahe for an (H, W) image with the default kernel size (_adapthist.py:78-79,
    :123-126, :141, :176)."""
    k = (max(H // 8, 1), max(W // 8, 1))
    start = tuple(kk // 2 for kk in k)
    end = tuple((kk - s % kk) % kk + int(math.ceil(kk / 2.0)) for kk, s in zip(k, (H, W)))
    padded = tuple(a + s + b for a, s, b in zip(start, (H, W), end))
    ns_hist = tuple(int(p / kk) - 1 for p, kk in zip(padded, k))
    return k, start, end, padded, ns_hist


def clahe_levels(u: np.ndarray, clip_limit: float = CLIP_LIMIT):
    """equalize_adapthist up to `_clahe`'s uint16 result (_adapthist.py:72-87): u is the uint16 image img_as_uint made.
    Returns (levels uint16 (H, W) in 0..16383, v14, maps)."""
    H, W = u.shape
    imin, imax = float(u.min()), float(u.max())
    # rescale_intensity(image, out_range=(0, NR_OF_GRAY - 1)) -> float64, then np.round and the cast (:74-76)
    c = np.clip(u, imin, imax)
    if imin != imax:
        r = (c - imin) / (imax - imin)
        r = np.asarray(r * (float(NR_OF_GRAY - 1) - 0.0) + 0.0, dtype=np.float64)
    else:
        r = np.clip(c, 0.0, float(NR_OF_GRAY - 1)).astype(np.float64)
    v14 = np.round(r).astype(np.uint16)
    k, start, end, padded, ns_hist = clahe_geometry(H, W)
    p = np.pad(v14, [[start[0], end[0]], [start[1], end[1]]], mode="reflect")
    bin_size = 1 + NR_OF_GRAY // NBINS
    b = (p // bin_size).astype(np.int64)                                   # lut[image], :133-137
    n_el = k[0] * k[1]
    clim = int(np.clip(clip_limit * n_el, 1, None)) if clip_limit > 0.0 else n_el
    maps = np.empty(ns_hist + (NBINS,), np.int64)
    for ty in range(ns_hist[0]):
        for tx in range(ns_hist[1]):
            y0, x0 = k[0] // 2 + ty * k[0], k[1] // 2 + tx * k[1]
            h = np.bincount(b[y0:y0 + k[0], x0:x0 + k[1]].ravel(), minlength=NBINS)
            h = clip_histogram(h, clim)
            m = np.cumsum(h).astype(float)                                 # map_histogram, :305-310
            m *= (NR_OF_GRAY - 1 - 0) / n_el
            m += 0
            np.clip(m, a_min=None, a_max=NR_OF_GRAY - 1, out=m)
            maps[ty, tx] = m.astype(int)                                   # (:310: the mapping is truncated to integers)
    # interpolation (:176-210), pixel by pixel over the padded image
    py, px = np.meshgrid(np.arange(padded[0]), np.arange(padded[1]), indexing="ij")
    by, bx = py // k[0], px // k[1]
    cy = (py % k[0]) / k[0]
    cx = (px % k[1]) / k[1]
    res = np.zeros(padded, np.float32)
    for e0 in (0, 1):
        for e1 in (0, 1):
            ty = np.clip(by + e0 - 1, 0, ns_hist[0] - 1)                   # map_array: edge-padded by one tile, :166-168
            tx = np.clip(bx + e1 - 1, 0, ns_hist[1] - 1)
            coeff = (cx if e1 else 1 - cx) * (cy if e0 else 1 - cy)
            res += (maps[ty, tx, b] * coeff).astype(np.float32)
    lv = res.astype(np.uint16)[start[0]:padded[0] - end[0], start[1]:padded[1] - end[1]]
    return lv, v14, maps


def rescale_unit(x: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """rescale_intensity(x, in_range=(lo, hi), out_range=(0, 1)) and, with lo, hi = min, max, rescale_intensity(x) of a
    float64 image (exposure.py:577-589)."""
    lo, hi = float(lo), float(hi)
    c = np.clip(x, lo, hi)
    if lo != hi:
        return np.asarray((c - lo) / (hi - lo) * (1.0 - 0.0) + 0.0, dtype=np.float64)
    return np.clip(c, 0.0, 1.0).astype(np.float64)


def label_equal(values: np.ndarray) -> np.ndarray:
    """skimage.measure.label of a non-boolean 2-D image with the defaults (background 0, full connectivity): pixels that
    are 8-neighbours and hold the same value belong together; labels count the components in raster order of their first
    pixel."""
    H, W = values.shape
    idx = np.arange(H * W).reshape(H, W)
    rows, cols = [], []
    for dy, dx in ((0, 1), (1, -1), (1, 0), (1, 1)):
        a = (slice(0, H - dy), slice(max(0, -dx), W - max(0, dx)))
        b = (slice(dy, H), slice(max(0, dx), W - max(0, -dx)))
        same = (values[a] == values[b]) & (values[a] != 0)
        rows.append(idx[a][same]); cols.append(idx[b][same])
    r, c = np.concatenate(rows), np.concatenate(cols)
    _, comp = connected_components(coo_matrix((np.ones(r.size, np.int8), (r, c)), shape=(H * W, H * W)), directed=False)
    fg = np.flatnonzero(values.ravel() != 0)
    out = np.zeros(H * W, np.int64)
    if fg.size:
        ids, first = np.unique(comp[fg], return_index=True)               # fg ascends: `first` orders by first pixel
        rank = np.empty(ids.size, np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(1, ids.size + 1)
        out[fg] = rank[np.searchsorted(ids, comp[fg])]
    return out.reshape(H, W)


def label_as_skimage(values: np.ndarray) -> np.ndarray:
    """skimage.measure.label(values) for any non-boolean dtype: `_ccomp.label_cython` works on an intp copy of the image
    (typed buffer `cnp.ndarray[DTYPE_t]`, DTYPE = np.intp), i.e. floats are truncated toward zero before neighbours are
    compared.  On detect_lines' image in [0, 1]: foreground = the pixels equal to 1.0."""
    return label_equal(np.array(values, dtype=np.intp))


def lines_debug(image: np.ndarray) -> dict:
    """Every intermediate of detect_lines (detection.py:156-201) for a 2-D image."""
    image = np.asanyarray(image)
    if image.ndim != 2:
        raise ValueError("too many values to unpack (expected 4)")           # regionprops .bbox of an n-D label image, :189
    H, W = image.shape
    min_area = len(image[0]) * (len(image) * 0.0065)
    max_area = len(image[0]) * (len(image) * 0.04)
    g = gaussian_float(image)
    t = local_threshold(g)
    x = g * (t > 0)
    x = ((x / 1.0) ** 0.5) * 1.0 * 1                                        # adjust_gamma, exposure.py:679 (scale 1, gain 1)
    u = to_uint16(x)
    lv, v14, maps = clahe_levels(u)
    e = rescale_unit(lv.astype(np.float64), lv.min(), lv.max())             # _adapthist.py:88-89
    p80, p99 = np.percentile(e, (80, 99))
    e2 = rescale_unit(e, p80, p99)
    lab = label_as_skimage(e2)
    strips = []
    objs = ndi.find_objects(lab)
    boxes = np.zeros((len(objs), 4), np.int64)
    for i, sl in enumerate(objs):
        minr, minc, maxr, maxc = sl[0].start, sl[1].start, sl[0].stop, sl[1].stop
        boxes[i] = (minr, minc, maxr, maxc)
        area = (maxc - minc) * (maxr - minr)
        if area > min_area and (maxc - minc) > (maxr - minr):
            minc = 0
            maxc = len(image[0])
        if ((maxc - minc) > (maxr - minr)) and (area < max_area and area > min_area):
            strips.append([minr, maxr, minc, maxc])
    return dict(gauss=g, local=t, gamma=x, u16=u, v14=v14, maps=maps, levels=lv, enhanced=e, p80=p80, p99=p99, rescaled=e2,
                labels=lab, boxes=boxes, strips=strips)


def detect_lines(image: np.ndarray) -> list:
    return lines_debug(image)["strips"]
