"""ctypes binding of include/lfa.h (liblfa_hip.so) plus the torch hand-off of device buffers.

torch is used for exactly three things: owning device memory, host<->device copies, and naming the current HIP
stream.  All arithmetic happens in the HIP library; if the library is missing this module raises at import --
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFA_LIB_PATH") or os.path.join(_HERE, "liblfa_hip.so")     # LFA_LIB_PATH: an experimental build (tools/)

LFA_U8, LFA_U16, LFA_F64 = 0, 1, 6
ELEM_CODES = {"uint8": 0, "uint16": 1, "int8": 2, "int16": 3, "int32": 4, "uint32": 5, "float64": 6}   # lfa.h LFA_E_*
DBG_MASK, DBG_OPENED, DBG_LABELS, DBG_TABLE, DBG_HIST, DBG_PLANE, DBG_ONEPASS, DBG_GREY = 0, 1, 2, 3, 4, 5, 6, 7
GRAY_NVARIANTS = 6

# numpy mirrors of the C structs (align=True reproduces the C layout)
STRIP_DT = np.dtype([("minr", "<i4"), ("maxr", "<i4"), ("minc", "<i4"), ("maxc", "<i4"), ("area", "<i4"),
                     ("label", "<i4")], align=True)
ROI_DT = np.dtype([("image", "<i4"), ("r0", "<i4"), ("r1", "<i4"), ("c0", "<i4"), ("c1", "<i4")], align=True)
PEAKS_DT = np.dtype([("n_peaks", "<i4"), ("length", "<i4"), ("start", "<i4", (2,)), ("end", "<i4", (2,)),
                     ("area", "<f8", (2,))], align=True)
INFO_DT = np.dtype([("xmin", "<i4"), ("xmax", "<i4"), ("v_thr", "<i4"), ("x_thr", "<i4"),
                    ("gmin", "<f8"), ("gmax", "<f8"), ("vmin", "<f8"), ("vmax", "<f8"), ("v_thr_f", "<f8"),
                    ("g_thr", "<f8"), ("otsu_bin", "<i4"), ("all_equal", "<i4"), ("ambiguous", "<u4"),
                    ("n_runs", "<i4"), ("n_labels", "<i4"), ("label00", "<i4"), ("n_candidates", "<i4"),
                    ("n_accepted", "<i4"), ("n_strips", "<i4"), ("status", "<i4")], align=True)
LINEBOX_DT = np.dtype([("minr", "<i4"), ("maxr", "<i4"), ("minc", "<i4"), ("maxc", "<i4")], align=True)
LINESINFO_DT = np.dtype([("umin", "<i4"), ("umax", "<i4"), ("cmin", "<i4"), ("cmax", "<i4"), ("p80", "<f8"), ("p99", "<f8"),
                         ("n_labels", "<i4"), ("n_boxes", "<i4"), ("status", "<i4"), ("pad", "<i4")], align=True)
LDBG_GAUSS, LDBG_U16, LDBG_LEVELS, LDBG_ROOTS, LDBG_LVAL = 0, 1, 2, 3, 4
assert STRIP_DT.itemsize == 24 and ROI_DT.itemsize == 20 and PEAKS_DT.itemsize == 40 and INFO_DT.itemsize == 104
assert LINEBOX_DT.itemsize == 16 and LINESINFO_DT.itemsize == 48


class LfaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"liblfa_hip error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; it must be the one HIP runtime of the process, so load it first and let
    # liblfa_hip.so bind to it (two HIP/HSA runtimes in one process do not both see the device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.lfa_version.restype = i32
    lib.lfa_last_error.restype = C.c_char_p
    lib.lfa_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.lfa_destroy.argtypes = [vp]
    lib.lfa_set_gamma_lut.argtypes = [vp, i32, vp, i32]
    lib.lfa_set_gray_variant.argtypes = [vp, i32]
    lib.lfa_set_debug_flags.argtypes = [vp, i32]
    lib.lfa_set_host_pow.argtypes = [vp, vp, i32]
    lib.lfa_set_debug_flags.restype = i32
    lib.lfa_detect_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp]
    lib.lfa_profile_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.lfa_profile_lut_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp, vp]
    lib.lfa_peaks_batch.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.lfa_profile_any_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.lfa_analyze_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.lfa_analyze_batch_static.argtypes = lib.lfa_analyze_batch.argtypes
    lib.lfa_reserve.argtypes = [vp, i32, i32]
    lib.lfa_workspace_bytes.argtypes = [vp]
    lib.lfa_workspace_bytes.restype = i64
    lib.lfa_static_workspace_bytes.restype = i64
    lib.lfa_gaussian_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, dbl, vp, i32, vp, i32, vp]
    lib.lfa_debug_fetch.argtypes = [vp, i32, i32, vp, i64, vp]
    lib.lfa_detect_lines_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    lib.lfa_lines_debug_fetch.argtypes = [vp, i32, i32, vp, i64, vp]
    lib.lfa_sync.argtypes = [vp, vp]
    lib.lfa_stream_wait_passes.argtypes = [vp, vp]
    lib.lfa_chain_passes.argtypes = [vp, vp]
    lib.lfa_timing_enable.argtypes = [vp, C.c_char_p]
    lib.lfa_timing_read.argtypes = [vp, i32, vp, vp, vp, vp]
    for name in ("lfa_timing_enable", "lfa_timing_read","lfa_create", "lfa_destroy", "lfa_set_gamma_lut", "lfa_set_gray_variant", "lfa_detect_batch",
                 "lfa_profile_batch", "lfa_profile_lut_batch", "lfa_peaks_batch", "lfa_analyze_batch", "lfa_gaussian_batch",
                 "lfa_debug_fetch", "lfa_sync", "lfa_stream_wait_passes", "lfa_chain_passes", "lfa_analyze_batch_static", "lfa_reserve",
                 "lfa_detect_lines_batch", "lfa_lines_debug_fetch", "lfa_profile_any_batch", "lfa_set_host_pow"):
        getattr(lib, name).restype = i32
    return lib


lib = _load()

EXPORTED = ("lfa_version", "lfa_last_error", "lfa_create", "lfa_destroy", "lfa_set_gamma_lut", "lfa_set_gray_variant",
            "lfa_set_debug_flags",
            "lfa_detect_batch", "lfa_profile_batch", "lfa_profile_lut_batch", "lfa_peaks_batch", "lfa_analyze_batch",
            "lfa_gaussian_batch", "lfa_debug_fetch", "lfa_sync", "lfa_timing_enable", "lfa_timing_read",
            "lfa_stream_wait_passes", "lfa_chain_passes", "lfa_analyze_batch_static", "lfa_static_workspace_bytes", "lfa_reserve",
            "lfa_workspace_bytes", "lfa_detect_lines_batch", "lfa_lines_debug_fetch",
            "lfa_profile_any_batch", "lfa_set_host_pow")


def _check(rc):
    if rc != 0:
        raise LfaError(rc, lib.lfa_last_error().decode(errors="replace"))


# ---- host-side tables that must carry the HOST numpy's bits ----------------------------------------------------------
def numpy_gamma_lut_u8(gamma=0.4):
    """skimage.exposure._adjust_gamma_u8 (exposure.py:611-617), built with this host's numpy."""
    lut = 255 * 1 * (np.linspace(0, 1, 256) ** gamma)
    return np.minimum(np.rint(lut), 255).astype("uint8")


def numpy_gamma_lut_u16(gamma=0.4):
    """adjust_gamma on every uint16 value (exposure.py:674-679), built with this host's numpy."""
    x = np.arange(65536, dtype=np.uint16)
    return (((x / 65535.0) ** gamma) * 65535.0 * 1).astype(np.uint16)


def point_op_lut(dtype, gamma_level=1, low_exposure=None, high_exposure=None, invert=False):
    """Table of MyGUI.analyze_plot's point operations (detection.py:535-539) over every value of `dtype` (uint8/uint16):
    adjust_gamma (exposure.py:666-681), rescale_intensity(in_range=(low, high)) when either bound is set
    (exposure.py:577-600, out_range='dtype'), util.invert (util/_invert.py:64-66).  Computed with this host's numpy on
    arange(all values), so table[v] has exactly the bits the reference produces for a pixel v."""
    dt = np.dtype(dtype)
    if dt == np.uint8:
        if gamma_level < 0:
            raise ValueError("Gamma should be a non-negative real number.")
        v = numpy_gamma_lut_u8(gamma_level)                      # adjust_gamma on uint8 IS this table (lut[image])
    elif dt == np.uint16:
        if gamma_level < 0:
            raise ValueError("Gamma should be a non-negative real number.")
        v = numpy_gamma_lut_u16(gamma_level)
    else:
        raise TypeError("point-operation tables exist for uint8 and uint16 pixels only")
    top = float(np.iinfo(dt).max)
    if low_exposure is not None or high_exposure is not None:
        imin, imax = map(float, (low_exposure, high_exposure))   # a single None raises TypeError, like the reference (:582)
        w = np.clip(v, imin, imax)
        if imin != imax:
            w = (w - imin) / (imax - imin)
            v = np.asarray(w * (top - 0.0) + 0.0, dtype=dt)
        else:
            v = np.clip(w, 0.0, top).astype(dt)
    if invert:
        v = np.subtract(np.iinfo(dt).max, v, dtype=dt)
    return np.ascontiguousarray(v, dtype=dt)


def detect_gray_variant(width=64):
    """Which rounding order `rgb @ coeffs` has in THIS host's numpy/BLAS for images `width` pixels wide (include/lfa.h
    LFA_GRAY_*).

    numpy evaluates `(H, W, 3) @ (3,)` row by row of the image: a matrix-vector product of a W x 3 matrix (BLAS gemv), and
    for W == 1 a plain dot product -- two routines with their own order of operations and use of fused multiply-adds, so
    the order is a function of the width.  Evaluates the candidate orders with exact float64 arithmetic on random
    triples laid out as rows of that width and returns the first that reproduces numpy's matmul on all of them
    (colorconv.py:876-877).  Raises if none does."""
    import math
    width = max(1, int(width))
    rows_n = max(2, -(-1024 // width))
    rng = np.random.default_rng(12345)
    px = rng.integers(0, 256, size=(rows_n * width, 3), dtype=np.uint8)
    x = np.multiply(px, 1.0 / 255, dtype=np.float64)
    c = np.array([0.2125, 0.7154, 0.0721])
    want = (x.reshape(rows_n, width, 3) @ c).reshape(-1)
    fma = getattr(math, "fma", None)
    if fma is None:
        libm = C.CDLL("libm.so.6")
        libm.fma.restype = C.c_double
        libm.fma.argtypes = [C.c_double] * 3
        fma = libm.fma
    c0, c1, c2 = (float(v) for v in c)
    forms = [
        lambda a, b, d: fma(d, c2, fma(a, c0, b * c1)),
        lambda a, b, d: (a * c0 + b * c1) + d * c2,
        lambda a, b, d: fma(d, c2, fma(b, c1, a * c0)),
        lambda a, b, d: fma(a, c0, fma(b, c1, d * c2)),
        lambda a, b, d: a * c0 + (b * c1 + d * c2),
        lambda a, b, d: fma(b, c1, a * c0) + d * c2,
    ]
    rows = [(float(a), float(b), float(d)) for a, b, d in x]
    for v, f in enumerate(forms):
        if all(f(*r) == w for r, w in zip(rows, want)):
            return v
    raise RuntimeError(f"this host's numpy matmul matches none of the known rgb2gray rounding orders for images {width} pixels wide")


_GRAY_VARIANT = {}


def host_gray_variant(width=64):
    """detect_gray_variant(width), remembered per width."""
    width = max(1, int(width))
    if width not in _GRAY_VARIANT:
        _GRAY_VARIANT[width] = detect_gray_variant(width)
    return _GRAY_VARIANT[width]


# ---- torch hand-off ----------------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def dev_bytes(nbytes, device):
    torch = _torch()
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def to_device(arr: np.ndarray, device):
    """numpy array (any dtype) -> 1-D uint8 CUDA tensor holding its bytes."""
    torch = _torch()
    a = np.ascontiguousarray(arr)
    flat = a.view(np.uint8).reshape(-1) if a.size else np.zeros(0, np.uint8)
    t = torch.from_numpy(flat.copy() if not flat.flags.writeable else flat)
    return t.to(device, non_blocking=False)


def from_device(t, dtype, count=None):
    a = t.cpu().numpy()
    n = (a.size // np.dtype(dtype).itemsize) if count is None else count
    return a[: n * np.dtype(dtype).itemsize].view(dtype)


def current_stream_ptr(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def np_dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.uint8:
        return LFA_U8
    if dt == np.uint16:
        return LFA_U16
    if dt == np.float64:
        return LFA_F64                 # lfa_detect_batch only (lfa.h); every other entry point refuses it
    raise TypeError(f"pixel dtype {dt} is not supported by the HIP path (uint8 and uint16 only)")


def host_threshold(info_row, colour: bool):
    """The reference's `threshold` value (detection.py:38/46) from an LfaImageInfo row.

    Integer images: exact on the device.  Colour images: the device reports gmin, gmax and the Otsu bin; the float64
    bin centre is rebuilt here with the HOST numpy's pow() so that it carries the same last bits as skimage's
    (thresholding.py:378-381, exposure.py:298-301)."""
    if not colour:
        return int(info_row["v_thr"])
    v = np.array([info_row["gmin"], info_row["gmax"]], dtype=np.float64) ** 0.4
    if info_row["all_equal"]:
        return float(v[0])
    edges = np.linspace(v[0], v[1], 257)
    k = int(info_row["otsu_bin"])
    return float((edges[k] + edges[k + 1]) / 2.0)


HOST_POW_DOUBLES = 2 + 257 + 256          # include/lfa.h LfaHostPow


def _first_gray_with(gmin, gmax, targets, strict, gamma):
    """Per target t: the smallest float64 g in [gmin, gmax] with g ** gamma >= t (strict: > t), by bisection over the bit
    patterns of the doubles (non-negative doubles are ordered like their int64 patterns) with THIS host's numpy power --
    the function the reference applies to every pixel (skimage/exposure/exposure.py:679).  +inf where gmax ** gamma
    does not reach the target, gmin where gmin ** gamma already does."""
    targets = np.asarray(targets, dtype=np.float64)

    def pred(bits):
        v = bits.view(np.float64) ** gamma
        return (v > targets) if strict else (v >= targets)

    lo = np.full(targets.shape, np.float64(gmin).view(np.int64), dtype=np.int64)
    hi = np.full(targets.shape, np.float64(gmax).view(np.int64), dtype=np.int64)
    none, at_min = ~pred(hi), pred(lo)
    while True:
        gap = hi - lo
        if not (gap > 1).any():
            break
        mid = lo + (gap >> 1)
        p = pred(mid)
        hi = np.where(p, mid, hi)
        lo = np.where(p, lo, mid)
    return np.where(none, np.inf, np.where(at_min, np.float64(gmin), hi.view(np.float64)))


def host_pow_tables(gmin, gmax, gamma=0.4):
    """One LfaHostPow record (include/lfa.h) for an image whose gray values span [gmin, gmax], every pow() evaluated by this
    host's numpy exactly as the reference evaluates it: vmin / vmax (exposure.py:679), the 257 histogram edges of
    threshold_otsu's 256-bin float histogram (np.linspace between them; exposure.py:298-301, numpy/lib/histograms.py:800-859)
    mapped back to gray space, and per Otsu bin the gray-space form of `gamma_image > bin centre` (thresholding.py:378-381,
    detection.py:47)."""
    v = np.array([gmin, gmax], dtype=np.float64) ** gamma
    out = np.empty(HOST_POW_DOUBLES, dtype=np.float64)
    out[0], out[1] = v[0], v[1]
    if v[0] == v[1]:
        out[2:] = np.inf
        out[2] = gmin
        return out
    edges = np.linspace(v[0], v[1], 257)
    centres = (edges[:-1] + edges[1:]) / 2.0
    out[2:259] = _first_gray_with(gmin, gmax, edges, False, gamma)
    out[259:] = _first_gray_with(gmin, gmax, centres, True, gamma)
    return out


class Engine:
    """One LfaHandle on one device.  Not thread-safe (like the reference's module globals)."""

    def __init__(self, device=0, max_h=4096, max_w=4096, max_c=3, max_batch=16, gray_variant=None, host_luts=True):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the LFA path runs on the GPU only (no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.max_batch = max_batch
        self._h = C.c_void_p()
        _check(lib.lfa_create(device, max_h, max_w, max_c, max_batch, C.byref(self._h)))
        if host_luts:
            l8 = numpy_gamma_lut_u8()
            l16 = numpy_gamma_lut_u16()
            _check(lib.lfa_set_gamma_lut(self._h, LFA_U8, l8.ctypes.data, 256))
            _check(lib.lfa_set_gamma_lut(self._h, LFA_U16, l16.ctypes.data, 65536))
        self._gray_fixed = gray_variant is not None                  # a caller's choice stays; otherwise it follows the image width
        self.gray_variant = host_gray_variant() if gray_variant is None else int(gray_variant)
        _check(lib.lfa_set_gray_variant(self._h, self.gray_variant))

    def close(self):
        if self._h:
            lib.lfa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw calls on device tensors (uint8 byte tensors) --------------------------------------------------------------
    def _gray_for(self, W, Cc):
        """rgb2gray's rounding order in the host's numpy depends on the image width (detect_gray_variant): set the handle to
        the one the reference would get for this call's images."""
        if Cc == 3 and not self._gray_fixed:
            v = host_gray_variant(W)
            if v != self.gray_variant:
                _check(lib.lfa_set_gray_variant(self._h, v))
                self.gray_variant = v

    def detect_raw(self, images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, max_strips):
        self._gray_for(W, Cc)
        strips = dev_bytes(n * max_strips * STRIP_DT.itemsize, self.device)
        counts = dev_bytes(n * 4, self.device)
        info = dev_bytes(n * INFO_DT.itemsize, self.device)
        _check(lib.lfa_detect_batch(self._h, images_t.data_ptr(), dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length,
                                    strips.data_ptr(), max_strips, counts.data_ptr(), info.data_ptr(),
                                    current_stream_ptr(self.device)))
        return strips, counts, info

    def analyze_raw(self, images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, max_strips, stride):
        torch = _torch()
        self._gray_for(W, Cc)
        strips = dev_bytes(n * max_strips * STRIP_DT.itemsize, self.device)
        counts = dev_bytes(n * 4, self.device)
        info = dev_bytes(n * INFO_DT.itemsize, self.device)
        rois = dev_bytes(n * max_strips * ROI_DT.itemsize, self.device)
        profiles = torch.empty(n * max_strips * stride, dtype=torch.float64, device=self.device)
        peaks = dev_bytes(n * max_strips * PEAKS_DT.itemsize, self.device)
        _check(lib.lfa_analyze_batch(self._h, images_t.data_ptr(), dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length,
                                     strips.data_ptr(), max_strips, counts.data_ptr(), info.data_ptr(), rois.data_ptr(),
                                     profiles.data_ptr(), stride, peaks.data_ptr(), current_stream_ptr(self.device)))
        return strips, counts, info, rois, profiles, peaks

    def profile_raw(self, images_t, dtype_code, n, H, W, Cc, rois_t, n_rois, inverse, rng, stride, luts_t=None, n_luts=0,
                    roi_lut_t=None):
        torch = _torch()
        profiles = torch.zeros(max(n_rois, 1) * stride, dtype=torch.float64, device=self.device)
        peaks = dev_bytes(max(n_rois, 1) * PEAKS_DT.itemsize, self.device)
        if luts_t is None:
            _check(lib.lfa_profile_batch(self._h, images_t.data_ptr(), dtype_code, n, H, W, Cc, rois_t.data_ptr(), n_rois,
                                         int(bool(inverse)), int(rng), profiles.data_ptr(), stride, peaks.data_ptr(),
                                         current_stream_ptr(self.device)))
        else:
            _check(lib.lfa_profile_lut_batch(self._h, images_t.data_ptr(), dtype_code, n, H, W, Cc, rois_t.data_ptr(), n_rois,
                                             luts_t.data_ptr(), int(n_luts),
                                             roi_lut_t.data_ptr() if roi_lut_t is not None else None,
                                             int(bool(inverse)), int(rng), profiles.data_ptr(), stride, peaks.data_ptr(),
                                             current_stream_ptr(self.device)))
        return profiles, peaks

    # -- numpy-level helpers -----------------------------------------------------------------------------------------
    def _prep(self, images: np.ndarray):
        images = np.asarray(images)
        if images.ndim == 3:            # (n,H,W)
            n, H, W = images.shape
            Cc = 1
        elif images.ndim == 4:
            n, H, W, Cc = images.shape
        else:
            raise ValueError("expected a batch (n,H,W) or (n,H,W,3)")
        return images, n, H, W, Cc, np_dtype_code(images.dtype)

    # -- images the device flagged `ambiguous`: decided again with the host's pow (lfa_set_host_pow) -------------------------
    @staticmethod
    def _ambiguous(inf):
        """indices of the images of an INFO_DT array that need the host-pow run"""
        return np.flatnonzero((inf["ambiguous"] != 0) & ((inf["status"] & 4) == 0))

    def _gather_images(self, images_t, dtype_code, H, W, Cc, idx):
        """the images `idx` of a resident batch as a contiguous batch of their own"""
        torch = _torch()
        ib = H * W * Cc * {LFA_U8: 1, LFA_U16: 2, LFA_F64: 8}[dtype_code]
        flat = images_t.reshape(-1).view(torch.uint8)
        rows = flat[: (flat.numel() // ib) * ib].reshape(-1, ib)
        return rows[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=flat.device)].contiguous()

    def _set_host_pow(self, inf_rows):
        tab = np.stack([host_pow_tables(float(r["gmin"]), float(r["gmax"])) for r in inf_rows])
        self._host_pow_t = to_device(tab, self.device)     # kept alive until the next call has run
        _check(lib.lfa_set_host_pow(self._h, self._host_pow_t.data_ptr(), len(inf_rows)))

    def detect(self, images: np.ndarray, min_maxr, min_maxc, line_length=10, max_strips=64, resolve_ambiguous=True):
        """Batch detect.  Returns (list of STRIP_DT arrays, INFO_DT array).  Images whose LfaImageInfo.ambiguous is non-zero
        (a pixel within 2^-40 of a histogram edge or of the threshold: only there could the device's pow() and the host's
        disagree) are run a second time with tables built by the host's numpy (host_pow_tables); their info rows then have
        status bit 2 set."""
        images, n, H, W, Cc, code = self._prep(images)
        out_s, out_i = [], []
        for b0 in range(0, n, self.max_batch):
            b1 = min(n, b0 + self.max_batch)
            t = to_device(images[b0:b1], self.device)
            ms = max_strips
            while True:
                strips, counts, info = self.detect_raw(t, code, b1 - b0, H, W, Cc, int(min_maxr), int(min_maxc),
                                                       int(line_length), ms)
                inf = from_device(info, INFO_DT, b1 - b0)
                if (inf["status"] & 1).any() and ms < 1024:
                    ms = min(1024, ms * 4)
                    continue
                break
            if (inf["status"] & 3).any():
                raise LfaError(-3, f"detect overflow (status {inf['status'].tolist()}): more than 1024 strips per image")
            cn = from_device(counts, np.int32, b1 - b0).copy()
            st = from_device(strips, STRIP_DT, (b1 - b0) * ms).reshape(b1 - b0, ms).copy()
            inf = inf.copy()
            rows = [st[i, :cn[i]].copy() for i in range(b1 - b0)]
            amb = self._ambiguous(inf) if resolve_ambiguous else []
            if len(amb):
                # pixels within 2^-40 of a histogram edge or of the threshold: those images again, with the host's pow()
                sub = self._gather_images(t, code, H, W, Cc, amb)
                ms2 = ms
                while True:
                    self._set_host_pow(inf[amb])
                    s2, c2, i2 = self.detect_raw(sub, code, len(amb), H, W, Cc, int(min_maxr), int(min_maxc), int(line_length), ms2)
                    i2 = from_device(i2, INFO_DT, len(amb)).copy()
                    if (i2["status"] & 1).any() and ms2 < 1024:
                        ms2 = min(1024, ms2 * 4)
                        continue
                    break
                if (i2["status"] & 3).any() or ((i2["status"] & 4) == 0).any():
                    raise LfaError(-3, f"the host-pow run of an ambiguous image did not go through (status {i2['status'].tolist()})")
                c2 = from_device(c2, np.int32, len(amb))
                s2 = from_device(s2, STRIP_DT, len(amb) * ms2).reshape(len(amb), ms2)
                for k, i in enumerate(amb):
                    rows[int(i)] = s2[k, :c2[k]].copy()
                inf[amb] = i2
            out_s += rows
            out_i.append(inf)
        return out_s, np.concatenate(out_i)

    @staticmethod
    def unpack_analysis(n, max_strips, stride, strips, counts, info, profiles, peaks):
        """Host arrays of one analyze_raw call -> (per image (strips, [profile...], peaks), INFO_DT array); None for the first
        element when an image overflowed max_strips (status bit 0: the caller re-runs with more)."""
        inf = info[: n * INFO_DT.itemsize].view(INFO_DT)
        if (inf["status"] & 2).any():
            raise LfaError(-3, f"analyze overflow (status {inf['status'].tolist()}): run capacity exceeded")
        if (inf["status"] & 1).any():
            return None, inf
        cn = counts[: n * 4].view(np.int32)
        st = strips[: n * max_strips * STRIP_DT.itemsize].view(STRIP_DT).reshape(n, max_strips)
        pk = peaks[: n * max_strips * PEAKS_DT.itemsize].view(PEAKS_DT).reshape(n, max_strips)
        pr = profiles.reshape(n, max_strips, stride)
        res = []
        for i in range(n):
            k = int(cn[i])
            res.append((st[i, :k].copy(), [pr[i, q, :pk[i, q]["length"]].copy() for q in range(k)], pk[i, :k].copy()))
        return res, inf

    def analyze_device(self, images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, max_strips, stride,
                       resolve_ambiguous=True, host_pow_rows=None):
        """analyze_raw on a resident batch + the same ladder as `detect`: an image with more accepted regions than
        max_strips (status bit 0) sends the batch round again with 4x the slots, up to the ABI's 1024; images flagged
        `ambiguous` go round again with the host's pow() tables (see `detect`).  host_pow_rows: INFO_DT rows (gmin, gmax) of
        the n images -- every run of the ladder then uses tables built from them (lfa_set_host_pow)."""
        ms = int(max_strips)
        while True:
            if host_pow_rows is not None:
                self._set_host_pow(host_pow_rows)
            out = self.analyze_raw(images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, ms, stride)
            strips, counts, info, _rois, profiles, peaks = [t.cpu().numpy() for t in out]
            res, inf = self.unpack_analysis(n, ms, stride, strips, counts, info, profiles, peaks)
            if res is not None:
                inf = inf.copy()
                if resolve_ambiguous and host_pow_rows is None:
                    return self.resolve_ambiguous(images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, ms, stride, res, inf)
                return res, inf
            if ms >= 1024:
                raise LfaError(-3, f"analyze overflow (status {inf['status'].tolist()}): more than 1024 strips per image")
            ms = min(1024, ms * 4)

    def resolve_ambiguous(self, images_t, dtype_code, n, H, W, Cc, min_maxr, min_maxc, line_length, max_strips, stride, res, inf):
        """(res, inf) of an analyze call with the images flagged `ambiguous` replaced by their host-pow run."""
        amb = self._ambiguous(inf)
        if not len(amb):
            return res, inf
        sub = self._gather_images(images_t, dtype_code, H, W, Cc, amb)
        r2, i2 = self.analyze_device(sub, dtype_code, len(amb), H, W, Cc, min_maxr, min_maxc, line_length, max_strips, stride,
                                     host_pow_rows=inf[amb].copy())
        if ((i2["status"] & 4) == 0).any():
            raise LfaError(-3, "the host-pow run of an ambiguous image did not go through")
        res = list(res)
        for k, i in enumerate(amb):
            res[int(i)] = r2[k]
        inf[amb] = i2
        return res, inf

    def analyze(self, images: np.ndarray, min_maxr, min_maxc, line_length=10, max_strips=64):
        """detect + image_analysis(strip, False) per strip.  Returns per image: (strips, [profile...], PEAKS_DT array)."""
        images, n, H, W, Cc, code = self._prep(images)
        stride = max(1, int(H * .6) + 1)
        res = []
        infos = []
        for b0 in range(0, n, self.max_batch):
            b1 = min(n, b0 + self.max_batch)
            t = to_device(images[b0:b1], self.device)
            r, inf = self.analyze_device(t, code, b1 - b0, H, W, Cc, int(min_maxr), int(min_maxc), int(line_length),
                                         max_strips, stride)
            res += r
            infos.append(inf)
        return res, np.concatenate(infos)

    def profiles(self, images: np.ndarray, rois: np.ndarray, inverse=False, rng=15, luts=None, roi_lut=None, strict=True):
        """Row-mean profiles + peaks of arbitrary ROIs (ROI_DT array) over a batch resident in one upload.

        luts: optional (n_luts, 256) uint8 / (n_luts, 65536) uint16 point-operation tables (`point_op_lut`), applied to
        every pixel before the row mean; roi_lut: table index per ROI (default: table 0).

        strict (default): a ROI outside its image, an inverted ROI or a table index out of range raises ValueError before
        anything is launched.  strict=False hands them to the device as they are: the C ABI marks each bad ROI with
        n_peaks = -1 (include/lfa.h), its profile comes back empty, the other ROIs are unaffected."""
        images, n, H, W, Cc, code = self._prep(images)
        if n > self.max_batch:
            raise ValueError("batch larger than the engine's max_batch")
        rois = np.ascontiguousarray(rois, dtype=ROI_DT)
        n_rois = len(rois)
        if n_rois == 0:
            return [], np.zeros(0, PEAKS_DT)
        bad = ((rois["image"] < 0) | (rois["image"] >= n) | (rois["r0"] < 0) | (rois["c0"] < 0) | (rois["r1"] > H) |
               (rois["c1"] > W * Cc) | (rois["r1"] < rois["r0"]) | (rois["c1"] < rois["c0"]))
        if strict and bad.any():
            raise ValueError(f"ROI {int(np.flatnonzero(bad)[0])} lies outside its image or is inverted: {rois[bad][0]}")
        stride = max(1, int((rois["r1"] - rois["r0"]).max()))
        t = to_device(images, self.device)
        rt = to_device(rois, self.device)
        lt = rlt = None
        n_luts = 0
        if luts is not None:
            luts = np.ascontiguousarray(luts)
            entries = 256 if images.dtype == np.uint8 else 65536
            if luts.dtype != images.dtype or luts.ndim not in (1, 2) or luts.shape[-1] != entries:
                raise ValueError(f"luts must be {images.dtype} tables of {entries} entries")
            n_luts = 1 if luts.ndim == 1 else luts.shape[0]
            lt = to_device(luts, self.device)
            if roi_lut is not None:
                roi_lut = np.ascontiguousarray(roi_lut, dtype=np.int32)
                if roi_lut.shape != (n_rois,):
                    raise ValueError("roi_lut needs one table index per ROI")
                if strict and ((roi_lut < 0) | (roi_lut >= n_luts)).any():
                    raise ValueError(f"roi_lut holds a table index outside [0, {n_luts})")
                rlt = to_device(roi_lut, self.device)
        profiles, peaks = self.profile_raw(t, code, n, H, W, Cc, rt, n_rois, inverse, rng, stride, lt, n_luts, rlt)
        pk = from_device(peaks, PEAKS_DT, n_rois)
        if strict and (pk["n_peaks"] < 0).any():           # the kernel's own bad-ROI mark (lfa.h): no profile was written
            raise LfaError(-1, f"ROI {int(np.flatnonzero(pk['n_peaks'] < 0)[0])} was rejected by the device")
        pr = profiles.cpu().numpy().reshape(n_rois, stride)
        return [pr[i, :(max(0, pk[i]["length"]) if pk[i]["n_peaks"] >= 0 else 0)].copy() for i in range(n_rois)], pk.copy()

    def profiles_any(self, regions: np.ndarray, rois: np.ndarray, inverse=False, rng=15):
        """Engine.profiles for regions of the other element types np.nanmean takes (lfa_profile_any_batch): (n,H,W[,C])
        int8 / int16 / int32 / uint32 / float64."""
        torch = _torch()
        regions = np.asarray(regions)
        if regions.ndim == 3:
            n, H, W = regions.shape
            Cc = 1
        else:
            n, H, W, Cc = regions.shape
        code = ELEM_CODES[regions.dtype.name]
        rois = np.ascontiguousarray(rois, dtype=ROI_DT)
        n_rois = len(rois)
        stride = max(1, int((rois["r1"] - rois["r0"]).max()))
        t, rt = to_device(regions, self.device), to_device(rois, self.device)
        profiles = torch.zeros(n_rois * stride, dtype=torch.float64, device=self.device)
        peaks = dev_bytes(n_rois * PEAKS_DT.itemsize, self.device)
        _check(lib.lfa_profile_any_batch(self._h, t.data_ptr(), code, n, H, W, Cc, rt.data_ptr(), n_rois, int(bool(inverse)), int(rng),
                                         profiles.data_ptr(), stride, peaks.data_ptr(), current_stream_ptr(self.device)))
        pk = from_device(peaks, PEAKS_DT, n_rois)
        if (pk["n_peaks"] < 0).any():
            raise LfaError(-1, f"ROI {int(np.flatnonzero(pk['n_peaks'] < 0)[0])} was rejected by the device")
        pr = profiles.cpu().numpy().reshape(n_rois, stride)
        return [pr[i, :max(0, pk[i]["length"])].copy() for i in range(n_rois)], pk.copy()

    def peaks(self, profiles_list, inverse=False, window_size=25, two_peaks=False, rng=15):
        n = len(profiles_list)
        if n == 0:
            return np.zeros(0, PEAKS_DT)
        lens = np.array([len(p) for p in profiles_list], dtype=np.int32)
        stride = max(1, int(lens.max()))
        buf = np.zeros((n, stride), dtype=np.float64)
        for i, p in enumerate(profiles_list):
            buf[i, :len(p)] = np.asarray(p, dtype=np.float64)
        bt = to_device(buf, self.device)
        lt = to_device(lens, self.device)
        out = dev_bytes(n * PEAKS_DT.itemsize, self.device)
        _check(lib.lfa_peaks_batch(self._h, bt.data_ptr(), lt.data_ptr(), n, stride, int(bool(inverse)), int(window_size),
                                   int(bool(two_peaks)), int(rng), out.data_ptr(), current_stream_ptr(self.device)))
        return from_device(out, PEAKS_DT, n).copy()

    def gaussian(self, images: np.ndarray, sigma=1.0, out_f32=False, host_weights=True):
        torch = _torch()
        images, n, H, W, Cc, code = self._prep(images)
        if n > self.max_batch:
            raise ValueError("batch larger than the engine's max_batch")
        t = to_device(images, self.device)
        out = torch.empty(n * H * W * Cc, dtype=torch.float32 if out_f32 else torch.float64, device=self.device)
        wptr, radius = None, 0
        if host_weights:
            # scipy.ndimage._filters._gaussian_kernel1d (order 0) with this host's numpy exp()
            radius = int(4.0 * float(sigma) + 0.5)
            x = np.arange(-radius, radius + 1)
            phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            w = np.ascontiguousarray(phi / phi.sum(), dtype=np.float64)
            wptr = w.ctypes.data
        _check(lib.lfa_gaussian_batch(self._h, t.data_ptr(), code, n, H, W, Cc, float(sigma), wptr, radius, out.data_ptr(),
                                      int(bool(out_f32)), current_stream_ptr(self.device)))
        shape = (n, H, W) if Cc == 1 else (n, H, W, Cc)
        return out.cpu().numpy().reshape(shape)

    @staticmethod
    def scipy_gaussian_taps(sigma, radius):
        """scipy.ndimage._filters._gaussian_kernel1d (order 0) with this host's numpy exp()"""
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        return np.ascontiguousarray(phi / phi.sum(), dtype=np.float64)

    def detect_lines(self, images: np.ndarray, max_boxes=256, host_weights=True):
        """detect_lines (detection.py:156-201) of every image of a (n,H,W) uint8 / uint16 batch.
        Returns (list of LINEBOX_DT arrays in the reference's order, LINESINFO_DT array)."""
        torch = _torch()
        images = np.asarray(images)
        if images.ndim != 3:
            raise ValueError("expected a batch (n,H,W) of 2-D images")
        n, H, W = images.shape
        code = np_dtype_code(images.dtype)
        w9 = self.scipy_gaussian_taps(1.0, 4) if host_weights else None
        w3 = self.scipy_gaussian_taps((3 - 1) / 6.0, 1) if host_weights else None
        out_b, out_i = [], []
        for b0 in range(0, n, self.max_batch):
            m = min(n, b0 + self.max_batch) - b0
            t = to_device(images[b0:b0 + m], self.device)
            cap = int(max_boxes)
            while True:
                boxes = dev_bytes(m * cap * LINEBOX_DT.itemsize, self.device)
                counts = torch.zeros(m, dtype=torch.int32, device=self.device)
                info = dev_bytes(m * LINESINFO_DT.itemsize, self.device)
                _check(lib.lfa_detect_lines_batch(self._h, t.data_ptr(), code, m, H, W, w9.ctypes.data if host_weights else None,
                                                  w3.ctypes.data if host_weights else None, boxes.data_ptr(), cap,
                                                  counts.data_ptr(), info.data_ptr(), current_stream_ptr(self.device)))
                inf = from_device(info, LINESINFO_DT, m)
                if (inf["status"] & 1).any() and cap < 65536:
                    cap = min(65536, cap * 8)
                    continue
                break
            if (inf["status"] & 1).any():
                raise LfaError(-3, "detect_lines: more than 65536 boxes per image")
            cn = from_device(counts, np.int32, m)
            bx = from_device(boxes, LINEBOX_DT, m * cap).reshape(m, cap)
            out_b += [bx[i, :cn[i]].copy() for i in range(m)]
            out_i.append(inf.copy())
        return out_b, np.concatenate(out_i)

    def lines_debug(self, image_index, what, H, W):
        """Intermediate `what` (LDBG_*) of image `image_index` of the last detect_lines call."""
        torch = _torch()
        dt, count = {LDBG_GAUSS: (torch.float64, H * W), LDBG_U16: (torch.int16, H * W), LDBG_LEVELS: (torch.int16, H * W),
                     LDBG_ROOTS: (torch.int32, H * W), LDBG_LVAL: (torch.float64, 16384)}[what]
        out = torch.empty(count, dtype=dt, device=self.device)
        _check(lib.lfa_lines_debug_fetch(self._h, image_index, what, out.data_ptr(), out.numel() * out.element_size(),
                                         current_stream_ptr(self.device)))
        a = out.cpu().numpy()
        if dt == torch.int16:
            a = a.view(np.uint16)
        return a.reshape(H, W) if count == H * W else a

    def debug(self, image_index, what, H, W):
        torch = _torch()
        if what in (DBG_MASK, DBG_OPENED):
            out = torch.empty(H * W, dtype=torch.uint8, device=self.device)
            nb = H * W
        elif what == DBG_LABELS:
            out = torch.empty(H * W, dtype=torch.int32, device=self.device)
            nb = H * W * 4
        else:
            raise ValueError(what)
        _check(lib.lfa_debug_fetch(self._h, image_index, what, out.data_ptr(), nb, current_stream_ptr(self.device)))
        return out.cpu().numpy().reshape(H, W)

    def debug_table(self, image_index, k):
        torch = _torch()
        out = torch.zeros(max(k, 1) * 5, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, image_index, DBG_TABLE, out.data_ptr(), max(k, 1) * 20,
                                   current_stream_ptr(self.device)))
        return out.cpu().numpy().reshape(-1, 5)[:k]

    def debug_hist(self, image_index, nbins=256):
        """Histogram counts of image `image_index` of the last call (LFA_DBG_HIST): 256 bins for colour / uint8 images."""
        torch = _torch()
        out = torch.zeros(nbins, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, image_index, DBG_HIST, out.data_ptr(), nbins * 4, current_stream_ptr(self.device)))
        return out.cpu().numpy().view(np.uint32)

    def debug_used_plane(self, image_index):
        """True if the mask of image `image_index` of the last call was decided from the class plane (LFA_DBG_PLANE)."""
        torch = _torch()
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, image_index, DBG_PLANE, out.data_ptr(), 4, current_stream_ptr(self.device)))
        return bool(out.cpu().numpy()[0])

    def debug_pred_path(self, image_index):
        """True if the mask of image `image_index` of the last call came from the predicted-threshold bit plane (LFA_DBG_PLANE == 2)."""
        torch = _torch()
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, image_index, DBG_PLANE, out.data_ptr(), 4, current_stream_ptr(self.device)))
        return int(out.cpu().numpy()[0]) == 2

    def debug_grey_path(self, image_index):
        """True if image `image_index` of the last call went through the equal-channel path (LFA_DBG_GREY)."""
        torch = _torch()
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, image_index, DBG_GREY, out.data_ptr(), 4, current_stream_ptr(self.device)))
        return bool(out.cpu().numpy()[0])

    def debug_one_pass(self):
        """True if the last call took the one-pass uint16 histogram (LFA_DBG_ONEPASS)."""
        torch = _torch()
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        _check(lib.lfa_debug_fetch(self._h, 0, DBG_ONEPASS, out.data_ptr(), 4, current_stream_ptr(self.device)))
        return bool(out.cpu().numpy()[0])

    def set_debug_flags(self, flags):
        _check(lib.lfa_set_debug_flags(self._h, int(flags)))

    def timing(self, kernel_filter):
        """"*": time every kernel, "<name>": that kernel only, None: off (include/lfa.h lfa_timing_enable)."""
        _check(lib.lfa_timing_enable(self._h, kernel_filter.encode() if kernel_filter else None))

    def timing_read(self, max_entries=64):
        names = C.create_string_buffer(32 * max_entries)
        ms = (C.c_double * max_entries)()
        cnt = (C.c_int32 * max_entries)()
        n = C.c_int32(0)
        _check(lib.lfa_timing_read(self._h, max_entries, names, ms, cnt, C.byref(n)))
        out = {}
        for k in range(n.value):
            out[names.raw[32 * k:32 * k + 32].split(b"\0")[0].decode()] = (ms[k], cnt[k])
        return out

    def sync(self):
        _check(lib.lfa_sync(self._h, current_stream_ptr(self.device)))


_ENGINES = {}


def shared_engine(device=0, H=0, W=0):
    """Process-wide engine per device, grown when a larger image arrives."""
    e = _ENGINES.get(device)
    if e is None or H > e._max_h or W > e._max_w:
        if e is not None:
            e.close()
        mh, mw = max(H, 2048, getattr(e, "_max_h", 0)), max(W, 2048, getattr(e, "_max_w", 0))
        e = Engine(device=device, max_h=mh, max_w=mw, max_c=3, max_batch=4)
        e._max_h, e._max_w = mh, mw
        _ENGINES[device] = e
    return e
