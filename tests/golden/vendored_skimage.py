"""Run the reference's vendored scikit-image 0.20.0 Python sources in the build container (fixture generation only).

The reference commits its venv (`/root/reference/image-analysis/lib/python3.10/site-packages`, SP below).  The skimage in
it cannot be imported as a package here: its compiled parts are darwin-arm64 Mach-O files and its package __init__s pull
them in.  But every function on the LFA path that determines an output is plain Python over numpy / scipy.ndimage, so
this module compiles exactly those source files FROM WHERE THEY LIE, at run time and in memory (nothing of them is
written anywhere), one namespace per file, with their `import` statements replaced by the names those statements would
have bound:

    SP/skimage/_shared/_warnings.py, _shared/utils.py        warn, decorators, _supported_float_type
    SP/skimage/util/dtype.py                                  dtype_range, dtype_limits, img_as_float (_convert)
    SP/skimage/util/_invert.py                                invert
    SP/skimage/exposure/exposure.py                           histogram, adjust_gamma, rescale_intensity
    SP/skimage/color/colorconv.py                             rgb2gray
    SP/skimage/_shared/filters.py                             gaussian
    SP/skimage/filters/thresholding.py (three functions)      _validate_image_histogram, threshold_otsu, threshold_local
    SP/skimage/morphology/misc.py, footprints.py (rectangle), binary.py      binary_erosion / binary_dilation / rectangle
    SP/skimage/measure/_label.py, _regionprops.py             label, regionprops (bbox, area)
    SP/skimage/exposure/_adapthist.py                         equalize_adapthist (detect_lines only)

Three names that numpy 2 removed are given back with their numpy-1.24 meaning (the vendored numpy is 1.24.3):
`np.float_` (= float64), `np.product` (= prod), `np.obj2sctype` (scalar type of a dtype-like).  Names bound by imports
of Cython modules that exist only as darwin binaries (`clabel`, `_moments`, `find_contours`, `marching_cubes`,
`euler_number`, `perimeter`, `draw`, multi-Otsu helpers, `integral_image`) are bound to None: nothing on the path calls them
(label() of a bool image goes to scipy.ndimage.label, _label.py:116-118; bbox and area need no moments), and a call
would fail loudly.

The arithmetic underneath is this container's numpy 2.2.6 / scipy 1.15.3, not the vendored 1.24.3 / 1.10.1.

This file needs /root/reference and is imported by tests/golden/make_golden.py only; no test imports it.
"""
import ast
import collections
import collections.abc
import contextlib
import functools
import inspect
import itertools
import math
import numbers
import os
import re
import sys
import types
import warnings

import numpy
import scipy.linalg
import scipy.ndimage
import scipy.spatial.distance

SP = "/root/reference/image-analysis/lib/python3.10/site-packages"


class _Numpy124(types.ModuleType):
    """numpy with the three names the vendored sources use and numpy 2 dropped."""

    def __init__(self):
        super().__init__("numpy")
        self.float_ = numpy.float64
        self.product = numpy.prod

    @staticmethod
    def obj2sctype(rep, default=None):
        if isinstance(rep, type) and issubclass(rep, numpy.generic):
            return rep
        try:
            return numpy.dtype(rep).type
        except Exception:
            return default

    def __getattr__(self, name):
        return getattr(numpy, name)


np124 = _Numpy124()


def load(relpath, names, only=None):
    """Compile SP/skimage/<relpath> without its import statements and run it in a fresh namespace pre-seeded with
    `names`.  only: keep just these top-level functions / classes / assignments (for files whose module body needs
    compiled modules)."""
    path = os.path.join(SP, "skimage", relpath)
    tree = ast.parse(open(path).read(), filename=path)
    body = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            continue
        if only is not None:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                if node.name not in only:
                    continue
            elif isinstance(node, ast.Assign):
                if not any(isinstance(t, ast.Name) and t.id in only for t in node.targets):
                    continue
            else:
                continue
        body.append(node)
    tree.body = body

    class DropRelativeImports(ast.NodeTransformer):          # function-local `from ..x import y` (e.g. _label.py:10)
        def visit_ImportFrom(self, node):
            return ast.Pass() if node.level > 0 else node

    tree = ast.fix_missing_locations(DropRelativeImports().visit(tree))
    ns = {"__name__": "vendored_skimage." + relpath.replace("/", ".")[:-3], "np": np124, "ndi": scipy.ndimage}
    ns.update(names)
    exec(compile(tree, path, "exec"), ns)
    return types.SimpleNamespace(**{k: v for k, v in ns.items() if not k.startswith("__")})


@functools.lru_cache(maxsize=None)
def skimage():
    """Namespace tree shaped like the parts of `skimage` that detection.py touches on the path."""
    std = dict(inspect=inspect, functools=functools, sys=sys, warnings=warnings, Iterable=collections.abc.Iterable,
               contextmanager=contextlib.contextmanager, re=re, os=os, itertools=itertools, math=math,
               OrderedDict=collections.OrderedDict, Sequence=collections.abc.Sequence, Integral=numbers.Integral,
               wraps=functools.wraps, atan2=math.atan2, PI=math.pi, sqrt=math.sqrt)
    w = load("_shared/_warnings.py", std)
    dtype = load("util/dtype.py", dict(std, warn=warnings.warn))
    utils = load("_shared/utils.py", dict(std, all_warnings=w.all_warnings, warn=w.warn,
                                          img_as_float=dtype.img_as_float))        # function-local import at utils.py:621
    inv = load("util/_invert.py", dict(dtype_limits=dtype.dtype_limits))
    exposure = load("exposure/exposure.py", dict(dtype_range=dtype.dtype_range, dtype_limits=dtype.dtype_limits, utils=utils))
    color = load("color/colorconv.py", dict(warn=warnings.warn, linalg=scipy.linalg, dtype=dtype, dtype_limits=dtype.dtype_limits,
                                            _supported_float_type=utils._supported_float_type,
                                            channel_as_last_axis=utils.channel_as_last_axis, identity=utils.identity,
                                            reshape_nd=utils.reshape_nd, slice_at_axis=utils.slice_at_axis))
    sfilters = load("_shared/filters.py", dict(std, _supported_float_type=utils._supported_float_type,
                                               convert_to_float=utils.convert_to_float, warn=utils.warn))
    thresholding = load("filters/thresholding.py",
                        dict(std, gaussian=sfilters.gaussian, _supported_float_type=utils._supported_float_type, warn=utils.warn,
                             histogram=exposure.histogram, dtype_limits=dtype.dtype_limits, integral_image=None,
                             _correlate_sparse=None, _validate_window_size=None),
                        only={"_validate_image_histogram", "threshold_otsu", "threshold_local"})
    misc = load("morphology/misc.py", dict(std, warn=utils.warn))
    footprints = load("morphology/footprints.py", dict(std, draw=None, morphology=None, deprecate_kwarg=utils.deprecate_kwarg),
                      only={"_footprint_is_sequence", "_shape_from_sequence", "rectangle", "square"})
    binary = load("morphology/binary.py", dict(std, _footprint_is_sequence=footprints._footprint_is_sequence,
                                               default_footprint=misc.default_footprint))
    mutil = load("morphology/_util.py", {})
    lab = load("measure/_label.py", dict(ndimage=scipy.ndimage, clabel=None, _resolve_neighborhood=mutil._resolve_neighborhood))
    rp = load("measure/_regionprops.py", dict(std, warn=warnings.warn, pdist=scipy.spatial.distance.pdist, _moments=None,
                                              find_contours=None, marching_cubes=None, euler_number=None, perimeter=None,
                                              perimeter_crofton=None))
    # equalize_adapthist: its decorator adapt_rgb(hsv_value) passes a non-colour image straight through (color/adapt_rgb.py)
    adapt = load("exposure/_adapthist.py", dict(std, numbers=numbers, _supported_float_type=utils._supported_float_type,
                                                adapt_rgb=lambda _f: (lambda fn: fn), hsv_value=None,
                                                rescale_intensity=exposure.rescale_intensity, img_as_uint=dtype.img_as_uint))
    S = types.SimpleNamespace
    return S(exposure=S(adjust_gamma=exposure.adjust_gamma, histogram=exposure.histogram,
                        rescale_intensity=exposure.rescale_intensity, equalize_adapthist=adapt.equalize_adapthist,
                        _clahe=adapt._clahe),
             color=S(rgb2gray=color.rgb2gray),
             filters=S(threshold_otsu=thresholding.threshold_otsu, threshold_local=thresholding.threshold_local,
                       gaussian=sfilters.gaussian),
             util=S(invert=inv.invert, img_as_float=dtype.img_as_float),
             morphology=S(binary_erosion=binary.binary_erosion, binary_dilation=binary.binary_dilation,
                          rectangle=footprints.rectangle),
             measure=S(label=lab.label, regionprops=rp.regionprops))


def reference_detect(min_maxr, min_maxc, angle_threshold=10):
    """detect_lateral_flow_tests as LFA-Image_Analysis/detection.py:31-153 defines it, bound to the vendored skimage.

    Its three dead-compute calls -- sk.feature.canny, sk.transform.hough_line, sk.transform.hough_line_peaks
    (detection.py:51-57; Cython, darwin only) -- are replaced by calls that return no lines: their only consumer is
    straighten_region (:97-115), which writes into a copy that is never read again (:69, :84, :86), and with no angles
    it does not even do that (mean of an empty list is NaN, `NaN > angle_threshold` is False).  Everything that
    determines the return value runs the reference's own lines.  Returns (function, namespace); the namespace also
    records every intermediate the function passes through the skimage calls (`trace`)."""
    sk = skimage()
    trace = {}

    def rec(name, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            trace[name] = out
            return out
        return wrapped

    S = types.SimpleNamespace
    empty = numpy.zeros(0)
    sk_ns = S(exposure=S(adjust_gamma=rec("gamma", sk.exposure.adjust_gamma)),
              color=S(rgb2gray=rec("gray", sk.color.rgb2gray)),
              feature=S(canny=lambda image: image),
              transform=S(hough_line=lambda edges: (empty, empty, empty),
                          hough_line_peaks=lambda h, theta, d: (empty, empty, empty), rotate=None))
    ns = {"np": numpy, "sk": sk_ns, "filters": S(threshold_otsu=rec("threshold", sk.filters.threshold_otsu)),
          "morphology": S(binary_erosion=rec("eroded", sk.morphology.binary_erosion),
                          binary_dilation=rec("dilated", sk.morphology.binary_dilation), rectangle=sk.morphology.rectangle),
          "label": rec("labels", sk.measure.label), "regionprops": rec("regions", sk.measure.regionprops),
          "min_maxr": int(min_maxr), "min_maxc": int(min_maxc), "angle_threshold": angle_threshold}
    src = open("/root/reference/LFA-Image_Analysis/detection.py").read().split("\n")
    for lo, hi in ((31, 94), (97, 115), (118, 127), (138, 153)):              # 1-based inclusive line ranges
        exec(compile("\n" * (lo - 1) + "\n".join(src[lo - 1:hi]), "/root/reference/LFA-Image_Analysis/detection.py", "exec"), ns)
    ns["trace"] = trace
    return ns["detect_lateral_flow_tests"], ns


class StopAtLabel(Exception):
    pass


def reference_detect_lines():
    """detect_lines as LFA-Image_Analysis/detection.py:156-201 defines it, bound to the vendored skimage, up to its call of
    sk.measure.label (:184): label() of a non-boolean image is Cython (`_ccomp`, a darwin binary), so the run stops there
    and hands back every intermediate the function passed through the skimage / numpy calls.  The matplotlib calls of
    :162-164 (a figure that shows the input) are bound to no-ops.  Returns run(image) -> trace dict."""
    sk = skimage()
    S = types.SimpleNamespace

    def run(image):
        trace = {}

        def rec(name, fn):
            def wrapped(*a, **k):
                out = fn(*a, **k)
                trace[name] = out
                return out
            return wrapped

        def stop(img):
            trace["label_input"] = img
            raise StopAtLabel()

        class NP(types.ModuleType):
            def __getattr__(self, name):
                return getattr(numpy, name)
        npx = NP("numpy")
        npx.percentile = rec("percentiles", numpy.percentile)
        fig = S(add_subplot=lambda *a, **k: None)
        ns = {"np": npx, "plt": S(figure=lambda **k: fig, imshow=lambda *a, **k: None),
              "sk": S(filters=S(gaussian=rec("gauss", sk.filters.gaussian), threshold_local=rec("local", sk.filters.threshold_local)),
                      exposure=S(adjust_gamma=rec("gamma", sk.exposure.adjust_gamma),
                                 equalize_adapthist=rec("enhanced", sk.exposure.equalize_adapthist),
                                 rescale_intensity=rec("rescaled", sk.exposure.rescale_intensity)),
                      measure=S(label=stop)),
              "regionprops": None}
        src = open("/root/reference/LFA-Image_Analysis/detection.py").read().split("\n")
        lo, hi = 156, 201
        exec(compile("\n" * (lo - 1) + "\n".join(src[lo - 1:hi]), "/root/reference/LFA-Image_Analysis/detection.py", "exec"), ns)
        try:
            ns["detect_lines"](image)
        except StopAtLabel:
            pass
        return trace
    return run
