#!/usr/bin/env python3
"""Generate the golden fixtures in this directory.  Runs ONLY in the build container (needs /root/reference).

What it produces (all of it data: inputs + expected outputs, no reference source text):

* excel2_frame.png / excel2_crop.png   the two images the reference's own export `images/excel2.xlsx` embeds
                                        (xl/media/image1.png = the 2048x2048 uint16 frame the GUI had open,
                                        xl/media/image2.png = the strip crop it analysed), copied byte for byte.
* excel2_known_answer.npz               column B of that workbook (the 357-value profile the reference wrote).
* ref_profile_peaks.npz                 inputs + outputs of the reference's numpy-only functions
                                        (detection.py:205-316: image_analysis, inverse_image_analysis,
                                        analyze_peaks) EXECUTED here from the reference's file.
* ref_region_lists.npz                  inputs + outputs of detection.py:118-153 (check_duplicate, remove_outliers).
* ref_skimage_detect.npz                the whole of detect_lateral_flow_tests (detection.py:31-153) EXECUTED here over the
                                        reference's vendored scikit-image 0.20.0 Python sources (vendored_skimage.py: rgb2gray,
                                        adjust_gamma, threshold_otsu + exposure.histogram, binary_erosion / dilation, label,
                                        regionprops), on small colour / uint8 / uint16 inputs and edge cases: inputs, the
                                        threshold's bits, SHA-256 of gray / gamma / mask / opened mask / label image, K, the
                                        bbox + area table, the returned crops' boxes.  `--only-skimage` writes this and the next.
* ref_skimage_vectors.npz               expected values the vendored skimage's OWN test-suite holds for the functions on the
                                        path (gamma 0.5 / 2.0 tables of exposure/tests/test_exposure.py:622-686, the Otsu image
                                        of filters/tests/test_thresholding.py:33-61, :307-314), read out of those files as data;
                                        plus outputs of the vendored gaussian / rescale_intensity / invert / adjust_gamma chain
                                        (analyze_plot's point operations, detection.py:535-539) and of gaussian (:166).
* image1_decoded.npz                    images/image1.jpg decoded with PIL (cv2 is absent here; the decoded
                                        array is stored because decoders may differ by an LSB).
* real_samples.npz                      seven more of the reference's sample inputs (images/), decoded with PIL: three
                                        512x512 grey JPEGs, the one colour photo (BGR), 768x1024 crops of two large
                                        grey JPEGs and of the 16-bit PNG.  Inputs only; `--only-samples` writes just
                                        this file.

The reference module cannot be imported as a whole (cv2 / tifffile / skimage / tkinter / xlsxwriter are not
installed and it opens a Tk window at import, detection.py:787-789); the numpy-only line ranges above are
compiled from its file at run time, in memory, and nothing of them is written anywhere.
"""
import io
import os
import re
import sys
import warnings
import zipfile

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_functions():
    src = open(os.path.join(REF, "LFA-Image_Analysis", "detection.py")).read().split("\n")
    ns = {"np": np, "min_maxr": 0, "min_maxc": 0}
    for lo, hi in ((118, 127), (138, 153), (205, 316)):          # 1-based inclusive line ranges
        exec(compile("\n".join(src[lo - 1:hi]), f"detection.py:{lo}-{hi}", "exec"), ns)
    return ns


def excel2():
    z = zipfile.ZipFile(os.path.join(REF, "images", "excel2.xlsx"))
    frame_png = z.read("xl/media/image1.png")
    crop_png = z.read("xl/media/image2.png")
    open(os.path.join(HERE, "excel2_frame.png"), "wb").write(frame_png)
    open(os.path.join(HERE, "excel2_crop.png"), "wb").write(crop_png)
    # the worksheet and drawing parts of the same workbook, as data: what the reference's export writes for this profile and
    # these pictures (image_analysis_amd.export.write_profile_xlsx must reproduce them byte for byte)
    open(os.path.join(HERE, "excel2_sheet1.xml"), "wb").write(z.read("xl/worksheets/sheet1.xml"))
    open(os.path.join(HERE, "excel2_drawing1.xml"), "wb").write(z.read("xl/drawings/drawing1.xml"))
    sheet = z.read("xl/worksheets/sheet1.xml").decode()
    vals = [float(v) for v in re.findall(r'<c r="B\d+"><v>([^<]+)</v>', sheet)]
    frame = np.array(Image.open(io.BytesIO(frame_png)))
    crop = np.array(Image.open(io.BytesIO(crop_png)))
    return frame, crop, np.array(vals, dtype=np.float64)


def pack_peaks(res):
    """[(s,e,a)[, (s2,e2,a2)]] -> (n, idx int64[2,2], area float64[2])"""
    idx = np.full((2, 2), -1, dtype=np.int64)
    area = np.full(2, np.nan)
    for i, (s, e, a) in enumerate(res):
        idx[i] = (s, e)
        area[i] = a
    return len(res), idx, area


def synth_strip(rng, h, w, c, dtype, n_bands):
    """A strip crop with dark (or bright) transverse bands, the shape image_analysis is meant for."""
    hi = 200 if dtype == np.uint8 else 3000
    base = rng.normal(hi, hi * 0.03, size=(h, w) + ((c,) if c else ()))
    rows = np.arange(h)[:, None]
    for _ in range(n_bands):
        pos = rng.uniform(0.27, 0.58) * h
        depth = rng.uniform(0.1, 0.5) * hi * rng.choice([-1, 1])
        prof = depth * np.exp(-0.5 * ((rows - pos) / rng.uniform(2, 8)) ** 2)
        base += prof if not c else prof[:, :, None]
    return np.clip(base, 0, np.iinfo(dtype).max).astype(dtype)


def two_band_strip(rng, h, w, c, dtype, order, sign):
    """Two transverse bands at the given height fractions: the first of `order` strong, the second weaker."""
    hi = 200 if dtype == np.uint8 else 3000
    base = rng.normal(hi * 0.6, hi * 0.01, size=(h, w) + ((c,) if c else ()))
    rows = np.arange(h)[:, None]
    for pos, depth in zip(order, (0.3, 0.15)):
        prof = sign * depth * hi * np.exp(-0.5 * ((rows - pos * h) / 5.0) ** 2)
        base += prof if not c else prof[:, :, None]
    return np.clip(base, 0, np.iinfo(dtype).max).astype(dtype)


def main():
    warnings.simplefilter("ignore")
    ns = load_reference_functions()
    image_analysis = ns["image_analysis"]
    inverse_image_analysis = ns["inverse_image_analysis"]
    analyze_peaks = ns["analyze_peaks"]
    rng = np.random.default_rng(20240711)

    # ---- excel2 known answer -------------------------------------------------------------------
    frame, crop, colB = excel2()
    prof, peaks = image_analysis(crop, False)
    n, idx, area = pack_peaks(peaks)
    np.savez_compressed(os.path.join(HERE, "excel2_known_answer.npz"),
                        profile_column_B=colB, crop_box=np.array([324, 1344, 806, 857]),
                        ref_profile=prof, ref_n_peaks=n, ref_peak_idx=idx, ref_peak_area=area)
    assert np.array_equal(frame[324:1344, 806:857], crop)
    assert np.array_equal(prof, colB)

    # ---- image_analysis / inverse_image_analysis on synthetic strips ---------------------------
    # `range` takes 1, 3 and 15 for BOTH functions: image_analysis adds `end_index + range` to a right-hand second
    # peak (detection.py:238-239) while inverse_image_analysis slices with `range` (:260-261) and adds the literal
    # `end_index + 1` (:274-275); only range != 1 on the inverse function tells the two apart.
    out = {}
    k = 0
    cases = []
    for dtype in (np.uint8, np.uint16):
        for c in (0, 3):
            for (h, w) in ((1020, 51), (400, 40), (64, 17), (37, 9), (12, 5)):
                for selected in (False, True):
                    cases.append((dtype, c, h, w, selected, int(rng.integers(1, 4)), None))
    # two bands of known order: the stronger one first, a weaker one well to its right (and mirrored), so that the
    # second peak is taken from the right-hand (resp. left-hand) remainder for every `range`
    for dtype in (np.uint8, np.uint16):
        for c in (0, 3):
            for (h, w) in ((1020, 51), (600, 30)):
                for order in ((0.33, 0.52), (0.52, 0.33)):
                    for sign in (-1, 1):
                        cases.append((dtype, c, h, w, False, 2, (order, sign)))
    right_inverse_wide = 0
    for rid, (dtype, c, h, w, selected, nb, two) in enumerate(cases):
        if two is None:
            reg = synth_strip(rng, h, w, c, dtype, nb)
        else:
            reg = two_band_strip(rng, h, w, c, dtype, *two)
        out[f"r{rid}"] = reg                               # every region once; its cases name it (c<k>_rid)
        for inverse in (False, True):
            if inverse and not c:
                continue                                   # detection.py:249 needs three dimensions
            fn = inverse_image_analysis if inverse else image_analysis
            for rg in (None, 1, 3, 15):                    # None: the function's own default
                try:
                    prof, peaks = fn(reg, selected) if rg is None else fn(reg, selected, rg)
                except ValueError:
                    continue                               # empty profile (argmax of empty) -- tested separately
                n, idx, area = pack_peaks(peaks)
                eff = (1 if inverse else 15) if rg is None else rg
                if inverse and eff != 1 and n == 2 and idx[1][0] > idx[0][1]:
                    right_inverse_wide += 1
                out[f"c{k}_rid"] = np.array(rid)
                out[f"c{k}_selected"] = np.array(selected)
                out[f"c{k}_inverse"] = np.array(inverse)
                out[f"c{k}_range"] = np.array(eff)
                out[f"c{k}_profile"] = prof
                out[f"c{k}_n"] = np.array(n)
                out[f"c{k}_idx"] = idx
                out[f"c{k}_area"] = area
                k += 1
    out["n_cases"] = np.array(k)
    assert right_inverse_wide >= 8, right_inverse_wide     # the cases that separate `+ range` from `+ 1`

    # ---- analyze_peaks on raw float profiles (plateaus, ties, monotone, tiny) ------------------
    profs = []
    for n in (1, 2, 3, 5, 8, 9, 26, 51, 52, 100, 129, 357, 714, 1000):
        for kind in range(6):
            x = np.arange(n, dtype=np.float64)
            if kind == 0:
                p = rng.normal(100, 5, n)
            elif kind == 1:
                p = np.round(rng.normal(100, 3, n))                     # many exact ties
            elif kind == 2:
                p = 50 + 40 * np.exp(-0.5 * ((x - n * 0.4) / max(n * 0.05, 1)) ** 2) + rng.normal(0, 0.5, n)
            elif kind == 3:
                p = x * 0.25 + 3                                        # monotone up
            elif kind == 4:
                p = np.full(n, 7.0)                                     # constant
            else:
                p = np.round(rng.uniform(0, 4, n)) / 20 + 119            # /20 grid like the excel2 profile
            profs.append(p)
    pk = {}
    for i, p in enumerate(profs):
        for inverse in (False, True):
            s, e, a = analyze_peaks(p, inverse=inverse)
            pk[f"p{i}_{int(inverse)}"] = np.array([s, e], dtype=np.int64)
            pk[f"a{i}_{int(inverse)}"] = np.array(a, dtype=np.float64)
        pk[f"x{i}"] = p
    pk["n_profiles"] = np.array(len(profs))
    np.savez_compressed(os.path.join(HERE, "ref_profile_peaks.npz"), **out, **pk)

    # ---- check_duplicate / remove_outliers -----------------------------------------------------
    rl = {}
    m = 0
    for trial in range(60):
        mm_r = int(rng.choice([0, 20, 100]))
        mm_c = int(rng.choice([0, 5, 12]))
        ns["min_maxr"], ns["min_maxc"] = mm_r, mm_c
        lst = []
        dup = []
        for _ in range(int(rng.integers(0, 12))):
            a = int(rng.integers(0, 300)); c = int(rng.integers(0, 300))
            b = a + int(rng.integers(1, 150)); d = c + int(rng.integers(1, 40))
            be, de = max(b, a + mm_r), max(d, c + mm_c)
            ok = ns["check_duplicate"](a, c, be, de, lst)
            dup.append(ok)
            if ok:
                lst.append([a, be, c, de])
        kept = ns["remove_outliers"](lst)
        rl[f"t{m}_params"] = np.array([mm_r, mm_c])
        rl[f"t{m}_accepted"] = np.array(lst, dtype=np.int64).reshape(-1, 4)
        rl[f"t{m}_kept"] = np.array(kept, dtype=np.int64).reshape(-1, 4)
        m += 1
    rl["n_trials"] = np.array(m)
    np.savez_compressed(os.path.join(HERE, "ref_region_lists.npz"), **rl)

    # ---- decoded sample image ------------------------------------------------------------------
    im = np.array(Image.open(os.path.join(REF, "images", "image1.jpg")))
    np.savez_compressed(os.path.join(HERE, "image1_decoded.npz"), gray=im)
    write_real_samples()
    write_skimage_fixtures()
    print("golden fixtures written:", sorted(os.listdir(HERE)))


def write_real_samples():
    """More of the reference's sample inputs (images/), decoded with PIL and stored as arrays: the small ones whole, a
    768x1024 crop of three large ones.  They are INPUTS only -- photo-like data on which the Otsu threshold sits on a
    populated grey level (the case the synthetic benchmark images do not have); expected values come from the oracle
    at test time."""
    out = {}
    for name in ("image2.jpeg", "image4.jpg", "image7.jpg"):                    # 512x512 grey JPEGs
        out[name.split(".")[0]] = np.array(Image.open(os.path.join(REF, "images", name)))
    rgb = np.array(Image.open(os.path.join(REF, "images", "test_strip_buggy.jpeg")).convert("RGB"))
    out["test_strip_buggy_bgr"] = np.ascontiguousarray(rgb[:, :, ::-1])        # cv2.imread order
    for name, key in (("WickImage.jpg", "wick_crop"), ("image5.jpg", "image5_crop")):
        a = np.array(Image.open(os.path.join(REF, "images", name)))
        out[key] = np.ascontiguousarray(a[640:1408, 512:1536])
    a16 = np.array(Image.open(os.path.join(REF, "images", "Austin's Image.png"))).astype(np.uint16)
    out["austin_u16_crop"] = np.ascontiguousarray(a16[640:1408, 512:1536])
    np.savez_compressed(os.path.join(HERE, "real_samples.npz"), **out)


def sha(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8).copy()


def crop_box(image, crop):
    """[minr, maxr, minc, maxc] of a view of `image` (as numpy clipped it)."""
    off = crop.__array_interface__["data"][0] - image.__array_interface__["data"][0]
    r, rem = divmod(off, image.strides[0])
    c = rem // image.strides[1]
    return [int(r), int(r + crop.shape[0]), int(c), int(c + crop.shape[1])]


def skimage_detect_cases():
    """(name, image, min_maxr, min_maxc, line_length) -- small inputs over every branch of the detect path."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from image_analysis_amd.synth import synth_rgb, synth_u16
    rng = np.random.default_rng(77)
    gui = lambda im: (int(int(im.shape[1] / 40) * 20), int(im.shape[1] / 40))          # detection.py:457-460
    cases = []
    for seed, (H, W) in ((11, (256, 192)), (12, (256, 192)), (13, (192, 256)), (3, (320, 384))):
        im = synth_rgb(H, W, 6 if W < 300 else 8, seed)
        cases.append((f"synth_rgb_{H}x{W}_s{seed}", im, *gui(im), 10))
    im = synth_rgb(203, 171, 5, 21)                                                   # ragged sides (scalar kernels)
    cases.append(("synth_rgb_ragged", im, *gui(im), 10))
    g1 = np.load(os.path.join(HERE, "image1_decoded.npz"))["gray"]
    cases.append(("image1_gray", g1, *gui(g1), 10))
    cases.append(("image1_bgr", np.ascontiguousarray(np.repeat(g1[:, :, None], 3, axis=2)), *gui(g1), 10))
    rs = np.load(os.path.join(HERE, "real_samples.npz"))
    for key in ("image2", "image4", "image7"):
        cases.append((key + "_bgr", np.ascontiguousarray(np.repeat(rs[key][:, :, None], 3, axis=2)), *gui(rs[key]), 10))
    cases.append(("test_strip_buggy_bgr", rs["test_strip_buggy_bgr"], *gui(rs["test_strip_buggy_bgr"]), 10))
    a = np.ascontiguousarray(rs["austin_u16_crop"][:384, :512])
    cases.append(("austin_u16_384x512", a, *gui(a), 10))
    fr = np.array(Image.open(os.path.join(HERE, "excel2_frame.png")))               # threshold inside the noise: many labels
    cases.append(("excel2_u16_512", np.ascontiguousarray(fr[300:812, 700:1212]), *gui(fr[300:812, 700:1212]), 10))
    wk = np.ascontiguousarray(rs["wick_crop"][200:584, 256:768])
    cases.append(("wick_bgr_384x512", np.ascontiguousarray(np.repeat(wk[:, :, None], 3, axis=2)), *gui(wk), 10))
    u = synth_u16(256, 256, 6, 5)
    cases.append(("synth_u16_256", u, *gui(u), 10))
    cases.append(("synth_u16_256_L5", u, 40, 4, 5))
    cases.append(("constant_u8", np.full((64, 48), 7, np.uint8), 20, 3, 10))
    cases.append(("constant_rgb", np.full((64, 48, 3), 9, np.uint8), 20, 3, 10))
    cases.append(("constant_u16", np.full((40, 40), 1234, np.uint16), 20, 3, 10))
    # strips touching all four borders, a foreground corner (labeled[0,0] != 0), overlapping expanded boxes, boxes that
    # reach beyond the image, vertical runs of 9 / 10 / 11 pixels, one wide outlier
    e = (rng.integers(20, 40, size=(160, 200))).astype(np.uint8)
    bright = lambda r0, r1, c0, c1: e.__setitem__((slice(r0, r1), slice(c0, c1)), rng.integers(200, 240, size=(r1 - r0, c1 - c0)))
    bright(0, 120, 0, 12); bright(30, 160, 30, 42); bright(10, 140, 44, 56); bright(5, 150, 188, 200)
    bright(20, 29, 70, 90); bright(40, 50, 70, 90); bright(60, 71, 70, 90); bright(90, 150, 100, 170)
    bright(0, 4, 120, 140); bright(156, 160, 120, 140)
    cases.append(("edges_u8", e.copy(), 100, 5, 10))
    cases.append(("edges_u8_mm0", e.copy(), 0, 0, 10))
    cases.append(("edges_rgb", np.ascontiguousarray(np.stack([e, np.clip(e.astype(int) + 2, 0, 255).astype(np.uint8), e], axis=2)), 100, 5, 10))
    cases.append(("edges_u16", (e.astype(np.uint16) * 13), 100, 5, 10))
    cases.append(("edges_u8_L3", e.copy(), 30, 5, 3))
    # uint16 colour images (16-bit RGB TIFFs): a foreground corner and borders, synthetic strips with 16-bit noise, a constant image
    e16 = np.stack([e.astype(np.uint16) * 257, np.clip(e.astype(int) + 2, 0, 255).astype(np.uint16) * 257 + 100, e.astype(np.uint16) * 250], axis=2)
    cases.append(("edges_rgb16", np.ascontiguousarray(e16), 100, 5, 10))
    c8 = synth_rgb(192, 256, 4, 31)
    c16 = (c8.astype(np.uint16) << 8) | rng.integers(0, 256, size=c8.shape).astype(np.uint16)
    cases.append(("synth_rgb16_192x256", np.ascontiguousarray(c16), *gui(c16[:, :, 0]), 10))
    cases.append(("constant_rgb16", np.full((40, 44, 3), 3000, np.uint16), 20, 3, 10))
    # float64 images: what rgb2gray / img_as_float hand on (2-D and colour), values above 1, a constant image
    cf = c8 / 255.0 + rng.random(c8.shape) * 1e-3
    cases.append(("synth_rgb_f64_192x256", np.ascontiguousarray(cf), *gui(c8[:, :, 0]), 10))
    gf = np.ascontiguousarray(cf @ np.array([0.2125, 0.7154, 0.0721]))
    cases.append(("synth_gray_f64_192x256", gf, *gui(c8[:, :, 0]), 10))
    cases.append(("synth_gray_f64_times_3p7", np.ascontiguousarray(c8[:, :, 1] * 3.7), *gui(c8[:, :, 0]), 10))
    cases.append(("u16_as_float_256", synth_u16(256, 256, 6, 5) / 65535.0, 40, 4, 5))
    cases.append(("constant_f64", np.full((40, 44), 0.25), 20, 3, 10))
    # ... with a foreground corner: the region test compares float box means (numpy's summation order)
    # (label(0,0) is 1 here, so only boxes whose mean exceeds 1 pass: values well above 1, and values scattered around 1)
    ef = e * 0.731 + rng.random(e.shape) * 1e-2
    cases.append(("edges_f64", np.ascontiguousarray(ef), 100, 5, 10))
    cases.append(("edges_f64_mm0", np.ascontiguousarray(ef), 0, 0, 10))
    cases.append(("edges_rgb_f64", np.ascontiguousarray(np.stack([ef, ef * 0.98 + 0.004, ef * 1.01], axis=2)), 100, 5, 10))
    en = e / 215.0 + rng.random(e.shape) * 1e-2                               # bright areas ~ 0.93 .. 1.12
    cases.append(("edges_f64_near_one", np.ascontiguousarray(en), 20, 3, 10))
    return cases


def write_skimage_fixtures():
    import vendored_skimage as V
    warnings.simplefilter("ignore")
    out = {}
    names = []
    for name, im, mm_r, mm_c, L in skimage_detect_cases():
        fn, ns = V.reference_detect(mm_r, mm_c)
        crops = fn(im, L)
        t = ns["trace"]
        names.append(name)
        out[f"{name}__image"] = im
        out[f"{name}__params"] = np.array([mm_r, mm_c, L], dtype=np.int64)
        out[f"{name}__threshold"] = np.asarray(t["threshold"])               # int for integer images, float64 for colour
        if im.ndim == 3:
            out[f"{name}__gray_sha"] = sha(t["gray"])
        out[f"{name}__gamma_sha"] = sha(t["gamma"])
        out[f"{name}__mask_sha"] = sha(np.asarray(t["gamma"] > t["threshold"], dtype=np.uint8))
        out[f"{name}__opened_sha"] = sha(np.asarray(t["dilated"], dtype=np.uint8))
        labels = np.asarray(t["labels"])
        out[f"{name}__labels_sha"] = sha(labels.astype(np.int32))
        out[f"{name}__n_labels"] = np.array(int(labels.max()), dtype=np.int64)
        out[f"{name}__label00"] = np.array(int(labels[0, 0]), dtype=np.int64)
        out[f"{name}__table"] = np.array([[*r.bbox, int(r.area)] for r in t["regions"]], dtype=np.int64).reshape(-1, 5)
        out[f"{name}__boxes"] = np.array([crop_box(im, c) for c in crops], dtype=np.int64).reshape(-1, 4)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_skimage_detect.npz"), **out)

    # ---- the excel2 frame (stored as png already): the reference's own detect on its own frame -------------------------
    frame = np.array(Image.open(os.path.join(HERE, "excel2_frame.png")))
    fn, ns = V.reference_detect(1020, 51)
    crops = fn(frame)
    t = ns["trace"]
    vec = {"excel2_threshold": np.asarray(t["threshold"]), "excel2_n_labels": np.array(int(np.asarray(t["labels"]).max())),
           "excel2_boxes": np.array([crop_box(frame, c) for c in crops], dtype=np.int64),
           "excel2_mask_sha": sha(np.asarray(t["gamma"] > t["threshold"], dtype=np.uint8)),
           "excel2_labels_sha": sha(np.asarray(t["labels"]).astype(np.int32))}

    # ---- expected values held by the vendored skimage's own tests, read out as data -------------------------------------
    import ast
    sp = os.path.join(V.SP, "skimage")

    def literal_arrays(path, func):
        tree = ast.parse(open(path).read())
        found = []
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name == func:
                for sub in ast.walk(node):
                    if isinstance(sub, ast.Call) and getattr(sub.func, "attr", "") == "array" and sub.args and isinstance(sub.args[0], ast.List):
                        found.append(np.array(ast.literal_eval(sub.args[0])))
        return found
    te = os.path.join(sp, "exposure", "tests", "test_exposure.py")
    vec["gamma_half_table"] = literal_arrays(te, "test_adjust_gamma_less_one")[0].astype(np.uint8)        # :626-653
    vec["gamma_two_table"] = literal_arrays(te, "test_adjust_gamma_greater_one")[0].astype(np.uint8)     # :664-683
    tt = os.path.join(sp, "filters", "tests", "test_thresholding.py")
    vec["otsu_simple_image"] = literal_arrays(tt, "setup_method")[0].astype(np.int64)                    # :34-38; == 2, -2: == 0, float: [2,3)
    assert vec["gamma_half_table"].size == 256 and vec["gamma_two_table"].size == 256 and vec["otsu_simple_image"].shape == (5, 5)

    # ---- analyze_plot point operations and gaussian through the vendored sources ----------------------------------------
    sk = V.skimage()
    rng = np.random.default_rng(5)
    k = 0
    for dt in (np.uint8, np.uint16):
        top = np.iinfo(dt).max
        allv = np.arange(top + 1, dtype=dt).reshape(-1, 256)          # 2-D: every value of the dtype once
        for gamma_level, lo, hi, inv in ((1, None, None, False), (0.5, None, None, False), (2, None, None, True),
                                         (1, 0.1 * top, 0.8 * top, False), (0.7, 10, top / 3, True), (1.5, 5, 5, False)):
            r = sk.exposure.adjust_gamma(allv, gamma_level)
            if lo is not None or hi is not None:
                r = sk.exposure.rescale_intensity(r, in_range=(lo, hi))
            if inv:
                r = sk.util.invert(r)
            vec[f"pointop{k}_params"] = np.array([gamma_level, -1 if lo is None else lo, -1 if hi is None else hi, inv], dtype=np.float64)
            vec[f"pointop{k}_dtype"] = np.array(np.dtype(dt).itemsize)
            vec[f"pointop{k}_table"] = np.asarray(r).reshape(-1)
            assert np.asarray(r).dtype == dt
            k += 1
    vec["n_pointops"] = np.array(k)
    g = 0
    for im in (rng.integers(0, 256, size=(37, 53), dtype=np.uint8), rng.integers(0, 256, size=(24, 31, 3), dtype=np.uint8),
               rng.integers(0, 4096, size=(33, 40), dtype=np.uint16)):
        vec[f"gauss{g}_image"] = im
        vec[f"gauss{g}_out"] = sk.filters.gaussian(im, sigma=1)                   # detection.py:166
        g += 1
    vec["n_gauss"] = np.array(g)
    np.savez_compressed(os.path.join(HERE, "ref_skimage_vectors.npz"), **vec)
    print("skimage fixtures:", len(names), "detect cases;", k, "point-op tables;", g, "gaussian images")


def detect_lines_cases():
    """(name, 2-D image) -- small inputs for detect_lines (detection.py:156-201)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from image_analysis_amd.synth import synth_lines
    rng = np.random.default_rng(77)
    flat = rng.integers(80, 90, size=(72, 88)).astype(np.uint16) * 300
    flat[20:50, 10:70] = 40000

    def tiny(seed, H=17, W=23):                      # tiny frames: many components pass the box filter
        r = np.random.default_rng(seed)
        img = r.normal(90, 12, (H, W))
        img[H // 5:H // 5 + max(1, H // 10), W // 10:W - W // 10] += 80
        img[H // 2:H // 2 + H // 8, W // 8:W - W // 4] += 50
        return np.clip(img, 0, 255).astype(np.uint8)
    return [("lines_96x128_u8", synth_lines(96, 128, 1, np.uint8)),
            ("lines_131x77_u16", synth_lines(131, 77, 2, np.uint16)),
            ("lines_200x264_u8", synth_lines(200, 264, 3, np.uint8)),
            ("lines_384x512_u8", synth_lines(384, 512, 4, np.uint8)),
            ("tiny_seed2", tiny(2)), ("tiny_seed7", tiny(7)), ("tiny_seed23", tiny(23)),
            ("noise_64x64_u8", rng.integers(0, 256, size=(64, 64), dtype=np.uint8)),
            ("noise_17x23_u8", rng.integers(0, 256, size=(17, 23), dtype=np.uint8)),
            ("noise_9x8_u8", rng.integers(0, 256, size=(9, 8), dtype=np.uint8)),
            ("constant_40x56_u8", np.full((40, 56), 77, np.uint8)),
            ("zeros_33x47_u8", np.zeros((33, 47), np.uint8)),
            ("plateau_72x88_u16", flat),
            ("dark_50x61_u16", rng.integers(0, 40, size=(50, 61)).astype(np.uint16))]


def write_detect_lines_fixtures():
    """The reference's detect_lines (detection.py:156-201) run over the vendored skimage sources up to its label() call
    (Cython): every intermediate as SHA-256, the percentiles and the image handed to label() as arrays."""
    import vendored_skimage as V
    warnings.simplefilter("ignore")
    run = V.reference_detect_lines()
    out, names = {}, []
    for name, im in detect_lines_cases():
        t = run(im)
        names.append(name)
        out[f"{name}__image"] = im
        for k in ("gauss", "local", "gamma", "enhanced"):
            out[f"{name}__{k}_sha"] = sha(t[k])
        out[f"{name}__percentiles"] = np.asarray(t["percentiles"], dtype=np.float64)
        out[f"{name}__label_input"] = np.asarray(t["label_input"], dtype=np.float64)
        assert np.array_equal(t["label_input"], t["rescaled"])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_detect_lines.npz"), **out)
    print("detect_lines fixtures:", len(names), "cases")


def write_dtype_fixtures():
    """image_analysis / inverse_image_analysis (the reference's own lines, detection.py:205-316) on regions of the other
    dtypes np.nanmean takes: float64 (with and without NaNs, rows wider than numpy's 128-element pairwise block), int8,
    int16, int32, uint32 and bool."""
    warnings.simplefilter("ignore")
    ns = load_reference_functions()
    rng = np.random.default_rng(424242)
    out, k = {}, 0
    regs = []
    for (h, w, c) in ((400, 40, 0), (120, 51, 3), (64, 17, 3), (90, 330, 0), (70, 300, 3), (37, 9, 0), (12, 5, 3)):
        u = synth_strip(rng, h, w, c, np.uint8, 2)
        f = u.astype(np.float64) / 255.0 + rng.normal(0, 1e-3, size=u.shape)
        regs.append(("f64", f))
        g = f.copy()
        g[rng.random(g.shape) < 0.02] = np.nan
        regs.append(("f64nan", g))
        regs.append(("i8", (u.astype(np.int16) - 128).astype(np.int8)))
        regs.append(("i16", (u.astype(np.int32) * 150 - 20000).astype(np.int16)))
        regs.append(("i32", (u.astype(np.int64) * 7000000 - 900000000).astype(np.int32)))
        regs.append(("u32", (u.astype(np.uint64) * 16000000).astype(np.uint32)))
        regs.append(("bool", u > int(np.median(u))))
    for rid, (kind, reg) in enumerate(regs):
        out[f"r{rid}"] = reg
        for inverse in (False, True):
            if inverse and reg.ndim != 3:
                continue
            fn = ns["inverse_image_analysis"] if inverse else ns["image_analysis"]
            for selected in (False, True):
                try:
                    prof, peaks = fn(reg, selected)
                except ValueError:
                    continue
                n, idx, area = pack_peaks(peaks)
                out[f"c{k}_rid"] = np.array(rid); out[f"c{k}_selected"] = np.array(selected); out[f"c{k}_inverse"] = np.array(inverse)
                out[f"c{k}_profile"] = prof; out[f"c{k}_n"] = np.array(n); out[f"c{k}_idx"] = idx; out[f"c{k}_area"] = area
                k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "ref_profile_dtypes.npz"), **out)
    print("dtype fixtures:", len(regs), "regions,", k, "cases")


if __name__ == "__main__":
    if "--only-dtypes" in sys.argv:
        sys.exit(write_dtype_fixtures())
    if "--only-lines" in sys.argv:
        sys.exit(write_detect_lines_fixtures())
    if "--only-samples" in sys.argv:
        sys.exit(write_real_samples())
    if "--only-skimage" in sys.argv:
        sys.exit(write_skimage_fixtures())
    sys.exit(main())
