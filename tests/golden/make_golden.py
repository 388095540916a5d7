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
    out = {}
    k = 0
    cases = []
    for dtype in (np.uint8, np.uint16):
        for c in (0, 3):
            for (h, w) in ((1020, 51), (400, 40), (64, 17), (37, 9), (12, 5)):
                for selected in (False, True):
                    cases.append((dtype, c, h, w, selected, int(rng.integers(1, 4))))
    for dtype, c, h, w, selected, nb in cases:
        reg = synth_strip(rng, h, w, c, dtype, nb)
        for inverse in (False, True):
            if inverse and not c:
                continue                                   # detection.py:249 needs three dimensions
            fn = inverse_image_analysis if inverse else image_analysis
            try:
                prof, peaks = fn(reg, selected)
            except ValueError:
                continue                                   # empty profile (argmax of empty) -- tested separately
            n, idx, area = pack_peaks(peaks)
            out[f"c{k}_region"] = reg
            out[f"c{k}_selected"] = np.array(selected)
            out[f"c{k}_inverse"] = np.array(inverse)
            out[f"c{k}_profile"] = prof
            out[f"c{k}_n"] = np.array(n)
            out[f"c{k}_idx"] = idx
            out[f"c{k}_area"] = area
            k += 1
    out["n_cases"] = np.array(k)

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


if __name__ == "__main__":
    if "--only-samples" in sys.argv:
        sys.exit(write_real_samples())
    sys.exit(main())
