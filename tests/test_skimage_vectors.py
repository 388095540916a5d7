"""Pins oracle/lfa_oracle.py to scikit-image 0.20.0's OWN code and its own test vectors (CPU only).

tests/golden/ref_skimage_detect.npz holds, for 22 small inputs, what the reference's detect_lateral_flow_tests
(detection.py:31-153) produced when executed over the reference's vendored skimage Python sources (rgb2gray, adjust_gamma,
threshold_otsu + exposure.histogram, binary_erosion / dilation, label, regionprops -- tests/golden/vendored_skimage.py;
only the dead canny / Hough calls were cut out).  ref_skimage_vectors.npz holds expected values read out of the vendored
skimage test-suite, and outputs of its gaussian / rescale_intensity / invert.  Every stage of the oracle must reproduce them
bit for bit."""
import os

import numpy as np
import pytest
from PIL import Image

from golden_util import GOLDEN, clipped, sha, skimage_detect_cases
from oracle import lfa_oracle as O

G, NAMES = skimage_detect_cases()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_detect_equals_the_reference_run_over_vendored_skimage(name):
    im = G[f"{name}__image"]
    mm_r, mm_c, L = (int(v) for v in G[f"{name}__params"])
    d = O.detect_debug(im, mm_r, mm_c, L)
    want_thr = G[f"{name}__threshold"]
    assert np.asarray(d["threshold"]).dtype.kind == want_thr.dtype.kind
    assert np.asarray(d["threshold"], dtype=want_thr.dtype).tobytes() == want_thr.tobytes()          # same bits
    if im.ndim == 3:
        assert np.array_equal(sha(O.rgb2gray(im)), G[f"{name}__gray_sha"])
    assert np.array_equal(sha(d["gamma"]), G[f"{name}__gamma_sha"])
    assert np.array_equal(sha(d["mask"].astype(np.uint8)), G[f"{name}__mask_sha"])
    assert np.array_equal(sha(d["opened"].astype(np.uint8)), G[f"{name}__opened_sha"])
    assert np.array_equal(sha(d["labels"].astype(np.int32)), G[f"{name}__labels_sha"])
    assert d["n_labels"] == int(G[f"{name}__n_labels"])
    assert int(d["background"]) == int(G[f"{name}__label00"])
    assert np.array_equal(d["table"], G[f"{name}__table"])
    assert clipped(d["strips"], *im.shape[:2]) == G[f"{name}__boxes"].tolist()


def test_fixture_covers_the_branches():
    """What the cases were chosen for must actually occur in them (else the fixture pins less than it claims)."""
    boxes = sum(len(G[f"{n}__boxes"]) for n in NAMES)
    assert boxes >= 60
    assert any(int(G[f"{n}__label00"]) != 0 for n in NAMES)                             # foreground corner
    assert any(len(G[f"{n}__boxes"]) == 0 for n in NAMES)                               # constant image -> no strips
    assert any(G[f"{n}__image"].ndim == 3 for n in NAMES) and any(G[f"{n}__image"].dtype == np.uint16 for n in NAMES)
    assert any(int(G[f"{n}__n_labels"]) > 50 for n in NAMES)                              # (1149 on the whole excel2 frame, below)
    assert any(len(G[f"{n}__table"]) > len(G[f"{n}__boxes"]) > 0 for n in NAMES)        # regions rejected


def test_excel2_numbers_are_reference_outputs():
    """5086 / 1149 / 40 and the exported crop box, produced by the reference's own detect on its own frame."""
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    assert int(v["excel2_threshold"]) == 5086 and int(v["excel2_n_labels"]) == 1149 and len(v["excel2_boxes"]) == 40
    assert [324, 1344, 806, 857] in v["excel2_boxes"].tolist()
    frame = np.array(Image.open(os.path.join(GOLDEN, "excel2_frame.png")))
    d = O.detect_debug(frame, 1020, 51)
    assert d["threshold"] == 5086 and d["n_labels"] == 1149
    assert clipped(d["strips"], 2048, 2048) == v["excel2_boxes"].tolist()
    assert np.array_equal(sha(d["mask"].astype(np.uint8)), v["excel2_mask_sha"])
    assert np.array_equal(sha(d["labels"].astype(np.int32)), v["excel2_labels_sha"])


def test_gamma_tables_of_the_vendored_test_suite():
    """skimage/exposure/tests/test_exposure.py:622-686: adjust_gamma(arange(256), 0.5) and (…, 2), all 256 entries."""
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    image = np.arange(0, 256, dtype=np.uint8)
    assert np.array_equal(O.adjust_gamma(image, 0.5), v["gamma_half_table"])
    assert np.array_equal(O.adjust_gamma(image, 2), v["gamma_two_table"])
    assert np.array_equal(O.gamma_lut_u8(0.5), v["gamma_half_table"])
    assert np.array_equal(O.adjust_gamma(image, 1), image)                              # :606-610
    with pytest.raises(ValueError):
        O.adjust_gamma(image, -1)                                                       # :689-692


def test_otsu_vectors_of_the_vendored_test_suite():
    """skimage/filters/tests/test_thresholding.py:33-61 (== 2, shifted by -2: == 0, float image: in [2, 3)) and
    :307-314 (one-colour image: that colour)."""
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    img = v["otsu_simple_image"]
    assert O.threshold_otsu(img) == 2
    assert O.threshold_otsu(img - 2) == 0
    assert 2 <= O.threshold_otsu(np.float64(img)) < 3
    assert O.threshold_otsu(np.ones((10, 10), dtype=np.uint8)) == 1
    assert O.threshold_otsu(np.ones((10, 10, 10), dtype=np.uint8)) == 1


def test_rgb2gray_vectors_of_the_vendored_test_suite():
    """skimage/color/tests/test_colorconv.py:354-380: white -> 1 (to 6 decimals), shape, alpha / 2-D input rejected."""
    x = np.array([1, 1, 1]).reshape((1, 1, 3)).astype(float)
    g = O.rgb2gray(x)
    assert g.shape == (1, 1) and abs(g[0, 0] - 1) < 1.5e-6
    with pytest.raises(ValueError):
        O.rgb2gray(np.empty((10, 10, 4)))
    with pytest.raises(ValueError):
        O.rgb2gray(np.empty((5, 5)))
    assert O.rgb2gray(np.random.rand(10, 10, 3)).dtype == np.float64


def test_point_operations_equal_the_vendored_skimage():
    """adjust_gamma -> rescale_intensity -> invert (MyGUI.analyze_plot, detection.py:535-539) on every value of uint8 and
    uint16, against the tables the vendored skimage functions produced."""
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    for k in range(int(v["n_pointops"])):
        gamma_level, lo, hi, inv = v[f"pointop{k}_params"]
        dt = np.uint8 if int(v[f"pointop{k}_dtype"]) == 1 else np.uint16
        r = O.adjust_gamma(np.arange(np.iinfo(dt).max + 1, dtype=dt), gamma_level)
        if lo >= 0:
            r = O.rescale_intensity(r, (lo, hi))
        if inv:
            r = O.invert(r)
        assert r.dtype == dt and np.array_equal(r, v[f"pointop{k}_table"]), k


def test_gaussian_equals_the_vendored_skimage():
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    for g in range(int(v["n_gauss"])):
        out = O.gaussian(v[f"gauss{g}_image"], 1.0)
        assert out.dtype == np.float64 and np.array_equal(out, v[f"gauss{g}_out"]), g
