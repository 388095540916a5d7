"""Pins oracle/lfa_oracle.py to the reference's own data and to outputs of the reference's numpy-only code.

CPU only.  Sources of truth (see tests/golden/make_golden.py):
  * excel2_*           -- artefacts written by the reference's own export (images/excel2.xlsx)
  * ref_profile_peaks  -- detection.py:205-316 executed in the build container
  * ref_region_lists   -- detection.py:118-153 executed in the build container
"""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import lfa_oracle as O


@pytest.fixture(scope="module")
def excel2(golden_dir):
    frame = np.array(Image.open(os.path.join(golden_dir, "excel2_frame.png")))
    crop = np.array(Image.open(os.path.join(golden_dir, "excel2_crop.png")))
    ka = np.load(os.path.join(golden_dir, "excel2_known_answer.npz"))
    return frame, crop, ka


def test_excel2_frame_shape(excel2):
    frame, crop, ka = excel2
    assert frame.shape == (2048, 2048) and frame.dtype == np.uint16
    assert crop.shape == (1020, 51) and crop.dtype == np.uint16


def test_excel2_detect_finds_the_exported_crop(excel2):
    frame, crop, ka = excel2
    min_maxc = int(frame.shape[1] / 40)
    min_maxr = int(min_maxc * 20)
    d = O.detect_debug(frame, min_maxr, min_maxc)
    assert d["threshold"] == 5086            # SURVEY.md section 4 (probed when the reference was surveyed)
    assert d["n_labels"] == 1149
    assert len(d["strips"]) == 40
    box = [int(v) for v in ka["crop_box"]]
    assert box in d["strips"]
    crops = O.detect_lateral_flow_tests(frame, min_maxr, min_maxc)
    assert any(c.shape == crop.shape and np.array_equal(c, crop) for c in crops)


def test_excel2_profile_and_peaks(excel2):
    frame, crop, ka = excel2
    prof, peaks = O.image_analysis(crop, False)
    assert np.array_equal(prof, ka["profile_column_B"])          # what the reference wrote into the workbook
    assert np.array_equal(prof, ka["ref_profile"])
    n = int(ka["ref_n_peaks"])
    assert len(peaks) == n == 2
    for i in range(n):
        assert (peaks[i][0], peaks[i][1]) == tuple(int(v) for v in ka["ref_peak_idx"][i])
        assert peaks[i][2] == ka["ref_peak_area"][i]
    assert peaks[0][:2] == (19, 43) and peaks[1][:2] == (1, 3)     # BASELINE.md known answer
    assert peaks[0][2] == 452.67500000000064 and peaks[1][2] == 1.0


def test_image_analysis_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_profile_peaks.npz"))
    n_cases = int(g["n_cases"])
    assert n_cases > 200
    assert {int(g[f'c{k}_range']) for k in range(n_cases) if bool(g[f'c{k}_inverse'])} == {1, 3, 15}
    for k in range(n_cases):
        reg = g[f"r{int(g[f'c{k}_rid'])}"]
        sel = bool(g[f"c{k}_selected"])
        inv = bool(g[f"c{k}_inverse"])
        fn = O.inverse_image_analysis if inv else O.image_analysis
        prof, peaks = fn(reg, sel, int(g[f"c{k}_range"]))
        assert np.array_equal(prof, g[f"c{k}_profile"]), k
        assert len(peaks) == int(g[f"c{k}_n"]), k
        for i, (s, e, a) in enumerate(peaks):
            assert (s, e) == tuple(int(v) for v in g[f"c{k}_idx"][i]), k
            assert a == g[f"c{k}_area"][i], k


def test_analyze_peaks_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_profile_peaks.npz"))
    for i in range(int(g["n_profiles"])):
        x = g[f"x{i}"]
        for inv in (0, 1):
            s, e, a = O.analyze_peaks(x, inverse=bool(inv))
            assert (s, e) == tuple(int(v) for v in g[f"p{i}_{inv}"]), (i, inv)
            assert a == float(g[f"a{i}_{inv}"]), (i, inv)


def test_region_lists_match_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_region_lists.npz"))
    for t in range(int(g["n_trials"])):
        mm_r, mm_c = (int(v) for v in g[f"t{t}_params"])
        acc = g[f"t{t}_accepted"].tolist()
        # replay: every accepted entry must pass check_duplicate against its predecessors, in order
        seen = []
        for a, b, c, d in acc:
            assert O.check_duplicate(a, c, b, d, seen, mm_r, mm_c)
            seen.append([a, b, c, d])
            assert not O.check_duplicate(a, c, b, d, seen, mm_r, mm_c)
        assert O.remove_outliers(acc, mm_c) == g[f"t{t}_kept"].tolist()


def test_empty_profile_raises_like_reference():
    with pytest.raises(ValueError):
        O.image_analysis(np.zeros((1, 5), np.uint8), False)       # int(1*.25)==int(1*.6): argmax of empty


@pytest.mark.parametrize("L", [1, 2, 3, 5, 10, 11])
def test_vertical_opening_is_run_length_filter(L):
    """The closed form the HIP kernel uses: a vertical run survives iff
    len + (touches top ? L//2 : 0) + (touches bottom ? (L-1)//2 : 0) >= L  (SURVEY.md section 3.1 D6)."""
    rng = np.random.default_rng(L)
    for H in (1, 2, 4, 5, 6, 9, 10, 11, 23, 64, 65, 130):
        m = rng.random((H, 37)) < 0.7
        got = O.vertical_opening(m, L)
        want = np.zeros_like(m)
        for c in range(m.shape[1]):
            r = 0
            while r < H:
                if not m[r, c]:
                    r += 1
                    continue
                s = r
                while r < H and m[r, c]:
                    r += 1
                ln = r - s
                eff = ln + (L // 2 if s == 0 else 0) + ((L - 1) // 2 if r == H else 0)
                if eff >= L:
                    want[s:r, c] = True
        assert np.array_equal(got, want), (L, H)


def test_label_order_is_raster_order_of_first_pixel():
    rng = np.random.default_rng(3)
    for _ in range(50):
        m = rng.random((40, 60)) < 0.35
        lab, k = O.label8(m)
        firsts = [np.flatnonzero(lab.ravel() == i)[0] for i in range(1, k + 1)]
        assert firsts == sorted(firsts)
        assert lab.dtype == np.int32


def test_point_operations_match_the_reference_docstrings():
    """rescale_intensity / invert restatements against the examples the reference's vendored skimage holds
    (skimage/exposure/exposure.py:532-534, skimage/util/_invert.py:43-49); beyond those two vectors the analyze_plot
    pre-processing is pinned on numpy semantics only ("parity unpinned" in the strict sense)."""
    img = np.array([51, 102, 153], dtype=np.uint8)
    assert O.rescale_intensity(img, (51, 153)).tolist() == [0, 127, 255]           # in_range='image' of the example
    assert O.rescale_intensity(img, (51, 153)).dtype == np.uint8
    inv = O.invert(np.array([[100, 0, 200], [0, 50, 0], [30, 0, 255]], np.uint8))
    assert inv.tolist() == [[155, 255, 55], [255, 205, 255], [225, 255, 0]]
    assert O.invert(np.array([0, 65535, 100], np.uint16)).tolist() == [65535, 0, 65435]
    # constant in_range: clip to the dtype range (exposure.py:600)
    assert O.rescale_intensity(img, (100, 100)).tolist() == [100, 100, 100]
    with pytest.raises(TypeError):
        O.rescale_intensity(img, (None, 100))


def test_profiles_of_other_region_dtypes_equal_the_reference_run():
    """image_analysis / inverse_image_analysis on float64 (with NaNs, rows beyond numpy's 128-element pairwise block), int8,
    int16, int32, uint32 and bool regions: vectors made by the reference's own lines (make_golden.py --only-dtypes)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_profile_dtypes.npz"))
    from oracle import lfa_oracle as O
    kinds = set()
    for k in range(int(G["n_cases"])):
        reg = G[f"r{int(G[f'c{k}_rid'])}"]
        kinds.add(reg.dtype.name)
        fn = O.inverse_image_analysis if G[f"c{k}_inverse"] else O.image_analysis
        prof, peaks = fn(reg, bool(G[f"c{k}_selected"]))
        assert prof.dtype == np.float64 and prof.tobytes() == G[f"c{k}_profile"].tobytes()
        assert len(peaks) == int(G[f"c{k}_n"])
        for i, (s, e, a) in enumerate(peaks):
            assert [s, e] == G[f"c{k}_idx"][i].tolist() and np.float64(a).tobytes() == G[f"c{k}_area"][i].tobytes()
    assert kinds == {"float64", "int8", "int16", "int32", "uint32", "bool"}
