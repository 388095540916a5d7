"""GPU parity of profiles + peaks (image_analysis / inverse_image_analysis / analyze_peaks) and of the gaussian stage.

Expected values: golden vectors made by executing the reference's own numpy-only code (tests/golden/make_golden.py)
and the known answer in the reference's export images/excel2.xlsx.  Bar: profiles bit-exact (integer sums, one float64
division), peak indices exact, areas exact (same summation order as numpy; tolerance 1e-9 relative is the documented
fallback and is asserted separately so a last-bit deviation is visible)."""
import os

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

from oracle import lfa_oracle as O


@pytest.fixture(scope="module")
def det():
    import image_analysis_amd.detection as det
    return det


def test_excel2_crop_known_answer(det, golden_dir):
    crop = np.array(Image.open(os.path.join(golden_dir, "excel2_crop.png")))
    ka = np.load(os.path.join(golden_dir, "excel2_known_answer.npz"))
    prof, peaks = det.image_analysis(crop, False)
    assert prof.dtype == np.float64
    assert np.array_equal(prof, ka["profile_column_B"])
    assert [(p[0], p[1]) for p in peaks] == [(19, 43), (1, 3)]
    assert peaks[0][2] == 452.67500000000064 and peaks[1][2] == 1.0


def test_image_analysis_reference_vectors(det, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_profile_peaks.npz"))
    n = int(g["n_cases"])
    exact_area = 0
    for k in range(n):
        reg = g[f"r{int(g[f'c{k}_rid'])}"]
        sel = bool(g[f"c{k}_selected"])
        inv = bool(g[f"c{k}_inverse"])
        fn = det.inverse_image_analysis if inv else det.image_analysis
        prof, peaks = fn(reg, sel, int(g[f"c{k}_range"]))
        assert np.array_equal(prof, g[f"c{k}_profile"]), k
        assert len(peaks) == int(g[f"c{k}_n"]), k
        for i, (s, e, a) in enumerate(peaks):
            assert (s, e) == tuple(int(v) for v in g[f"c{k}_idx"][i]), k
            want = g[f"c{k}_area"][i]
            assert abs(a - want) <= 1e-9 * max(1.0, abs(want)), k
            exact_area += int(a == want)
    assert exact_area == sum(int(g[f"c{k}_n"]) for k in range(n))       # bit-exact on this platform


def test_analyze_peaks_reference_vectors(det, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_profile_peaks.npz"))
    import image_analysis_amd._capi as capi
    eng = capi.shared_engine(0)
    profs = [g[f"x{i}"] for i in range(int(g["n_profiles"]))]
    for inv in (0, 1):
        pk = eng.peaks(profs, inverse=bool(inv))
        for i in range(len(profs)):
            assert (int(pk[i]["start"][0]), int(pk[i]["end"][0])) == tuple(int(v) for v in g[f"p{i}_{inv}"]), (i, inv)
            assert pk[i]["area"][0] == float(g[f"a{i}_{inv}"]), (i, inv)
    s, e, a = det.analyze_peaks(profs[20], inverse=False)
    assert (s, e) == tuple(int(v) for v in g["p20_0"]) and a == float(g["a20_0"])


def test_profile_errors_like_reference(det):
    with pytest.raises(ValueError):
        det.image_analysis(np.zeros((1, 5), np.uint8), False)
    with pytest.raises(ValueError):
        det.inverse_image_analysis(np.zeros((40, 20), np.uint8), False)      # 2-D input: detection.py:249


def test_select_region_rois_batch():
    """Many ROIs over a small batch in one call (config-5 shape) against the oracle, selected=True."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_rois
    imgs = np.stack([synth_rgb(600, 400, 4, 50 + i) for i in range(3)])
    rois = synth_rois(3, 600, 400, per_image=40, seed=5).astype(capi.ROI_DT)
    eng = capi.Engine(device=0, max_h=600, max_w=400, max_batch=3)
    profs, pk = eng.profiles(imgs, rois, inverse=False, rng=15)
    for j, r in enumerate(rois):
        reg = imgs[r["image"], r["r0"]:r["r1"], r["c0"]:r["c1"]]
        wp, wk = O.image_analysis(reg, True)
        assert np.array_equal(profs[j], wp), j
        assert pk[j]["n_peaks"] == len(wk), j
        for i, (s, e, a) in enumerate(wk):
            assert (pk[j]["start"][i], pk[j]["end"][i]) == (s, e), j
            assert pk[j]["area"][i] == a, j
    eng.close()


def test_analyze_batch_end_to_end():
    """detect + profile + peaks in one device pass vs the oracle, colour and 16-bit."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_u16
    for imgs in (np.stack([synth_rgb(384, 512, 8, 70 + i) for i in range(4)]),
                 np.stack([synth_u16(512, 512, 10, 80 + i) for i in range(3)])):
        mm_c = int(imgs.shape[2] / 40)
        mm_r = mm_c * 20
        eng = capi.Engine(device=0, max_h=512, max_w=512, max_batch=4)
        res, info = eng.analyze(imgs, mm_r, mm_c)
        for i, (strips, profiles, peaks) in enumerate(res):
            d = O.detect_debug(imgs[i], mm_r, mm_c)
            assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips] == d["strips"]
            for k, (a, b, c, e) in enumerate(d["strips"]):
                wp, wk = O.image_analysis(imgs[i][a:b, c:e], False)
                assert np.array_equal(profiles[k], wp)
                assert peaks[k]["n_peaks"] == len(wk)
                for q, (s0, e0, a0) in enumerate(wk):
                    assert (peaks[k]["start"][q], peaks[k]["end"][q]) == (s0, e0)
                    assert peaks[k]["area"][q] == a0
        eng.close()


def _check_analyze(eng, imgs, mm_r, mm_c):
    res, info = eng.analyze(imgs, mm_r, mm_c, max_strips=256)
    total = 0
    for i, (strips, profiles, peaks) in enumerate(res):
        d = O.detect_debug(imgs[i], mm_r, mm_c)
        assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips] == d["strips"]
        for k, (a, b, c, e) in enumerate(d["strips"]):
            crop = imgs[i][a:b, c:e]
            h = crop.shape[0]
            if int(h * .6) - int(h * .25) <= 0:
                assert peaks[k]["n_peaks"] == 0
                continue
            wp, wk = O.image_analysis(crop, False)
            assert np.array_equal(profiles[k], wp, equal_nan=True)
            assert peaks[k]["n_peaks"] == len(wk)
            for q, (s0, e0, a0) in enumerate(wk):
                assert (peaks[k]["start"][q], peaks[k]["end"][q]) == (s0, e0)
                assert peaks[k]["area"][q] == a0 or (np.isnan(a0) and np.isnan(peaks[k]["area"][q]))
            total += 1
    return total, info


def test_analyze_window_length_hint_and_its_fallback():
    """lfa_analyze_batch sizes its LDS profile slots for the longest window of the handle's last completed call.  Windows
    that turn out longer (here: min_maxr tripled on the next call) take the global-memory path; results stay the oracle's."""
    import torch
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(640, 512, 6, 500 + i) for i in range(2)])
    eng = capi.Engine(device=0, max_h=640, max_w=512, max_batch=2)
    try:
        for mm_r in (120, 120, 600, 600, 120):                     # short windows, then 3-5x longer ones than the hint, then short again
            n, _ = _check_analyze(eng, imgs, mm_r, 12)
            torch.cuda.synchronize()
            assert n >= 4
        eng.set_debug_flags(32)                                    # worst-case slots
        _check_analyze(eng, imgs, 600, 12)
        eng.set_debug_flags(0)
    finally:
        eng.close()


def test_analyze_uint16_frames_on_a_settled_handle():
    """detect + profile + peaks on 16-bit frames, the same handle four times: two-pass histogram first, then the one-pass
    form with the class plane (synthetic frames: thresholds in a sparse part of the histogram) -- strips, profiles and
    peaks stay the oracle's; the plane is used once the handle has settled."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_u16
    imgs = np.stack([synth_u16(512, 512, 10, 90 + i) for i in range(3)])
    eng = capi.Engine(device=0, max_h=512, max_w=512, max_batch=3)
    try:
        used = []
        for _ in range(4):
            n, _info = _check_analyze(eng, imgs, 240, 12)
            assert n >= 3
            used.append([eng.debug_used_plane(i) for i in range(3)])
        assert used[0] == [False] * 3
        assert eng.debug_one_pass()
        assert any(used[-1]), used
    finally:
        eng.close()


def test_analyze_with_the_largest_strip_list():
    """max_strips = 1024 is the API's upper limit: k_finalize's accepted / kept lists (dynamic LDS) are then at their largest."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(384, 512, 8, 300 + i) for i in range(2)])
    eng = capi.Engine(device=0, max_h=384, max_w=512, max_batch=2)
    try:
        res, info = eng.analyze(imgs, 240, 12, max_strips=1024)
        for i, (strips, profiles, peaks) in enumerate(res):
            d = O.detect_debug(imgs[i], 240, 12)
            assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips] == d["strips"] and len(strips) >= 4
    finally:
        eng.close()


@pytest.mark.parametrize("key", ["image2", "image7", "test_strip_buggy_bgr", "wick_crop", "austin_u16_crop"])
def test_analyze_reference_sample_images(golden_dir, key):
    """detect + profile + peaks on the reference's sample inputs (tests/golden/real_samples.npz) against the oracle."""
    import image_analysis_amd._capi as capi
    img = np.load(os.path.join(golden_dir, "real_samples.npz"))[key]
    if img.dtype == np.uint8 and img.ndim == 2:
        img = np.ascontiguousarray(np.repeat(img[:, :, None], 3, axis=2))       # cv2.imread of a grey JPEG
    eng = capi.Engine(device=0, max_h=img.shape[0], max_w=img.shape[1], max_batch=1)
    try:
        mm_c = int(img.shape[1] / 40)
        _check_analyze(eng, img[None], mm_c * 20, mm_c)
    finally:
        eng.close()


@pytest.mark.parametrize("dtype,c", [(np.uint8, 0), (np.uint8, 3), (np.uint16, 0)])
def test_analyze_many_candidates_overflow_the_row_sum_pool(dtype, c):
    """More strips than the window row-sum pool (16*H entries) holds: later strips take the pixel path; boxes are clipped
    by the image at the bottom and the right."""
    import image_analysis_amd._capi as capi
    rng = np.random.default_rng(3)
    H, W = 160, 1024
    img = rng.integers(5, 30, size=(H, W)).astype(np.float64)
    for k in range(100):
        r0 = int(rng.integers(0, H - 14)); c0 = 4 + 10 * k
        img[r0:r0 + 12, c0:c0 + 9] = 200 + rng.integers(0, 40, size=(12, 9))
    hi = 255 if dtype == np.uint8 else 4000
    img = np.clip(img * (hi / 255.0), 0, hi).astype(dtype)
    if c:
        img = np.stack([img, np.clip(img.astype(int) + 3, 0, hi).astype(dtype), np.clip(img.astype(int) - 2, 0, hi).astype(dtype)], axis=2)
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=2)
    total, info = _check_analyze(eng, np.stack([img, img[::-1].copy()]), H, 3)
    assert total > 100 and info["n_candidates"].min() >= 60
    eng.close()


@pytest.mark.parametrize("dtype,c", [(np.uint8, 0), (np.uint8, 3), (np.uint16, 0)])
def test_analyze_with_foreground_corner_uses_box_means(dtype, c):
    """labeled[0,0] != 0: the mean test compares against 1.0, so box sums matter (and the window rows come from the pool)."""
    import image_analysis_amd._capi as capi
    rng = np.random.default_rng(8)
    H, W = 256, 512
    hi = 255 if dtype == np.uint8 else 3000
    img = rng.integers(0, 3, size=(H, W)).astype(np.int64)            # background mean ~1: boxes hover around the test value
    img[0:30, 0:40] = hi                                              # component at the origin -> label 1 at (0,0)
    for k in range(9):
        c0 = 60 + 48 * k
        img[40 + 5 * k:40 + 5 * k + 150, c0:c0 + 14] = hi - 20 * k
    img[200:212, 300:309] = hi // 2                                   # small bright blob in a dark, wide box
    img = img.astype(dtype)
    if c:
        img = np.stack([img, (img // 2).astype(dtype), (img // 3).astype(dtype)], axis=2)
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=1)
    total, info = _check_analyze(eng, img[None], 200, 12)
    assert info["label00"][0] == 1 and total >= 5
    eng.close()


def test_analyze_unaligned_width_takes_the_pixel_path():
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    img = np.ascontiguousarray(synth_rgb(300, 401, 5, 21)[:, :397])
    eng = capi.Engine(device=0, max_h=300, max_w=400, max_batch=1)
    total, _ = _check_analyze(eng, img[None], 180, 9)
    assert total >= 3
    eng.close()


@pytest.mark.parametrize("shape,dtype", [((200, 300), np.uint8), ((130, 97), np.uint16), ((96, 140, 3), np.uint8),
                                          ((5, 7), np.uint8), ((300, 33, 3), np.uint8)])
def test_gaussian_matches_scipy(det, shape, dtype):
    rng = np.random.default_rng(1)
    img = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
    got = det.gaussian(img, sigma=1)
    want = O.gaussian(img, 1.0)
    assert got.dtype == np.float64 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)     # SURVEY.md section 8a row a16 tolerance
    assert np.array_equal(got, want)                                  # and in fact bit-exact (same tap order, no FMA)


@pytest.mark.parametrize("sigma", [0.5, 1.5, 2.0])
def test_gaussian_other_sigmas(det, sigma):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(70, 300, 3)).astype(np.uint8)
    got = det.gaussian(img, sigma=sigma)
    np.testing.assert_allclose(got, O.gaussian(img, sigma), rtol=1e-12, atol=1e-15)


# ---- analyze_plot pre-processing as a per-ROI table (detection.py:535-543) --------------------
def _same_analysis(got, want):
    assert np.array_equal(got[0], want[0])
    assert len(got[1]) == len(want[1])
    for (s, e, a), (ws, we, wa) in zip(got[1], want[1]):
        assert (s, e) == (ws, we) and a == wa


@pytest.mark.parametrize("gamma,lo,hi,inv", [(1, None, None, False), (0.45, None, None, False), (1.8, 20, 180, False),
                                             (1, None, None, True), (0.7, 30, 220, True), (1, 90, 90, False)])
@pytest.mark.parametrize("selected", [False, True])
def test_analyze_plot_colour_region(det, gamma, lo, hi, inv, selected):
    from image_analysis_amd.synth import synth_rgb
    img = synth_rgb(700, 480, 4, 77)
    region = img[40:640, 100:163]                               # a strip-sized view (non-contiguous, odd width)
    got = det.analyze_plot(region, selected, gamma, lo, hi, inv)
    want = O.analyze_plot(region, selected, gamma, lo, hi, inv)
    _same_analysis(got, want)


@pytest.mark.parametrize("gamma,lo,hi", [(1, None, None), (0.5, None, None), (1.5, 100, 3000), (1, 0, 4095.5)])
def test_analyze_plot_uint16_region(det, golden_dir, gamma, lo, hi):
    crop = np.array(Image.open(os.path.join(golden_dir, "excel2_crop.png")))
    assert crop.dtype == np.uint16
    for selected in (False, True):
        _same_analysis(det.analyze_plot(crop, selected, gamma, lo, hi, False), O.analyze_plot(crop, selected, gamma, lo, hi, False))
    # the GUI's default settings are the identity: same answer as image_analysis (and the excel2 known answer)
    if gamma == 1 and lo is None:
        _same_analysis(det.analyze_plot(crop, False), det.image_analysis(crop, False))


def test_analyze_plot_errors(det):
    reg = np.zeros((60, 30), np.uint8)
    with pytest.raises(ValueError):
        det.analyze_plot(reg, False, gamma_level=-0.5)
    with pytest.raises(TypeError):
        det.analyze_plot(reg, False, 1, None, 100)               # float(None), exposure.py:582
    with pytest.raises(ValueError):
        det.analyze_plot(reg, False, invert=True)                # inverse_image_analysis needs 3-D (detection.py:249)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_profile_tables_per_roi_batch(dtype):
    """Many ROIs, several tables, one call; uint8 widths that are not multiples of 4 take the pixel path."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_u16, synth_rois
    if dtype == np.uint8:
        imgs = np.stack([synth_rgb(640, 258, 3, 9 + i)[:, :, 1] for i in range(2)])          # 2-D uint8, W*C % 4 != 0
        imgs3 = np.stack([synth_rgb(640, 256, 3, 19 + i) for i in range(2)])
    else:
        imgs = np.stack([synth_u16(640, 256, 6, 9 + i) for i in range(2)])
        imgs3 = None
    params = [(1, None, None, False), (0.4, None, None, False), (2.0, 5, 200 if dtype == np.uint8 else 3000, True)]
    luts = np.stack([capi.point_op_lut(dtype, *p) for p in params])
    for batch in (imgs, imgs3):
        if batch is None:
            continue
        n, H, W = batch.shape[:3]
        rois = synth_rois(n, H, W, per_image=30, seed=3).astype(capi.ROI_DT)
        which = (np.arange(len(rois)) % len(params)).astype(np.int32)
        eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=n)
        profs, pk = eng.profiles(batch, rois, inverse=False, rng=15, luts=luts, roi_lut=which)
        for j, r in enumerate(rois):
            reg = batch[r["image"], r["r0"]:r["r1"], r["c0"]:r["c1"]]
            g, lo, hi, inv = params[which[j]]
            reg = O.adjust_gamma(reg, g)
            if lo is not None:
                reg = O.rescale_intensity(reg, (lo, hi))
            if inv:
                reg = O.invert(reg)
            wp, wk = O.image_analysis(reg, True)
            assert np.array_equal(profs[j], wp), j
            assert pk[j]["n_peaks"] == len(wk), j
            for i, (s, e, a) in enumerate(wk):
                assert (pk[j]["start"][i], pk[j]["end"][i]) == (s, e) and pk[j]["area"][i] == a, j
        # a table index out of range marks that ROI, and only that ROI, as bad
        which_bad = which.copy()
        which_bad[1] = len(params)
        with pytest.raises(ValueError):
            eng.profiles(batch, rois, luts=luts, roi_lut=which_bad)
        _p, pk2 = eng.profiles(batch, rois, luts=luts, roi_lut=which_bad, strict=False)
        assert len(_p[1]) == 0 and len(_p[0]) == pk[0]["length"]
        assert pk2[1]["n_peaks"] == -1 and pk2[0]["n_peaks"] == pk[0]["n_peaks"] and pk2[2]["area"][0] == pk[2]["area"][0]
        eng.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_profiles_peak_logic_against_oracle(seed):
    """Seeded fuzz of k_peaks: lengths 1..700 (wide peaks cross numpy's 128-term pairwise split), quantised values
    (ties and plateaus), ramps, both polarities, several window sizes and ranges, single- and two-peak logic."""
    import warnings
    import image_analysis_amd._capi as capi
    rng = np.random.default_rng(500 + seed)
    eng = capi.shared_engine(0)
    profs = []
    for _ in range(60):
        n = int(rng.integers(1, 700))
        x = np.arange(n)
        kind = rng.integers(0, 4)
        if kind == 0:
            p = rng.normal(100, 10, n)
        elif kind == 1:                                             # broad bumps: walks and trapezoid sums of hundreds of terms
            p = 50 + 80 * np.exp(-0.5 * ((x - rng.uniform(0, n)) / rng.uniform(3, 150)) ** 2) + rng.normal(0, 0.5, n)
        elif kind == 2:
            p = np.cumsum(rng.normal(0, 1, n)) + rng.uniform(-50, 50)
        else:
            p = rng.integers(0, 4, n).astype(np.float64)            # heavy ties
        if rng.random() < 0.5:
            p = np.round(p * 4) / 4
        profs.append(np.ascontiguousarray(p, dtype=np.float64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for inverse in (False, True):
            for window in (25, 3):
                pk = eng.peaks(profs, inverse=inverse, window_size=window)
                for i, p in enumerate(profs):
                    s, e, a = O.analyze_peaks(p, inverse=inverse, window_size=window)
                    assert (int(pk[i]["start"][0]), int(pk[i]["end"][0])) == (s, e), (seed, i, inverse, window)
                    assert pk[i]["area"][0] == a or (np.isnan(a) and np.isnan(pk[i]["area"][0])), (seed, i, inverse, window)
            for r in (15, 1):
                pk = eng.peaks(profs, inverse=inverse, two_peaks=True, rng=r)
                for i, p in enumerate(profs):
                    want = O._two_peaks(p, inverse, r, fallback=not inverse)
                    assert pk[i]["n_peaks"] == len(want), (seed, i, inverse, r)
                    for q, (s, e, a) in enumerate(want):
                        assert (int(pk[i]["start"][q]), int(pk[i]["end"][q])) == (s, e), (seed, i, inverse, r, q)
                        assert pk[i]["area"][q] == a, (seed, i, inverse, r, q)


@pytest.mark.parametrize("kind", ["rgb", "u16", "u8"])
def test_dense_rois_take_the_band_kernel_and_match_the_oracle(kind):
    """lfa_profile_batch with many ROIs per image (configs[4]): ROI array sorted by image -> k_profile_band (row bands staged
    in LDS, every ROI crossing a band served from it); the same ROIs shuffled -> the per-ROI kernel.  Both must equal the
    oracle bit for bit; degenerate ROIs (no rows, no columns) included."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_rois, synth_u16
    if kind == "rgb":
        imgs = np.stack([synth_rgb(200, 256, 4, 70 + i) for i in range(3)])
    elif kind == "u16":
        imgs = np.stack([synth_u16(200, 256, 6, 70 + i) for i in range(3)])
    else:
        imgs = np.stack([synth_rgb(200, 256, 4, 70 + i)[:, :, 0].copy() for i in range(3)])
    n, H, W = imgs.shape[:3]
    rng = np.random.default_rng(8)
    rois = np.zeros(n * 48, dtype=capi.ROI_DT)
    rois["image"] = np.repeat(np.arange(n), 48)
    h = rng.integers(1, 150, size=len(rois)); w = rng.integers(1, 64, size=len(rois))
    rois["r0"] = (rng.random(len(rois)) * (H - h)).astype(int); rois["r1"] = rois["r0"] + h
    rois["c0"] = (rng.random(len(rois)) * (W - w)).astype(int); rois["c1"] = rois["c0"] + w
    rois["r1"][5] = rois["r0"][5]                              # no rows: empty profile
    rois["c1"][7] = rois["c0"][7]                              # no columns: NaN profile (np.nanmean of an empty slice)
    rois[9] = (1, 0, H, 0, W)                                  # the whole image
    rois = rois[np.argsort(rois["image"], kind="stable")]
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=n)
    try:
        for order in (np.arange(len(rois)), rng.permutation(len(rois))):
            rr = rois[order]
            eng.timing("*")
            profs, pk = eng.profiles(imgs, rr, inverse=False, rng=15)
            kt = eng.timing_read()
            eng.timing(None)
            assert "k_profile_band" in kt and "k_roi_index" in kt
            for q, r in enumerate(rr):
                reg = imgs[r["image"], r["r0"]:r["r1"], r["c0"]:r["c1"]]
                if reg.shape[0] == 0:
                    assert len(profs[q]) == 0 and pk[q]["n_peaks"] == 0
                    continue
                wp, wk = O.image_analysis(reg, True)
                assert np.array_equal(profs[q], wp, equal_nan=True), q
                if reg.shape[1] == 0:
                    continue
                assert pk[q]["n_peaks"] == len(wk), q
                for i, (s, e, a) in enumerate(wk):
                    assert (pk[q]["start"][i], pk[q]["end"][i]) == (s, e) and pk[q]["area"][i] == a, q
    finally:
        eng.close()


@pytest.mark.parametrize("shape,dtype", [((64, 256, 3), np.uint8), ((37, 1000, 3), np.uint8), ((50, 352), np.uint8), ((130, 100), np.uint16),
                                          ((40, 2048), np.uint16), ((33, 97, 3), np.uint8), ((5, 7), np.uint8)])
@pytest.mark.parametrize("sigma", [1.0, 2.0, 0.5])
def test_gaussian_float32_output(shape, dtype, sigma):
    """out_f32 (no counterpart in the reference, which returns float64): float32 arithmetic on the vector path (rows of a
    multiple of 4 elements), the float64 kernel with a rounded store otherwise; either way within 1e-6 of the float64
    result, on values in [0, 1] (stated tolerance of the float32 variant: atol 1e-6)."""
    import image_analysis_amd._capi as capi
    rng = np.random.default_rng(4)
    img = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
    eng = capi.Engine(device=0, max_h=shape[0], max_w=shape[1], max_batch=2)
    try:
        got = eng.gaussian(np.stack([img, img[::-1].copy()]), sigma=sigma, out_f32=True)
        assert got.dtype == np.float32
        for k, im in enumerate((img, img[::-1])):
            want = O.gaussian(im, sigma)
            np.testing.assert_allclose(got[k].astype(np.float64), want, rtol=0, atol=1e-6)
    finally:
        eng.close()


def test_regions_of_other_dtypes_equal_the_reference_run():
    """The drop-in's image_analysis / inverse_image_analysis on float64 (NaNs, rows wider than numpy's 128-element pairwise
    block), int8, int16, int32, uint32 and bool regions against vectors made by the reference's own lines
    (tests/golden/ref_profile_dtypes.npz): profiles, peak indices and areas bit for bit (lfa_profile_any_batch)."""
    import os
    import image_analysis_amd.detection as detection
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_profile_dtypes.npz"))
    kinds = set()
    for k in range(int(G["n_cases"])):
        reg = G[f"r{int(G[f'c{k}_rid'])}"]
        kinds.add(reg.dtype.name)
        fn = detection.inverse_image_analysis if G[f"c{k}_inverse"] else detection.image_analysis
        prof, peaks = fn(reg, bool(G[f"c{k}_selected"]))
        assert prof.dtype == np.float64 and prof.tobytes() == G[f"c{k}_profile"].tobytes(), (k, reg.dtype)
        assert len(peaks) == int(G[f"c{k}_n"])
        for i, (s, e, a) in enumerate(peaks):
            assert [s, e] == G[f"c{k}_idx"][i].tolist() and np.float64(a).tobytes() == G[f"c{k}_area"][i].tobytes(), (k, i)
    assert len(kinds) == 6
    with pytest.raises(NotImplementedError):
        detection.image_analysis(np.zeros((40, 20), np.float32), True)


def _check_rois_against_oracle(eng, imgs, rois, want_band, rng):
    """profiles + peaks of `rois` in the order given and shuffled; each ROI's oracle result is computed once."""
    want = {}
    for order in (np.arange(len(rois)), rng.permutation(len(rois))):
        rr = rois[order]
        eng.timing("*")
        profs, pk = eng.profiles(imgs, rr, inverse=False, rng=15)
        kt = eng.timing_read()
        eng.timing(None)
        assert ("k_profile_band" in kt) == want_band, sorted(kt)
        for q, (r, src) in enumerate(zip(rr, order)):
            if int(src) not in want:
                want[int(src)] = O.image_analysis(imgs[r["image"], r["r0"]:r["r1"], r["c0"]:r["c1"]], True)
            wp, wk = want[int(src)]
            assert profs[q].tobytes() == wp.tobytes(), (q, r)
            assert pk[q]["n_peaks"] == len(wk), (q, r)
            for i, (s, e, a) in enumerate(wk):
                assert (pk[q]["start"][i], pk[q]["end"][i]) == (s, e) and pk[q]["area"][i] == a, (q, r, i)


def test_dense_rois_at_the_benchmarked_shape_1536x2048x3():
    """BASELINE configs[4] at the shape bench.py reports: 1536x2048x3 uint8 images, 256 ROIs per image from synth_rois
    (h ~ U{64..512}, w ~ U{16..64}).  Rows are 6144 bytes, so k_profile_band runs at its ceiling -- a 4-row band of exactly
    24 576 bytes, six 16-byte pieces per thread.  ROI array sorted by image -> band kernel; shuffled -> k_profile.  Every
    profile bit for bit, every peak index and area equal to the oracle (detection.py:205-245 on the slice of :529-533)."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_rois
    H, W, n = 1536, 2048, 2
    imgs = np.stack([synth_rgb(H, W, 8, 1234 + i) for i in range(n)])
    rois = synth_rois(n, H, W, per_image=256).astype(capi.ROI_DT)
    assert len(rois) == 512 and (rois["r1"] - rois["r0"]).max() > 480
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=n)
    try:
        _check_rois_against_oracle(eng, imgs, rois, True, np.random.default_rng(3))
    finally:
        eng.close()


def test_dense_rois_on_2048x2048_uint16_frames():
    """The same on 16-bit frames: 4096-byte rows, a 4-row band of 16 KiB (8 rows would pass the 24 KiB ceiling)."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rois, synth_u16
    H = W = 2048
    imgs = np.stack([synth_u16(H, W, 40, 1234 + i) for i in range(2)])
    rois = synth_rois(2, H, W, per_image=160, seed=5).astype(capi.ROI_DT)
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=2)
    try:
        _check_rois_against_oracle(eng, imgs, rois, True, np.random.default_rng(4))
    finally:
        eng.close()


def test_more_rois_on_one_image_than_a_band_block_lists_at_once():
    """k_profile_band lists the ROIs of its image 512 at a time (kBandList): one image carrying 1100 ROIs takes three rounds
    per band block, its neighbour carries 12.  Includes ROIs that end on the image's last row / column."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_rois
    H, W = 768, 1024
    imgs = np.stack([synth_rgb(H, W, 6, 300 + i) for i in range(2)])
    a = synth_rois(1, H, W, per_image=1100, seed=11).astype(capi.ROI_DT)
    b = synth_rois(1, H, W, per_image=12, seed=12).astype(capi.ROI_DT)
    b["image"] = 1
    a[3] = (0, H - 70, H, W - 20, W)                         # bottom-right corner of the image
    a[700] = (0, 0, 64, 0, 16)                               # top-left corner
    rois = np.concatenate([a, b])
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=2)
    try:
        _check_rois_against_oracle(eng, imgs, rois, True, np.random.default_rng(5))
    finally:
        eng.close()
