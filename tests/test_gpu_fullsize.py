"""GPU parity at BASELINE.json's full image sizes (oracle still finishes in about a second per image) and
size-independent properties on a larger batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lfa_oracle as O


def _strips(st):
    return [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in st]


def _check_full(imgs, max_batch):
    import image_analysis_amd._capi as capi
    H, W = imgs.shape[1:3]
    mm_c = int(W / 40)
    mm_r = mm_c * 20
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=max_batch)
    res, info = eng.analyze(imgs, mm_r, mm_c)
    assert (info["ambiguous"] == 0).all()
    for i, (strips, profiles, peaks) in enumerate(res):
        d = O.detect_debug(imgs[i], mm_r, mm_c)
        assert capi.host_threshold(info[i], imgs.ndim == 4) == d["threshold"]
        assert info[i]["n_labels"] == d["n_labels"]
        assert _strips(strips) == d["strips"]
        for k, (a, b, c, e) in enumerate(d["strips"]):
            wp, wk = O.image_analysis(imgs[i][a:b, c:e], False)
            assert np.array_equal(profiles[k], wp)
            assert [(int(peaks[k]["start"][q]), int(peaks[k]["end"][q]), float(peaks[k]["area"][q]))
                    for q in range(int(peaks[k]["n_peaks"]))] == [(s0, e0, float(a0)) for s0, e0, a0 in wk]
    eng.close()
    return res, info


def test_config2_size_colour_1536x2048():
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(1536, 2048, 8, 1234 + i) for i in range(3)])
    res, info = _check_full(imgs, 3)
    assert all(len(r[0]) == 8 for r in res)


def test_config3_size_uint16_2048x2048():
    from image_analysis_amd.synth import synth_u16
    imgs = np.stack([synth_u16(2048, 2048, 40, 1234 + i) for i in range(2)])
    res, info = _check_full(imgs, 2)
    assert all(len(r[0]) >= 30 for r in res) and info["n_labels"].min() > 300


def test_config4_size_colour_3072x4096():
    from image_analysis_amd.synth import synth_rgb
    imgs = synth_rgb(3072, 4096, 8, 77)[None]
    res, info = _check_full(imgs, 1)
    assert len(res[0][0]) == 8


@pytest.mark.parametrize("kind", ["colour", "uint8", "uint16"])
def test_images_above_2_24_pixels(kind):
    """More than 2^24 pixels: skimage's float32 histogram weights (thresholding.py:309) are no longer exact and round in
    bin order; the colour path restates that order anyway, the integer path switches to the sequential float32 walk
    (lfa_algo.h otsu_seq_*, pinned on the host against the oracle with far larger counts)."""
    from image_analysis_amd.synth import synth_rgb, synth_u16
    H, W = 4104, 4112                                      # 16.88 M pixels
    if kind == "colour":
        img = synth_rgb(H, W, 8, 7)
    elif kind == "uint8":
        img = np.ascontiguousarray(synth_rgb(H, W, 8, 8)[:, :, 1])
    else:
        img = synth_u16(H, W, 40, 9)
    assert img.shape[0] * img.shape[1] > (1 << 24)
    res, info = _check_full(img[None], 1)
    assert len(res[0][0]) >= 8


def test_batch_properties_and_determinism():
    """A tiled batch must give every copy of an image the same answer, twice in a row (atomics / union-find order
    must not leak into results), and flipping an image left-right must mirror the strip columns."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    base = np.stack([synth_rgb(768, 1024, 6, 500 + i) for i in range(4)])
    imgs = np.concatenate([base] * 6)                        # 24 images
    mm_c = int(1024 / 40)
    mm_r = mm_c * 20
    eng = capi.Engine(device=0, max_h=768, max_w=1024, max_batch=24)
    r1, i1 = eng.analyze(imgs, mm_r, mm_c)
    r2, i2 = eng.analyze(imgs, mm_r, mm_c)
    for k in range(24):
        a, b, c = r1[k], r1[k % 4], r2[k]
        for x in (b, c):
            assert np.array_equal(a[0], x[0]) and np.array_equal(a[2], x[2])
            assert all(np.array_equal(p, q) for p, q in zip(a[1], x[1]))
    assert np.array_equal(i1["n_labels"], i2["n_labels"])
    flipped = np.ascontiguousarray(base[:, :, ::-1])
    rf, _ = eng.analyze(flipped, mm_r, mm_c)
    for k in range(4):
        want = O.detect_debug(flipped[k], mm_r, mm_c)["strips"]
        assert _strips(rf[k][0]) == want
    eng.close()
