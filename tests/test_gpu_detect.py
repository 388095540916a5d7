"""GPU parity of the detect path against the oracle, through the C ABI (liblfa_hip.so via ctypes).

Bar: bit-exact Otsu threshold, mask, opened mask, label image, bbox/area table, accepted strips and crops."""
import os

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

from oracle import lfa_oracle as O


@pytest.fixture(scope="module")
def capi():
    import image_analysis_amd._capi as capi
    return capi


@pytest.fixture(scope="module")
def eng(capi):
    e = capi.Engine(device=0, max_h=2048, max_w=2048, max_c=3, max_batch=8)
    yield e
    e.close()


def gui_params(image):
    mm_c = int(image.shape[1] / 40)        # detection.py:457-460
    return int(mm_c * 20), mm_c


ALLOW_AMBIGUOUS = False      # tools/detect_fuzz.py sets it: an image with a pixel exactly on a histogram edge is flagged, and still checked
AMBIGUOUS_SEEN = [0]


def check_detect(capi, eng, image, mm_r, mm_c, L=10, batch_index=0, res=None):
    d = O.detect_debug(image, mm_r, mm_c, L)
    if res is None:
        strips, info = eng.detect(image[None], mm_r, mm_c, L)
    else:
        strips, info = res
    H, W = image.shape[:2]
    inf = info[batch_index]
    colour = image.ndim == 3
    assert (inf["status"] & 3) == 0
    AMBIGUOUS_SEEN[0] += int(inf["ambiguous"] != 0)
    # a pixel within 2^-40 of a histogram edge: Engine.detect has decided that image again with the host's pow() (status bit 2)
    assert inf["ambiguous"] == 0 or (inf["status"] & 4)
    thr = capi.host_threshold(inf, colour)
    assert thr == d["threshold"], (thr, d["threshold"])
    if len(info) > 1 and (np.asarray(info["status"]) & 4).any():
        # the handle's last call was the host-pow run of a sub-batch: the intermediates of this batch's images are not at batch_index
        got = [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips[batch_index]]
        assert got == d["strips"] and inf["n_labels"] == d["n_labels"]
        return d
    mask = eng.debug(batch_index, capi.DBG_MASK, H, W).astype(bool)
    assert np.array_equal(mask, d["mask"])
    opened = eng.debug(batch_index, capi.DBG_OPENED, H, W).astype(bool)
    assert np.array_equal(opened, d["opened"])
    labels = eng.debug(batch_index, capi.DBG_LABELS, H, W)
    assert labels.dtype == np.int32
    assert np.array_equal(labels, d["labels"])
    assert inf["n_labels"] == d["n_labels"]
    assert inf["label00"] == int(d["background"])
    table = eng.debug_table(batch_index, d["n_labels"])
    assert np.array_equal(table.astype(np.int64), d["table"])
    got = [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips[batch_index]]
    assert got == d["strips"]
    assert [int(s["label"]) for s in strips[batch_index]] == d["strip_labels"]
    assert [int(s["area"]) for s in strips[batch_index]] == d["strip_areas"]
    assert inf["n_accepted"] == len(d["accepted"])
    return d


def test_excel2_frame_uint16(capi, eng, golden_dir):
    frame = np.array(Image.open(os.path.join(golden_dir, "excel2_frame.png")))
    mm_r, mm_c = gui_params(frame)
    d = check_detect(capi, eng, frame, mm_r, mm_c)
    assert d["threshold"] == 5086 and d["n_labels"] == 1149 and len(d["strips"]) == 40
    assert [324, 1344, 806, 857] in d["strips"]            # the crop the reference exported (images/excel2.xlsx)


def test_image1_gray_and_bgr(capi, eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "image1_decoded.npz"))["gray"]
    mm_r, mm_c = gui_params(g)
    check_detect(capi, eng, g, mm_r, mm_c)
    bgr = np.repeat(g[:, :, None], 3, axis=2)               # what cv2.imread returns for a grayscale JPEG
    check_detect(capi, eng, bgr, mm_r, mm_c)


@pytest.mark.parametrize("key", ["image2", "image4", "image7", "test_strip_buggy_bgr", "wick_crop", "image5_crop", "austin_u16_crop"])
def test_reference_sample_images(capi, golden_dir, key):
    """The reference's own sample inputs (tests/golden/real_samples.npz, decoded in the build container): photo-like data,
    thresholds on populated grey levels, hundreds of labels.  Grey JPEGs go through the 2-D path and -- as three equal
    channels, which is what cv2.imread hands the reference -- through the colour path, with the class plane left to the
    default choice and forced."""
    img = np.load(os.path.join(golden_dir, "real_samples.npz"))[key]
    H, W = img.shape[:2]
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=1)
    try:
        mm_r, mm_c = gui_params(img)
        check_detect(capi, eng, img, mm_r, mm_c)
        if img.dtype == np.uint8:
            bgr = img if img.ndim == 3 else np.ascontiguousarray(np.repeat(img[:, :, None], 3, axis=2))
            for flags in (0, 4):
                eng.set_debug_flags(flags)
                check_detect(capi, eng, bgr, mm_r, mm_c)
            eng.set_debug_flags(0)
        else:
            # 16-bit frame: the same handle again -- one-pass histogram, then the class plane centred on its own
            # threshold (default choice, then forced)
            for flags in (0, 0, 4, 2):
                eng.set_debug_flags(flags)
                check_detect(capi, eng, img, mm_r, mm_c)
            eng.set_debug_flags(0)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_synthetic_rgb(capi, eng, seed):
    from image_analysis_amd.synth import synth_rgb
    img = synth_rgb(384, 512, 8, seed)
    mm_r, mm_c = gui_params(img)
    d = check_detect(capi, eng, img, mm_r, mm_c)
    assert len(d["strips"]) >= 4


@pytest.mark.parametrize("seed", [4, 5])
def test_colour_float64_fallback_path(capi, eng, seed):
    """The float64 compare path (taken when a threshold is 'critical' for the integer form) must give the same answer."""
    from image_analysis_amd.synth import synth_rgb
    img = synth_rgb(256, 384, 6, seed)
    mm_r, mm_c = gui_params(img)
    eng.set_debug_flags(1)
    try:
        check_detect(capi, eng, img, mm_r, mm_c)
        check_detect(capi, eng, np.ascontiguousarray(img[:75, :100]), 12, 3)       # scalar (ragged) kernels
    finally:
        eng.set_debug_flags(0)


def test_colour_bin_plane_paths_agree(capi, eng):
    """k_mask decides from the class plane k_hist_rgb wrote (class above the Otsu bin's: set, below: clear, equal: exact compare on the image).
    Images with few distinct gray levels put many pixels INTO the threshold's class; the plane path with its fix-up loop
    (debug flag 4 forces it), the plane-less path (debug flag 2) and the default choice must all reproduce the oracle."""
    rng = np.random.default_rng(2024)
    cases = []
    base = rng.integers(50, 61, size=(192, 256), dtype=np.uint8)                 # 11 levels: one value per occupied bin
    img = np.stack([base, base, base], axis=2)
    img[20:170, 40:52] = 58
    img[20:170, 100:112] = 60
    cases.append(img)
    two = np.full((192, 256, 3), 90, np.uint8)                                   # two levels + a sprinkle of a third
    two[10:180, 60:75] = 91
    two[10:180, 140:150] = (91, 90, 92)
    two[rng.random((192, 256)) < 0.01] = 89
    cases.append(two)
    wide = rng.integers(0, 256, size=(192, 256, 3), dtype=np.uint8)              # every bin populated, channels independent
    wide[30:160, 80:96] = (250, 251, 249)
    cases.append(wide)
    for img in cases:
        for flags in (0, 2, 4):              # default (plane only when few pixels share the class), no plane, plane always
            eng.set_debug_flags(flags)
            try:
                d = check_detect(capi, eng, img, 60, 10)
            finally:
                eng.set_debug_flags(0)
        assert d["mask"].any() and not d["mask"].all()


def _valley_image(rng, H=192, W=256):
    """dark background, bright strips and five stray pixels in between: the Otsu bin is the strays' bin, and only they
    share its class"""
    g = rng.integers(30, 41, size=(H, W)).astype(np.uint8)
    for c0 in (40, 120, 200):
        g[20:170, c0:c0 + 12] = rng.integers(190, 210, size=(150, 12))
    for k in range(5):
        g[5 + k, 3 + 17 * k] = 60
    return np.ascontiguousarray(np.stack([g, g, g], axis=2))


def _photo_image(rng, H=192, W=256):
    """smooth ramp under the strips: every grey level is populated, thousands of pixels share the threshold's class"""
    ramp = np.linspace(20, 235, W)[None, :] + rng.normal(0, 3, size=(H, W))
    g = np.clip(ramp, 0, 255).astype(np.uint8)
    g[20:170, 60:72] = 250
    return np.ascontiguousarray(np.stack([g, g, g], axis=2))


def _few_share_the_class(img):
    """the rule of k_otsu_rgb, from the oracle's histogram: pixels in the 16 bins of the Otsu bin's class <= H*W / 2048"""
    v = O.adjust_gamma(O.rgb2gray(img), 0.4)
    counts, centers = O.histogram_image_range(v)
    best = O.otsu_from_hist(counts, centers)[1]
    return int(counts[(best & ~15):(best & ~15) + 16].sum()) * 2048 <= img.shape[0] * img.shape[1]


def test_class_plane_is_chosen_per_image_and_per_call(capi):
    """k_otsu_rgb lets an image use the class plane only when few pixels share the threshold's class; a handle stops
    writing the plane while most images of its last completed call could not use it, and starts again when they can.
    Every intermediate is checked against the oracle whichever way the choice goes."""
    rng = np.random.default_rng(99)
    eng = capi.Engine(device=0, max_h=192, max_w=256, max_batch=4)
    eng.set_debug_flags(2048 | 8192)     # (these images have three equal channels: keep the handle out of the equal-channel mode,
                                         #  which needs no class plane, and out of the predicted-threshold mode, which writes another
                                         #  plane -- this test is about the class plane)
    try:
        valley = np.stack([_valley_image(rng) for _ in range(4)])
        photo = np.stack([_photo_image(rng) for _ in range(4)])
        assert all(_few_share_the_class(v) for v in valley) and not any(_few_share_the_class(p) for p in photo)
        mixed = np.stack([valley[0], photo[0], valley[1], valley[2]])

        def run(batch):
            res = eng.detect(batch, 60, 10)
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 60, 10, batch_index=i, res=res)
            return [eng.debug_used_plane(i) for i in range(len(batch))]
        assert run(mixed) == [True, False, True, True]              # per image
        assert run(photo) == [False] * 4                            # 3 of 4 could before: plane still written, nobody uses it
        assert run(valley) == [False] * 4                           # last call: 0 of 4 could -> this call wrote no plane
        assert run(valley) == [True] * 4                            # last call: 4 of 4 could have -> plane back on
        eng.set_debug_flags(4 | 2048 | 8192)
        assert run(photo) == [True] * 4                             # forced: the fix-up loop settles the pixels of the class
        eng.set_debug_flags(0)
    finally:
        eng.close()


def _u16_frame(rng, top, H=160, W=256):
    g = rng.integers(top // 40, top // 8, size=(H, W)).astype(np.uint16)
    for c0 in (30, 110, 190):
        blk = g[15:140, c0:c0 + 12]
        blk[...] = rng.integers(top // 2, top, size=blk.shape)
    return g


@pytest.mark.parametrize("top", [4095, 16383, 16384, 65535])
def test_uint16_one_pass_and_two_pass_histograms_agree(capi, eng, top):
    """uint16 frames: the histogram indexed by pixel value (no min/max pass; values >= 16384 counted with global atomics)
    and the two-pass form (min/max first, bins relative to the minimum) give the oracle's threshold, mask, labels, strips."""
    rng = np.random.default_rng(top)
    img = _u16_frame(rng, top)
    img[0, 0] = top                                                 # the largest value is present
    for flags in (8, 16):
        eng.set_debug_flags(flags)
        try:
            check_detect(capi, eng, img, 60, 10)
            assert eng.debug_one_pass() == (flags == 16)
        finally:
            eng.set_debug_flags(0)
    const = np.full((64, 64), 1234, np.uint16)                      # constant frame: one occupied bin
    for flags in (8, 16):
        eng.set_debug_flags(flags)
        try:
            check_detect(capi, eng, const, 10, 3)
        finally:
            eng.set_debug_flags(0)


def test_uint16_one_pass_is_chosen_per_call(capi):
    """A handle drops the min/max pass once its last completed uint16 call held no value of 16384 or more, and takes it
    up again after a call that did."""
    rng = np.random.default_rng(5)
    eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=2)
    try:
        low = np.stack([_u16_frame(rng, 4095) for _ in range(2)])
        high = np.stack([_u16_frame(rng, 60000) for _ in range(2)])

        def run(batch):
            res = eng.detect(batch, 60, 10)
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 60, 10, batch_index=i, res=res)
            return eng.debug_one_pass()
        assert run(low) is False            # nothing known yet: two passes
        assert run(low) is True             # the last call stayed below 16384
        assert run(high) is True            # still the one-pass form: exact, the high values go through global atomics
        assert run(high) is False           # ... and the handle has learnt
        assert run(low) is False
        assert run(low) is True
    finally:
        eng.close()


def _u16_separated(rng, H=160, W=256, noise=False):
    """dark background, a dozen pixels above it, and bright strips (noise=True: values all over the range)"""
    if noise:
        g = rng.integers(0, 4096, size=(H, W)).astype(np.uint16)
    else:
        g = rng.integers(90, 110, size=(H, W)).astype(np.uint16)
        g[rng.integers(0, H, 12), rng.integers(0, W, 12)] = 150            # the threshold lands here, in a nearly empty range
    for c0 in (30, 110, 190):
        blk = g[15:140, c0:c0 + 12]                                            # (smaller frames: what fits)
        blk[...] = rng.integers(2000, 3000, size=blk.shape)
    return g


@pytest.mark.parametrize("W", [256, 250])
def test_uint16_class_plane_paths_agree(capi, W):
    """uint16 frames, one-pass histogram: the mask decided from the 4-bit class plane (forced with debug flag 4: first
    without a predicted threshold, so that nearly every pixel is of the threshold's class and takes the exact compare,
    then with the classes centred on the last call's threshold) and the mask compared on the frame itself (flag 2) both
    give the oracle's threshold, mask, labels and strips.  A width that is no multiple of 16 has no plane."""
    rng = np.random.default_rng(21)
    eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=3)
    try:
        batch = np.stack([_u16_separated(rng, W=W), _u16_frame(rng, 4095, W=W), _u16_separated(rng, W=W, noise=True)])
        for flags in (16 | 4, 16 | 4, 16 | 4, 16 | 2, 16):
            eng.set_debug_flags(flags | 8192)               # (8192: not the predicted-threshold bit plane -- this test is about the class plane)
            res = eng.detect(batch, 60, 10)
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 60, 10, batch_index=i, res=res)
                if flags & 4:
                    assert eng.debug_used_plane(i) == (W % 16 == 0)
                if flags & 2:
                    assert not eng.debug_used_plane(i)
            eng.sync() if hasattr(eng, "sync") else None
    finally:
        eng.close()


def test_uint16_class_plane_is_chosen_per_image_and_per_call(capi):
    """The uint16 class plane needs the one-pass histogram and a threshold to centre its classes on (the mean of the
    handle's last call).  Per image it is used when the threshold's class is narrow and nearly empty; per call it is
    written while at least half of the last call's frames could use it.  The test carries a host model of those rules."""
    rng = np.random.default_rng(22)
    eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=2)
    eng.set_debug_flags(8192)                               # (not the predicted-threshold bit plane, which takes the class plane's place)
    st = {"one_pass": False, "have_pred": False, "plane_on": True, "mean": 0, "max": 65535, "spread": 0}
    seen = set()
    try:
        sep = np.stack([_u16_separated(rng) for _ in range(2)])
        noisy = np.stack([_u16_separated(rng, noise=True) for _ in range(2)])

        def run(batch):
            res = eng.detect(batch, 60, 10)
            info = res[1]
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 60, 10, batch_index=i, res=res)
            used = [eng.debug_used_plane(i) for i in range(len(batch))]
            pred = st["one_pass"] and st["have_pred"]
            wide = 4 if st["max"] < 4096 else 5 if st["max"] < 8192 else 6
            ps = 0 if st["spread"] <= 10 else wide          # thresholds of the last call close together: one value per class
            pa = max(0, st["mean"] - (7 if ps == 0 else (15 << wide) // 2))
            coulds = []
            for i in range(len(batch)):
                x = int(info[i]["x_thr"])
                q = min(max(x - pa, 0) >> ps, 15)
                settled = q <= 14 and x == pa + ((q + 1) << ps) - 1      # the class of x_thr ends at x_thr: clear as a whole
                lo = pa + (q << ps)
                n_in = int(((batch[i] >= lo) & (batch[i] < lo + (1 << ps))).sum())
                coulds.append(bool(pred and (settled or (1 <= q <= 14 and n_in * 2048 <= batch[i].size))))
            expect = [bool(pred and st["plane_on"] and c) for c in coulds]
            assert used == expect, (used, expect, st, ps, pa)
            seen.update(used)
            if pred:
                st["plane_on"] = 2 * sum(coulds) >= len(batch)
            thr = [int(info[i]["x_thr"]) for i in range(len(batch))]
            st["max"] = int(batch.max()); st["one_pass"] = st["max"] < 16384
            st["mean"] = sum(thr) // len(batch); st["spread"] = max(thr) - min(thr); st["have_pred"] = True
            return used
        assert run(sep) == [False, False]           # nothing known: two-pass histogram, no plane
        assert run(sep) == [True, True]             # one pass, classes centred on the first call's thresholds
        run(noisy)                                  # plane written, but the threshold's class is crowded: frames re-read
        run(sep); run(sep); run(sep)
        mixed = np.stack([sep[0], noisy[0]])
        for _ in range(3):
            run(mixed)
        assert seen == {True, False}
    finally:
        eng.close()


def test_one_handle_across_shapes_and_dtypes(capi):
    """What a handle learnt from one kind of input (one-pass histogram, class-plane centre, window length) must not leak
    into the next: frames of other sizes, colour images and 8-bit 2-D images in between, bigger batches after smaller."""
    rng = np.random.default_rng(24)
    eng = capi.Engine(device=0, max_h=192, max_w=256, max_c=3, max_batch=3)
    try:
        a = np.stack([_u16_separated(rng) for _ in range(2)])                       # 160 x 256
        b = np.stack([_u16_separated(rng, H=96, W=128, noise=True) for _ in range(3)])
        c = np.stack([_u16_frame(rng, 60000, H=192, W=240)])                        # high values: back to two passes; W % 16 == 0
        rgb = rng.integers(0, 256, size=(2, 64, 64, 3), dtype=np.uint8)
        rgb[:, 10:50, 20:28] = 250
        g8 = rng.integers(0, 100, size=(1, 100, 90), dtype=np.uint8)
        g8[:, 20:80, 30:36] = 220
        for batch in (a, a, b, rgb, a, b, b, g8, c, c, a, a, a, b):
            res = eng.detect(batch, 40, 6)
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 40, 6, batch_index=i, res=res)
    finally:
        eng.close()


def test_uint16_series_with_a_stable_threshold(capi):
    """Frames of one scene: thresholds within a few values of each other, lying inside the background noise (a crowded
    class whatever its width).  After one call the classes are one value wide around the mean threshold, the threshold is
    the last value of its class, and the mask comes from the plane alone -- same results as from the frames."""
    rng = np.random.default_rng(23)
    base = _u16_separated(rng, noise=True)
    series = np.stack([np.ascontiguousarray(np.roll(base, (3 * k, 5 * k), (0, 1))) for k in range(3)])
    series[1, 0, 0] += 1; series[2, 5, 5] -= 1                      # not quite the same histogram
    eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=3)
    try:
        used = []
        for call in range(4):
            res = eng.detect(series, 60, 10)                        # (threshold inside the noise: the handle does not try the bit plane)
            for i in range(3):
                check_detect(capi, eng, series[i], 60, 10, batch_index=i, res=res)
            used.append([eng.debug_used_plane(i) for i in range(3)])
        thr = [int(res[1][i]["x_thr"]) for i in range(3)]
        assert max(thr) - min(thr) <= 10
        assert used[0] == [False] * 3                               # two-pass histogram
        assert used[-1] == [True] * 3, used
    finally:
        eng.close()


def test_colour_extreme_values(capi, eng):
    """black and saturated pixels, all 256 levels per channel, strongly different channels"""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, size=(128, 256, 3), dtype=np.uint8)
    img[:40] = 0
    img[40:90, 30:60] = 255
    img[90:, :, 0] = 255
    img[90:, :, 2] = 0
    check_detect(capi, eng, img, 30, 8)
    img2 = np.zeros((64, 64, 3), np.uint8)
    img2[10:50, 20:30] = (255, 255, 255)
    check_detect(capi, eng, img2, 20, 5)


def test_synthetic_u16_small(capi, eng):
    from image_analysis_amd.synth import synth_u16
    img = synth_u16(512, 512, 10, 5)
    mm_r, mm_c = gui_params(img)
    check_detect(capi, eng, img, mm_r, mm_c)


def test_full_range_uint16(capi, eng):
    rng = np.random.default_rng(11)
    img = (rng.random((256, 320)) * 65535).astype(np.uint16)
    img[40:200, 50:80] = 60000
    img[0, 0] = 0
    img[1, 1] = 65535
    check_detect(capi, eng, img, 40, 8)


@pytest.mark.parametrize("shape,dtype", [((100, 75), np.uint8), ((75, 100, 3), np.uint8), ((130, 97), np.uint16),
                                          ((65, 33), np.uint8), ((64, 48, 3), np.uint8), ((1, 40), np.uint8),
                                          ((40, 1), np.uint16), ((129, 16), np.uint8)])
def test_ragged_shapes(capi, eng, shape, dtype):
    rng = np.random.default_rng(sum(shape))
    hi = 255 if dtype == np.uint8 else 4000
    base = np.repeat(rng.random((max(1, shape[0] // 8 + 1),) + shape[1:2]) < 0.5, 8, axis=0)[:shape[0]]
    img = np.where(base, int(hi * 0.8), int(hi * 0.1)).astype(np.float64)
    if len(shape) == 3:
        img = img[:, :, None] + rng.integers(-3, 4, size=shape)
    else:
        img = img + rng.integers(-3, 4, size=shape)
    img = np.clip(img, 0, hi).astype(dtype)
    check_detect(capi, eng, img, 12, 3)


@pytest.mark.parametrize("dtype,c", [(np.uint8, 0), (np.uint16, 0), (np.uint8, 3)])
def test_constant_image_gives_no_strips(capi, eng, dtype, c):
    shape = (64, 80) + ((3,) if c else ())
    img = np.full(shape, 77, dtype=dtype)
    d = check_detect(capi, eng, img, 20, 4)
    assert d["strips"] == [] and d["n_labels"] == 0


def test_border_runs_and_run_lengths(capi, eng):
    """runs of 4/5/6 at the top and 5/6/7 at the bottom, interior runs of 9/10/11 (opening keeps >= 10, border rule)."""
    img = np.zeros((96, 160), np.uint8)
    for k, ln in enumerate((4, 5, 6)):
        img[0:ln, 10 + 12 * k:14 + 12 * k] = 200
    for k, ln in enumerate((5, 6, 7)):
        img[96 - ln:96, 60 + 12 * k:64 + 12 * k] = 200
    for k, ln in enumerate((9, 10, 11)):
        img[30:30 + ln, 100 + 12 * k:104 + 12 * k] = 200
    img[40:90, 20:30] = 200          # something big enough to be a strip
    d = check_detect(capi, eng, img, 30, 6)
    assert d["n_labels"] >= 5


def test_component_at_origin_raises_background_label(capi, eng):
    """labeled[0,0] != 0 -> `background_color` is a label id (detection.py:67); means must exceed it."""
    img = np.zeros((120, 200), np.uint8)
    img[0:40, 0:30] = 210            # touches (0,0): label 1, background_color = 1.0
    img[50:110, 80:110] = 220
    img[50:110, 150:180] = 1         # dark, below threshold
    d = check_detect(capi, eng, img, 30, 10)
    assert d["background"] == 1.0


def test_overlapping_and_out_of_image_boxes(capi, eng):
    """expanded boxes overlap (first in label order wins) and extend past the image (numpy clips the crop)."""
    img = np.zeros((200, 300), np.uint8)
    img[150:190, 20:40] = 200        # maxr = 150+120 = 270 > H
    img[20:60, 25:45] = 210          # expanded box reaches down to 140: no overlap with the first? 20+120=140 < 150
    img[30:70, 60:80] = 205          # overlaps box 2 after column expansion (25+40=65 >= 60)
    img[100:140, 280:299] = 215      # maxc = 280+40 > W
    d = check_detect(capi, eng, img, 120, 40)
    assert len(d["accepted"]) < d["n_labels"]


@pytest.mark.parametrize("L", [1, 2, 3, 7, 11, 16])
def test_line_length_variants(capi, eng, L):
    rng = np.random.default_rng(L)
    base = np.repeat(rng.random((30, 120)) < 0.45, 5, axis=0)
    img = np.where(base, 190, 20).astype(np.uint8)
    img[3, :] = 20
    check_detect(capi, eng, img, 25, 5, L=L)


def test_many_small_components(capi, eng):
    rng = np.random.default_rng(42)
    base = np.repeat(rng.random((40, 600)) < 0.35, 10, axis=0)
    img = (np.where(base, 3000, 100) + rng.integers(0, 50, size=base.shape)).astype(np.uint16)
    d = check_detect(capi, eng, img, 40, 15)
    assert d["n_labels"] > 300


def test_batch_of_distinct_images_matches_per_image(capi, eng):
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(256, 384, 6, 100 + i) for i in range(5)])
    mm_r, mm_c = gui_params(imgs[0])
    res = eng.detect(imgs, mm_r, mm_c, 10)
    for i in range(5):
        check_detect(capi, eng, imgs[i], mm_r, mm_c, batch_index=i, res=res)


def test_dropin_detect_returns_views_equal_to_oracle_crops(golden_dir):
    import image_analysis_amd.detection as det
    from image_analysis_amd.synth import synth_rgb
    img = synth_rgb(384, 512, 8, 9)
    det.min_maxc = int(img.shape[1] / 40)
    det.min_maxr = int(det.min_maxc * 20)
    crops = det.detect_lateral_flow_tests(img)
    want = O.detect_lateral_flow_tests(img, det.min_maxr, det.min_maxc)
    assert len(crops) == len(want) > 0
    for a, b in zip(crops, want):
        assert a.base is not None and np.shares_memory(a, img)
        assert a.shape == b.shape and np.array_equal(a, b)
    with pytest.raises(ValueError):
        det.detect_lateral_flow_tests(np.zeros((4, 4, 4, 4), np.uint8))
    with pytest.raises(ValueError):
        det.detect_lateral_flow_tests(np.zeros((8, 8, 4), np.uint8))


@pytest.mark.parametrize("shape", [(70, 8200), (4100, 48), (4100, 8200)])
def test_very_wide_and_very_tall_images(capi, shape):
    """k_components keeps the columns' offsets (W + 1 <= 8192) and the per-row root counts (H <= 4096) in LDS; wider and
    taller images go through the global arrays instead -- same labels, boxes and strips."""
    H, W = shape
    rng = np.random.default_rng(H + W)
    img = rng.integers(10, 60, size=(H, W)).astype(np.uint8)
    for _ in range(40):                                             # vertical bars, some of them touching, some at the borders
        r0, c0 = int(rng.integers(0, H - 12)), int(rng.integers(0, W - 4))
        img[r0:r0 + int(rng.integers(12, max(13, H // 3))), c0:c0 + int(rng.integers(1, 4))] = int(rng.integers(180, 255))
    img[:30, W - 2:] = 250
    img[H - 30:, :2] = 250
    img[rng.random((H, W)) < 0.001] = 255                           # salt: removed by the opening
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=1)
    try:
        check_detect(capi, eng, img, H // 2, max(1, W // 40))
    finally:
        eng.close()


@pytest.mark.parametrize("L", [1, 2, 3, 9, 10, 11, 17, 40, 63, 64, 65, 100])
def test_opening_of_noisy_masks(capi, eng, L):
    """Masks that are mostly noise (many short vertical runs per 64-row word: the run walk first drops, bit-parallel, the
    runs that cannot survive) at several densities, heights around the word size and line lengths around it: opened
    mask, labels and strips against scipy's erosion + dilation."""
    rng = np.random.default_rng(300 + L)
    for H in (63, 64, 65, 130, 200):
        for density in (0.3, 0.5, 0.8, 0.95):
            W = int(rng.choice([48, 67]))
            img = np.where(rng.random((H, W)) < density, 200, 20).astype(np.uint8)
            img[:, 5] = 200                                             # a full column, a run from the top, one to the bottom
            img[: min(H, L // 2 + 1), 9] = 200
            img[max(0, H - (L - 1) // 2 - 1):, 13] = 200
            img[:, 8] = 20; img[:, 10] = 20; img[:, 12] = 20; img[:, 14] = 20
            check_detect(capi, eng, img, int(rng.integers(0, H)), int(rng.integers(0, W // 2)), L)


@pytest.mark.parametrize("seed", range(12))
def test_random_small_images_against_oracle(capi, eng, seed):
    """Seeded fuzz: random shapes (down to a few pixels), dtypes, blob layouts, thresholds of every kind, line lengths and
    min_maxr / min_maxc -- every intermediate compared with the oracle."""
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        kind = rng.integers(0, 3)                                   # 0: uint8 2-D, 1: uint16 2-D, 2: colour
        H = int(rng.integers(1, 200)) if rng.random() < 0.8 else int(rng.integers(1, 12))
        W = int(rng.integers(1, 200)) if rng.random() < 0.8 else int(rng.integers(1, 12))
        if kind != 0 and rng.random() < 0.5:
            W = max(16, (W // 16) * 16)                             # the aligned paths (class plane)
        top = 255 if kind != 1 else int(rng.choice([255, 4095, 65535]))
        base = rng.integers(0, max(2, top // 3), size=(H, W))
        img = base.astype(np.int64)
        for _b in range(int(rng.integers(0, 6))):                   # bright vertical-ish blobs
            r0, c0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            hh, ww = int(rng.integers(1, max(2, H))), int(rng.integers(1, max(2, W // 4 + 1)))
            img[r0:r0 + hh, c0:c0 + ww] += int(rng.integers(top // 3, top))
        if rng.random() < 0.3:
            img[rng.random((H, W)) < 0.02] = top                    # salt
        img = np.clip(img, 0, top)
        if kind == 0:
            arr = img.astype(np.uint8)
        elif kind == 1:
            arr = img.astype(np.uint16)
        else:
            ch = [np.clip(img + rng.integers(-3, 4, size=(H, W)), 0, 255) for _c in range(3)]
            if rng.random() < 0.35:                                 # three equal channels (cv2.imread of a grey file): the
                ch = [ch[0]] * 3                                    # equal-channel mode, entered by the handle itself or forced
            arr = np.ascontiguousarray(np.stack(ch, axis=2).astype(np.uint8))
            if rng.random() < 0.1:
                arr[int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(0, 3))] ^= 1      # ... but for one pixel
        L = int(rng.choice([1, 2, 5, 10, 10, 10, 17]))
        mm_r, mm_c = int(rng.integers(0, max(1, H))), int(rng.integers(0, max(1, W // 2 + 1)))
        # colour: default choice of the class plane, or forced; uint16: default, or one-pass histogram with the plane forced
        # (its classes centred on whatever threshold the shared handle met last), or the two-pass histogram
        eng.set_debug_flags(int(rng.choice([0, 4, 0, 1024, 1024 | 1, 1024 | 4096, 8192, 16384, 32768, 32768 | 1, 65536, 32768 | 65536])) if kind == 2 else (int(rng.choice([0, 16 | 4, 8, 16 | 32768, 16 | 8192, 16 | 16384])) if kind == 1 else 0))
        try:
            check_detect(capi, eng, arr, mm_r, mm_c, L)
        finally:
            eng.set_debug_flags(0)


@pytest.mark.parametrize("W", [1, 2, 3])
def test_colour_images_one_to_three_pixels_wide(capi, eng, W):
    """numpy evaluates rgb2gray's `rgb @ coeffs` row by row of the image: a dot product for one-pixel-wide images, a
    matrix-vector product otherwise -- two rounding orders on most hosts (capi.detect_gray_variant(width)).  The binding
    sets the handle's order from the width of the call's images; found by tools/detect_fuzz.py (seed 408: a 142 x 1 image
    whose brightest pixel's gray value differs in the last bit between the two orders)."""
    rng = np.random.default_rng(1408)
    for _ in range(30):
        H = int(rng.integers(20, 200))
        img = rng.integers(0, 90, size=(H, W, 3)).astype(np.uint8)
        r0 = int(rng.integers(0, H // 2))
        img[r0:r0 + int(rng.integers(5, H // 2)), :, :] += 150
        check_detect(capi, eng, np.ascontiguousarray(img), int(rng.integers(0, H)), 0, int(rng.choice([1, 3, 10])))
    wide = rng.integers(0, 256, size=(64, 80, 3)).astype(np.uint8)      # and back to an ordinary width on the same handle
    check_detect(capi, eng, wide, 10, 2)


def test_uint16_colour_images_detect_profile_and_peaks():
    """16-bit RGB images (what tifffile hands the reference for an RGB16 TIFF): the scalar float64 paths of every kernel.
    detect against the oracle (threshold, boxes), then the whole batch call against detect + image_analysis of the oracle;
    the detect stage is also held to the vendored-skimage fixtures in test_gpu_skimage.py (cases *_rgb16)."""
    import image_analysis_amd.detection as detection
    from image_analysis_amd.synth import synth_rgb
    from oracle import lfa_oracle as O
    rng = np.random.default_rng(3)
    ims = []
    for seed in (41, 42):
        c8 = synth_rgb(192, 256, 4, seed)
        ims.append(((c8.astype(np.uint16) << 8) | rng.integers(0, 256, size=c8.shape).astype(np.uint16)))
    ims = np.stack(ims)
    mm_c = int(256 / 40)
    detection.min_maxr, detection.min_maxc = mm_c * 20, mm_c
    try:
        got = detection.analyze_batch(ims)
        for im, res in zip(ims, got):
            want_crops = O.detect_lateral_flow_tests(im, mm_c * 20, mm_c)
            assert len(res) == len(want_crops) > 0
            for (crop, prof, peaks), wc in zip(res, want_crops):
                assert crop.shape == wc.shape and np.array_equal(crop, wc)
                wp, wpk = O.image_analysis(wc, False)
                assert prof.tobytes() == wp.tobytes()
                assert [(s, e) for s, e, _ in peaks] == [(s, e) for s, e, _ in wpk]
                assert all(np.float64(a).tobytes() == np.float64(b).tobytes() for (_, _, a), (_, _, b) in zip(peaks, wpk))
            single = detection.detect_lateral_flow_tests(im)
            assert [c.shape for c in single] == [c.shape for c in want_crops]
    finally:
        detection.min_maxr, detection.min_maxc = 0, 0


def _edge_tie_image(seed=0):
    """Equal-channel colour image with gmin = 0 (a black pixel), gmax = gray(128,128,128) and pixels (4,4,4): their gray is
    exactly gmax / 32, so their gamma value is vmax / 4 = edge 64 of the 256-bin histogram whenever pow() scales cleanly --
    a pixel ON a histogram edge, whose bin depends on the last bit of pow().  Bright strips on a dark ground so that the
    detect path has something to find."""
    rng = np.random.default_rng(seed)
    H, W = 192, 256
    g = rng.integers(0, 20, size=(H, W)).astype(np.uint8)
    for c0 in (30, 90, 150, 210):
        g[20:170, c0:c0 + 14] = rng.integers(100, 129, size=(150, 14))
    g[0, 0] = 0
    g[1, 1] = 128
    g[175:180, 10:200] = 4                                  # 950 pixels on the edge
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def test_pixels_on_a_histogram_edge_are_decided_with_the_hosts_pow(capi):
    """`LfaImageInfo.ambiguous` was counted and then ignored.  The reference bins pow(gray, 0.4) computed by the
    HOST's numpy (exposure.py:679, histograms.py:800-859); the device's own pow() may differ in the last bit, which matters
    only for pixels within 2^-40 of an edge -- exactly what `ambiguous` counts.  Engine.detect / analyze now run such an
    image again with tables built by the host's numpy (lfa_set_host_pow): histogram, threshold, mask and strips equal the
    oracle's, whatever the device's pow() says -- shown here by also moving every device edge one double up / down (debug
    flags 256 / 512)."""
    img = _edge_tie_image()
    mm_r, mm_c = gui_params(img)
    H, W = img.shape[:2]
    want_hist, _ = O.histogram_image_range(O.adjust_gamma(O.rgb2gray(img), 0.4))
    e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=2)
    raw_equal = {}
    try:
        for flags in (0, 256, 512):
            e.set_debug_flags(flags)
            # raw run: the device's own edges
            _strips, info = e.detect(img[None], mm_r, mm_c, resolve_ambiguous=False)
            assert info[0]["ambiguous"] >= 950 and (info[0]["status"] & 4) == 0
            raw_equal[flags] = np.array_equal(e.debug_hist(0), want_hist)
            # product path: decided again with the host's pow
            d = check_detect(capi, e, img, mm_r, mm_c)
            assert np.array_equal(e.debug_hist(0), want_hist)
            strips, info = e.detect(np.stack([img, img[::-1].copy()]), mm_r, mm_c)
            assert (info["status"] & 4).all() and (info["ambiguous"] >= 950).all()
            check_detect(capi, e, img, mm_r, mm_c, res=(strips, info), batch_index=0)
            check_detect(capi, e, np.ascontiguousarray(img[::-1]), mm_r, mm_c, res=(strips, info), batch_index=1)
            # the whole job: profiles and peaks of the strips found in the host-pow run
            res, inf = e.analyze(img[None], mm_r, mm_c)
            assert inf[0]["status"] & 4
            got = [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in res[0][0]]
            assert got == d["strips"] and len(got) >= 1
            for k, (a, b, c, e2) in enumerate(d["strips"]):
                wp, wk = O.image_analysis(img[a:b, c:e2], False)
                assert np.array_equal(res[0][1][k], wp)
        # the 950 pixels sit on edge 64 or one double beside it: of the device's own edges as computed, moved one double up and
        # moved one double down, at least one set bins them differently from the host (which one depends on the two pow()s)
        assert not all(raw_equal.values()), raw_equal
        # tables are used once, and only by a call with as many images
        e.set_debug_flags(0)
        _s, i1 = e.detect(img[None], mm_r, mm_c, resolve_ambiguous=False)
        assert (i1[0]["status"] & 4) == 0
        e._set_host_pow(i1)
        t = capi.to_device(np.stack([img, img]), e.device)
        with pytest.raises(capi.LfaError):
            e.detect_raw(t, capi.LFA_U8, 2, H, W, 3, mm_r, mm_c, 10, 16)
        _s, i2 = e.detect(img[None], mm_r, mm_c, resolve_ambiguous=False)     # withdrawn by the failed call
        assert (i2[0]["status"] & 4) == 0
    finally:
        e.close()


def test_ambiguous_images_in_a_stream_and_in_analyze_batch(capi, monkeypatch):
    """The batch entry points (detection.analyze_batch, frames.analyze_stream) and the drop-in function no longer drop the
    count: an edge-tie image among ordinary ones comes out equal to the oracle."""
    import image_analysis_amd.detection as det
    from image_analysis_amd import frames as F
    from image_analysis_amd.synth import synth_rgb
    tie = _edge_tie_image(1)
    H, W = tie.shape[:2]
    plain = synth_rgb(H, W, 4, 5)
    mm_r, mm_c = gui_params(tie)
    monkeypatch.setattr(det, "min_maxc", mm_c)
    monkeypatch.setattr(det, "min_maxr", mm_r)
    imgs = np.stack([plain, tie, plain, tie, tie])
    want = [O.detect_debug(im, mm_r, mm_c)["strips"] for im in imgs]
    out = det.analyze_batch(imgs, max_batch=4)
    got = list(F.analyze_stream(iter(imgs), batch=2, min_maxr=mm_r, min_maxc=mm_c))
    for i, im in enumerate(imgs):
        assert [c.shape[:2] for c, _p, _k in out[i]] == [(min(b, H) - a, min(e, W) - c) for a, b, c, e in want[i]]
        assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in got[i][1]] == want[i]
        for k, (a, b, c, e) in enumerate(want[i]):
            wp, _wk = O.image_analysis(im[a:b, c:e], False)
            assert np.array_equal(out[i][k][1], wp) and np.array_equal(got[i][2][k], wp)
    crops = det.detect_lateral_flow_tests(tie)
    assert [c.shape[:2] for c in crops] == [(min(b, H) - a, min(e, W) - c) for a, b, c, e in want[1]]


def test_equal_channel_images_take_one_pass_and_match_the_oracle(capi, golden_dir):
    """cv2.imread hands the reference three equal channels for every grey file (detection.py:454; all of its sample JPEGs):
    gray is then a function of the byte value, so ONE pass (k_grey_hist: byte histogram + a check of every pixel's channels)
    replaces the min/max and the histogram pass.  Forced on for a mixed batch: a grey photo, a colour image (falls back to
    the ordinary passes), a grey image with one unequal pixel at its very end (falls back too), a constant grey image, a
    grey image with 2 values; every intermediate against the oracle, and LFA_DBG_GREY tells which way each image went."""
    from image_analysis_amd.synth import synth_rgb
    g = np.load(os.path.join(golden_dir, "image1_decoded.npz"))["gray"][:256, :320]
    H, W = g.shape
    grey = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    colour = synth_rgb(H, W, 5, 21)
    almost = grey.copy(); almost[-1, -1, 2] ^= 1
    const = np.full((H, W, 3), 77, np.uint8)
    two = np.where(g[:, :, None] > 100, np.uint8(200), np.uint8(30)).repeat(3, axis=2)
    imgs = [grey, colour, almost, const, np.ascontiguousarray(two)]
    mm_r, mm_c = gui_params(grey)
    e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=len(imgs))
    try:
        # (1: float64 compare path; 4: class plane forced for the images that fall back; 4096: no byte plane for the mask pass;
        #  line_length 17: the byte plane's home, the bbox buffer, is then smaller than the image and the plane is left out)
        for flags, L in ((1024, 10), (1024 | 1, 10), (1024 | 4, 10), (1024 | 4096, 10), (1024, 17), (1024, 3)):
            e.set_debug_flags(flags)
            for i, im in enumerate(imgs):                     # one by one: the debug fetches look at the last call
                check_detect(capi, e, im, mm_r, mm_c, L)
                assert e.debug_grey_path(0) == (i in (0, 3, 4)), (flags, i)
            if L != 10:
                continue
            for i, im in enumerate(imgs):
                check_detect(capi, e, im, mm_r, mm_c)
                assert e.debug_grey_path(0) == (i in (0, 3, 4)), (flags, i)
            res = e.detect(np.stack(imgs), mm_r, mm_c)
            assert [e.debug_grey_path(i) for i in range(len(imgs))] == [True, False, False, True, True]
            for i, im in enumerate(imgs):
                d = O.detect_debug(im, mm_r, mm_c)
                got = [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in res[0][i]]
                assert got == d["strips"] and capi.host_threshold(res[1][i], True) == d["threshold"], (flags, i)
                assert int(res[1][i]["n_labels"]) == d["n_labels"]
        e.set_debug_flags(2048)
        check_detect(capi, e, grey, mm_r, mm_c)
        assert not e.debug_grey_path(0)
    finally:
        e.close()


def test_a_handle_finds_grey_batches_by_itself(capi, golden_dir):
    """The mode follows the handle's last completed call: the min/max pass notices images whose channels are equal throughout,
    the count travels with the handle's other counters, and the next call takes the one-pass path while at least half of
    the colour images were grey; a colour batch switches it off again.  Results equal the oracle at every step."""
    from image_analysis_amd.synth import synth_rgb
    g = np.load(os.path.join(golden_dir, "image1_decoded.npz"))["gray"][:192, :256]
    H, W = g.shape
    greys = np.stack([np.repeat(np.roll(g, 7 * k, axis=1)[:, :, None], 3, axis=2) for k in range(3)])
    colours = np.stack([synth_rgb(H, W, 4, 30 + k) for k in range(3)])
    mm_r, mm_c = gui_params(g)
    e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=3)
    try:
        def run(batch):
            strips, info = e.detect(batch, mm_r, mm_c)
            path = [e.debug_grey_path(i) for i in range(len(batch))]
            for i, im in enumerate(batch):
                d = O.detect_debug(im, mm_r, mm_c)
                assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips[i]] == d["strips"]
                assert capi.host_threshold(info[i], True) == d["threshold"] and int(info[i]["n_labels"]) == d["n_labels"]
            return path
        assert run(greys) == [False] * 3              # first call: the ordinary passes, which count the grey images
        assert run(greys) == [True] * 3               # now in grey mode
        mixed = np.stack([greys[0], colours[0], greys[1]])
        assert run(mixed) == [True, False, True]      # still on (3 of 3 grey last time); the colour image falls back
        assert run(colours) == [False] * 3            # on (2 of 3), every image falls back
        assert run(colours) == [False] * 3            # off (0 of 3, counted on every pixel while the mode was on: out for this and three more calls)
        for _ in range(3):
            assert run(greys) == [False] * 3
        assert run(greys) == [True] * 3
        # grey where the sampled look of the min/max pass falls, coloured elsewhere is possible: the mode then costs a wasted pass
        # once, finds the exact count at odds with the hint, and backs off -- results are right throughout (run() checks them)
        speck = greys.copy()
        speck[:, 100, 37, 1] ^= 0x10
        seen = [run(speck) for _ in range(7)]
        assert all(p == [False] * 3 for p in seen)
    finally:
        e.close()


def test_predicted_threshold_mode_matches_the_oracle(capi):
    """A handle that has completed a colour call expects the next call's integer mask thresholds inside the range the last
    call's had: the histogram pass then writes one bit per pixel (at or above that window) instead of the 4-bit class, lists
    the pixel groups inside the window, and the mask comes from the bits plus an exact compare of the listed pixels
    (k_mask_fix).  Checked here: the mode follows the calls made on the handle; every intermediate equals the oracle when the
    prediction holds, when it fails (another scene: compare on the image), with a window without margin (thresholds ON its
    ends), with every pixel inside the window (forced: each one goes through the list), and when that list overflows."""
    from image_analysis_amd.synth import synth_rgb
    H, W = 384, 512
    scene = np.stack([synth_rgb(H, W, 5, 70 + k) for k in range(4)])
    dim = np.ascontiguousarray((scene.astype(np.float64) * 0.45).astype(np.uint8))               # thresholds far below the first scene's
    soft = scene.copy()                                                                          # blurred strip edges: pixels inside the window
    for k in range(4):
        a = soft[k].astype(np.float64)
        soft[k] = ((a + np.roll(a, 1, axis=1) + np.roll(a, 2, axis=1) + np.roll(a, 3, axis=1)) / 4).astype(np.uint8)
    mm_r, mm_c = gui_params(scene[0])
    e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=4)
    try:
        def run(batch, L=10):
            res = e.detect(batch, mm_r, mm_c, L)
            for i in range(len(batch)):
                check_detect(capi, e, batch[i], mm_r, mm_c, L, batch_index=i, res=res)
            return [e.debug_pred_path(i) for i in range(len(batch))]
        assert run(scene) == [False] * 4                 # nothing to predict from
        assert run(scene) == [True] * 4
        e.close()
        e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=4)
        assert run(soft) == [False] * 4
        assert run(soft) == [True] * 4                   # blurred edges: pixels inside the window, decided through the list
        assert run(scene) == [False] * 4                 # (sharp edges, other extremes: other thresholds)
        e.close()
        e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=4)
        run(scene)
        assert run(scene[:1]) == [True]
        assert run(dim) == [False] * 4                   # window of the bright scene: every image falls back to the compare on the image
        for _ in range(8):
            assert run(dim) == [False] * 4               # ... and the handle stays out of the mode for eight calls
        assert run(dim) == [True] * 4
        mixed = np.stack([dim[0], scene[0], dim[1], dim[2]])
        assert run(mixed) == [True, False, True, True]   # per image
        e.close()
        e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=4)
        e.set_debug_flags(16384)                         # no margin: window = [smallest, largest] threshold of the call before
        run(scene)
        assert run(scene) == [True] * 4                  # two of the thresholds are the window's ends
        e.set_debug_flags(32768)                         # every pixel inside the window
        assert run(soft) == [True] * 4
        assert run(soft, L=100) == [False] * 4           # the list lives in a buffer sized by line_length: it overflows, compare on the image
        e.set_debug_flags(32768 | 1)                     # float64 compare path: no integer threshold, no bits
        assert run(scene) == [False] * 4
        e.set_debug_flags(8192)
        assert run(scene) == [False] * 4
        e.set_debug_flags(0)
    finally:
        e.close()


def test_kernel_timing_filter_and_launch_stride(capi, eng):
    """lfa_timing_enable: "*" brackets every launch, "<kernel>" that kernel only, "<kernel>/N" every N-th launch of it (what
    bench.py uses for the dominant kernel, so that the event records between a lane's kernels do not stretch the step); a
    stride below 1 is refused."""
    from image_analysis_amd.synth import synth_rgb
    img = synth_rgb(96, 128, 2, 5)
    mm_r, mm_c = gui_params(img)
    def calls(k):
        for _ in range(k):
            eng.detect(img[None], mm_r, mm_c, 10)
        return eng.timing_read()
    try:
        eng.timing("*")
        everything = calls(2)
        assert everything["k_hist_rgb"][1] == 2 and everything["k_minmax_rgb"][1] == 2 and len(everything) > 5
        eng.timing("k_hist_rgb")
        only = calls(3)
        assert list(only) == ["k_hist_rgb"] and only["k_hist_rgb"][1] == 3 and only["k_hist_rgb"][0] > 0.0
        eng.timing("k_hist_rgb/4")
        some = calls(9)
        assert list(some) == ["k_hist_rgb"] and some["k_hist_rgb"][1] == 3          # launches 0, 4 and 8
        assert capi.lib.lfa_timing_enable(eng._h, b"k_hist_rgb/0") == -1              # LFA_ERR_ARG
    finally:
        eng.timing(None)


@pytest.mark.parametrize("shape", [(70, 64), (129, 128), (300, 576), (64, 192), (1, 64)])
def test_predicted_threshold_bits_turned_in_the_run_extraction_or_by_k_mask(capi, shape):
    """Widths that are multiples of 64: an image decided from the predicted-threshold bit plane never gets a transposed mask --
    k_runs_count loads 64 x 64 bit blocks of the plane and turns them in the wave (blocks that are all clear or all set are not
    turned), k_mask_fix sets the listed pixels in the plane.  Debug flag 65536 sends the same call through k_mask instead.  Both
    ways equal the oracle in every intermediate: heights off the 64-row grid, noisy masks (no flat block, short runs that the
    opening drops), flat images with a few strips, and every pixel through the list (flag 32768)."""
    from image_analysis_amd.synth import synth_rgb
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    strips = np.stack([synth_rgb(max(H, 16), W, 3, 500 + k)[:H] for k in range(3)])
    noisy = strips.copy()
    speck = rng.random((3, H, W)) < 0.3
    noisy[speck] = 255 - noisy[speck]                                  # salt and pepper on both sides of any threshold
    mm_r, mm_c = gui_params(strips[0])
    for batch in (strips, noisy):
        for flags in (0, 65536, 32768, 32768 | 65536):
            e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=3)
            try:
                e.set_debug_flags(flags & 65536)
                e.detect(batch, mm_r, mm_c, 10)                        # the call the window comes from
                e.set_debug_flags(flags)
                for L in (10, 3):
                    res = e.detect(batch, mm_r, mm_c, L)
                    took = [e.debug_pred_path(i) for i in range(3)]
                    for i in range(3):
                        check_detect(capi, e, batch[i], mm_r, mm_c, L, batch_index=i, res=res)
                    assert all(took) or (flags & 32768) or H == 1, (flags, took)   # (forced window: the list may overflow -> compare on the image)
            finally:
                e.close()


def test_predicted_threshold_mode_in_analyze_batch_and_across_shapes(capi):
    """lfa_analyze_batch on one handle, call after call, with shapes and scenes changing in between: strips, profiles and peaks
    equal the oracle whichever way each call's masks were decided."""
    from image_analysis_amd.synth import synth_rgb
    e = capi.Engine(device=0, max_h=384, max_w=512, max_batch=3)
    try:
        took = []
        for step, (H, W, seed, scale) in enumerate([(384, 512, 1, 1.0), (384, 512, 1, 1.0), (192, 256, 3, 1.0), (192, 256, 4, 0.6),
                                                   (192, 256, 4, 0.6), (384, 496, 6, 1.0), (384, 496, 7, 1.0)]):
            batch = np.stack([np.ascontiguousarray((synth_rgb(H, W, 4, 100 * seed + k) * scale).astype(np.uint8)) for k in range(3)])
            mm_r, mm_c = gui_params(batch[0])
            res, info = e.analyze(batch, mm_r, mm_c)
            took.append([e.debug_pred_path(i) for i in range(3)])
            for i, (strips, profiles, peaks) in enumerate(res):
                d = O.detect_debug(batch[i], mm_r, mm_c)
                assert capi.host_threshold(info[i], True) == d["threshold"] and info[i]["n_labels"] == d["n_labels"]
                assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips] == d["strips"], (step, i)
                for k, (a, b, c, w) in enumerate(d["strips"]):
                    wp, wk = O.image_analysis(batch[i][a:b, c:w], False)
                    assert np.array_equal(profiles[k], wp)
                    assert [(int(peaks[k]["start"][q]), int(peaks[k]["end"][q]), float(peaks[k]["area"][q]))
                            for q in range(int(peaks[k]["n_peaks"]))] == [(s0, e0, float(a0)) for s0, e0, a0 in wk]
        assert all(took[1]) and not any(took[0])
    finally:
        e.close()


def test_predicted_threshold_mode_on_uint16_frames(capi):
    """uint16 frames on the one-pass histogram take the predicted-threshold mode too (pixel value in place of G, T = x_thr + 1): one bit
    per pixel against the window of the last call's thresholds instead of the 4-bit class plane, undecided groups listed.  The handle
    tries only while the window would have held at least half of the last call's frames; every intermediate equals the oracle whichever way."""
    rng = np.random.default_rng(31)
    eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=3)
    try:
        sep = np.stack([_u16_separated(rng) for _ in range(3)])
        soft = sep.copy()                                            # blurred strip edges: pixels inside the window (few enough
        for k in range(3):                                           # near the threshold for the handle to try: 1/1024 of 40 960)
            a = soft[k].astype(np.float64)
            soft[k] = ((a + np.roll(a, 1, axis=1)) / 2).astype(np.uint16)
        noisy = np.stack([_u16_separated(rng, noise=True) for _ in range(3)])
        bright = np.ascontiguousarray(np.minimum(sep.astype(np.int64) * 3, 16000).astype(np.uint16))     # thresholds elsewhere

        def run(batch, L=10):
            res = eng.detect(batch, 60, 10, L)
            for i in range(len(batch)):
                check_detect(capi, eng, batch[i], 60, 10, L, batch_index=i, res=res)
            return [eng.debug_pred_path(i) for i in range(len(batch))]
        assert run(sep) == [False] * 3                   # two-pass histogram
        assert run(sep) == [False] * 3                   # one pass: this call reports its thresholds
        assert run(sep) == [False] * 3                   # this one how many frames the window made of them would have held
        assert run(sep) == [True] * 3
        eng.close()
        eng = capi.Engine(device=0, max_h=160, max_w=256, max_batch=3)
        run(soft); run(soft); run(soft)
        assert run(soft) == [True] * 3                   # pixels inside the window, decided through the list
        assert run(bright) == [False] * 3                # another scene: every frame is compared on the frame itself
        for _ in range(8):
            assert run(bright) == [False] * 3            # ... and the handle stays out of the mode for eight calls
        assert run(bright) == [True] * 3
        mixed = np.stack([bright[0], sep[0], bright[1]])
        assert run(mixed) == [True, False, True]         # per image
        run(noisy); run(noisy)
        assert run(noisy) == [False] * 3                 # thresholds inside the noise: not tried
        eng.set_debug_flags(16 | 32768)                  # every pixel inside the window: each one goes through the list
        assert run(sep) == [True] * 3
        eng.set_debug_flags(8192)
        assert run(sep) == [False] * 3
        eng.set_debug_flags(0)
    finally:
        eng.close()
    eng = capi.Engine(device=0, max_h=400, max_w=256, max_batch=2)
    try:
        tall = np.stack([_u16_separated(rng, H=400) for _ in range(2)])
        eng.set_debug_flags(16 | 32768)
        assert run(tall) == [True] * 2
        assert run(tall, L=300) == [False] * 2           # the list lives in a buffer sized by line_length: it overflows, compare on the frame
        eng.set_debug_flags(0)
    finally:
        eng.close()


def test_predicted_threshold_mode_on_equal_channel_images(capi, golden_dir):
    """Grey mode + predicted-threshold mode: k_grey_hist writes the bits itself, in grey levels (G = 10000 * b) against the last call's
    range of thresholds WITHOUT margin or bucket rounding -- a series with one byte threshold has no undecided pixel, any other
    series falls back per image.  Checked: identical thresholds (rolled copies: bits hold, even when a multiple of 10000 lies
    beside the threshold), a brighter copy (other threshold: compare on the image), a colour image and an almost-grey image inside
    the call (k_hist_rgb writes their bits and lists), every pixel through the list (forced), the float64 compare path."""
    from image_analysis_amd.synth import synth_rgb
    g = np.load(os.path.join(golden_dir, "image1_decoded.npz"))["gray"][:192, :256]
    H, W = g.shape
    greys = np.stack([np.repeat(np.roll(g, 7 * k, axis=1)[:, :, None], 3, axis=2) for k in range(3)])
    brighter = np.ascontiguousarray(np.minimum(greys.astype(np.int32) + 40, 255).astype(np.uint8))
    colour = synth_rgb(H, W, 4, 30)
    almost = greys[1].copy(); almost[-1, -1, 2] ^= 1
    mm_r, mm_c = gui_params(g)
    e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=3)
    try:
        def run(batch):
            res = e.detect(batch, mm_r, mm_c)
            for i in range(len(batch)):
                check_detect(capi, e, batch[i], mm_r, mm_c, batch_index=i, res=res)
            return [e.debug_grey_path(i) for i in range(len(batch))], [e.debug_pred_path(i) for i in range(len(batch))]
        assert run(greys) == ([False] * 3, [False] * 3)          # ordinary passes; they count the grey images and report the thresholds
        assert run(greys) == ([True] * 3, [True] * 3)            # grey mode, and the bits hold: one threshold, no undecided pixel
        assert run(greys) == ([True] * 3, [True] * 3)
        mixed = np.stack([greys[0], colour, almost])
        grey_path, pred_path = run(mixed)
        assert grey_path == [True, False, False] and pred_path[0]  # the colour and the almost-grey image: k_hist_rgb's bits, or the image
        assert run(brighter)[1] == [False] * 3                   # other thresholds: every image compared on the image itself
        e.set_debug_flags(1024 | 32768)                          # grey mode and every pixel through the list
        assert run(greys) == ([True] * 3, [True] * 3)
        assert run(mixed)[1] == [True] * 3
        e.set_debug_flags(1024 | 32768 | 1)                      # float64 compare path: no integer threshold, no bits
        assert run(greys)[1] == [False] * 3
        e.set_debug_flags(1024 | 8192)
        assert run(greys) == ([True] * 3, [False] * 3)
        e.set_debug_flags(0)
    finally:
        e.close()
