"""detect_lines (detection.py:156-201) on the GPU, through the C ABI (lfa_detect_lines_batch).

* against tests/golden/ref_detect_lines.npz -- the reference's own lines run over its vendored scikit-image sources up to
  label() -- with no oracle in between: percentile bits and the float image handed to label(), rebuilt here from the
  device's level image and its per-level value table;
* against oracle/lines_oracle.py on larger and odd-shaped images, every stage bit for bit: Gaussian, img_as_uint image,
  CLAHE levels, percentiles, label image (from the device's component roots), label count and the returned boxes;
* through the drop-in module, including its exceptions."""
import os

import numpy as np
import pytest

from golden_util import GOLDEN

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(GOLDEN, "ref_detect_lines.npz"))
NAMES = [str(n) for n in G["names"]]


@pytest.fixture(scope="module")
def capi():
    import image_analysis_amd._capi as capi
    return capi


@pytest.fixture(scope="module")
def eng(capi):
    e = capi.Engine(device=0, max_h=1536, max_w=2048, max_c=1, max_batch=4)
    e.set_debug_flags(64)                                           # keep the Gaussian image for lines_debug(LDBG_GAUSS)
    yield e
    e.close()


def labels_from_roots(roots):
    """label image (raster order of first pixel) from the per-pixel root indices the device keeps (-1 = background)"""
    r = roots.ravel()
    fg = r >= 0
    ids = np.unique(r[fg])
    assert np.array_equal(r[ids], ids)                      # every root is its own root and the smallest index of its component
    out = np.zeros(r.size, np.int64)
    out[fg] = np.searchsorted(ids, r[fg]) + 1
    return out.reshape(roots.shape)


@pytest.mark.parametrize("name", NAMES)
def test_hip_equals_the_reference_run(capi, eng, name):
    im = G[f"{name}__image"]
    H, W = im.shape
    boxes, info = eng.detect_lines(im[None])
    inf = info[0]
    assert np.array_equal(np.array([inf["p80"], inf["p99"]]).view(np.uint64), G[f"{name}__percentiles"].view(np.uint64))
    levels = eng.lines_debug(0, capi.LDBG_LEVELS, H, W)
    lval = eng.lines_debug(0, capi.LDBG_LVAL, H, W)
    assert np.array_equal(lval[levels].view(np.uint64), G[f"{name}__label_input"].view(np.uint64))


def _against_oracle(capi, eng, im):
    from oracle import lines_oracle as L
    H, W = im.shape
    d = L.lines_debug(im)
    boxes, info = eng.detect_lines(im[None])
    inf = info[0]
    assert np.array_equal(eng.lines_debug(0, capi.LDBG_GAUSS, H, W).view(np.uint64), d["gauss"].view(np.uint64))
    assert np.array_equal(eng.lines_debug(0, capi.LDBG_U16, H, W), d["u16"])
    assert (int(inf["umin"]), int(inf["umax"])) == (int(d["u16"].min()), int(d["u16"].max()))
    assert np.array_equal(eng.lines_debug(0, capi.LDBG_LEVELS, H, W), d["levels"])
    assert (int(inf["cmin"]), int(inf["cmax"])) == (int(d["levels"].min()), int(d["levels"].max()))
    assert np.float64(inf["p80"]).tobytes() == np.float64(d["p80"]).tobytes()
    assert np.float64(inf["p99"]).tobytes() == np.float64(d["p99"]).tobytes()
    lab = labels_from_roots(eng.lines_debug(0, capi.LDBG_ROOTS, H, W))
    assert np.array_equal(lab, d["labels"])
    assert int(inf["n_labels"]) == int(d["labels"].max())
    got = [[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in boxes[0]]
    assert got == d["strips"]
    assert int(inf["n_boxes"]) == len(d["strips"]) and int(inf["status"]) == 0
    return d


@pytest.mark.parametrize("name", NAMES)
def test_hip_equals_oracle_on_the_fixture_images(capi, eng, name):
    _against_oracle(capi, eng, G[f"{name}__image"])


@pytest.mark.parametrize("shape,dtype,seed", [((384, 512), np.uint8, 4), ((768, 1024), np.uint16, 5), ((301, 517), np.uint8, 6),
                                              ((1536, 2048), np.uint8, 7), ((8, 8), np.uint8, 8), ((2, 3), np.uint8, 9)])
def test_hip_equals_oracle_on_larger_and_odd_frames(capi, eng, shape, dtype, seed):
    from image_analysis_amd.synth import synth_lines
    if min(shape) >= 64:
        im = synth_lines(shape[0], shape[1], seed, dtype)
    else:
        im = np.random.default_rng(seed).integers(0, 256, size=shape).astype(dtype)
    d = _against_oracle(capi, eng, im)
    if shape == (384, 512):
        assert d["strips"] == [[100, 107, 0, 512], [293, 300, 0, 512]]


@pytest.mark.parametrize("key", ["image2", "image4", "image7", "wick_crop", "image5_crop", "austin_u16_crop", "strip_channel0", "excel2_frame"])
def test_hip_equals_oracle_on_the_reference_sample_images(capi, eng, key):
    """the reference's own sample inputs (decoded in the build container, tests/golden/real_samples.npz), its 12-bit frame, and
    one channel of its strip crop -- the kind of image detect_lines was written for"""
    if key == "excel2_frame":
        from PIL import Image
        im = np.array(Image.open(os.path.join(GOLDEN, "excel2_frame.png")))[:1024, :1536]
    else:
        rs = np.load(os.path.join(GOLDEN, "real_samples.npz"))
        im = np.ascontiguousarray(rs["test_strip_buggy_bgr"][:, :, 0]) if key == "strip_channel0" else rs[key]
    _against_oracle(capi, eng, im)


def test_batch_of_different_images_and_box_overflow(capi, eng):
    from oracle import lines_oracle as L
    ims = np.stack([G["tiny_seed2__image"], G["tiny_seed7__image"], G["tiny_seed23__image"]])
    want = [L.detect_lines(im) for im in ims]
    boxes, info = eng.detect_lines(ims)
    got = [[[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in bx] for bx in boxes]
    assert got == want
    boxes, info = eng.detect_lines(ims, max_boxes=2)              # more boxes than slots: the binding asks again with more
    got = [[[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in bx] for bx in boxes]
    assert got == want


def test_box_table_is_clean_between_calls_of_different_shapes(capi, eng):
    """the box extents table is never initialised per call (k_ccl_collect puts back what it reads): a large frame, a small
    one and a batch whose images sit at other offsets in the same table, twice over, each against the oracle"""
    from oracle import lines_oracle as L
    from image_analysis_amd.synth import synth_lines
    big, small = synth_lines(768, 1024, 31, np.uint8), synth_lines(200, 264, 32, np.uint8)
    batch = np.stack([synth_lines(384, 512, 33 + k, np.uint8) for k in range(3)])
    want_batch = [L.detect_lines(im) for im in batch]
    assert all(len(w) > 0 for w in want_batch) and len(L.detect_lines(big)) > 0
    for _ in range(2):
        _against_oracle(capi, eng, big)
        _against_oracle(capi, eng, small)
        boxes, info = eng.detect_lines(batch)
        assert [[[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in bx] for bx in boxes] == want_batch
        boxes, info = eng.detect_lines(batch, max_boxes=1)          # the overflow path leaves the table clean too
        assert [[[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in bx] for bx in boxes] == want_batch


def test_boxes_of_components_that_spread_left_and_below_their_first_pixel(capi, eng):
    """a component's box is kept as extents from its raster-first pixel: bright shapes whose first pixel is neither their
    leftmost nor their only top-row pixel (a wide bar hanging from a thin stem on its right, a staircase to the left)"""
    H, W = 256, 320
    rng = np.random.default_rng(5)
    im = rng.integers(0, 40, size=(H, W)).astype(np.uint8)
    im[56:64, 286:294] = 255; im[64:68, 150:294] = 255                 # stem at the right end, bar to the left below it
    for k in range(6):
        im[150 + 3 * k:153 + 3 * k, 300 - 24 * k - 30:300 - 24 * k] = 255   # staircase: every step further left
    d = _against_oracle(capi, eng, im)
    # (label() sees the image cast to intp: only the pixels at 1.0 are foreground, so there are few components; what this
    #  case is about is that some component reaches left of its raster-first pixel)
    lab = d["labels"]
    assert lab.max() >= 1
    spreads = False
    for k in range(1, int(lab.max()) + 1):
        rr, cc = np.nonzero(lab == k)
        spreads |= bool(cc.min() < cc[0])
    assert spreads


def test_without_the_debug_flag_the_gaussian_image_is_not_kept_and_the_results_are_the_same(capi, eng):
    from image_analysis_amd.synth import synth_lines
    im = synth_lines(384, 512, 4, np.uint16)
    boxes, info = eng.detect_lines(im[None])
    want_levels = eng.lines_debug(0, capi.LDBG_LEVELS, 384, 512)
    plain = capi.Engine(device=0, max_h=384, max_w=512, max_c=1, max_batch=1)
    boxes2, info2 = plain.detect_lines(im[None])
    assert np.array_equal(np.asarray(boxes[0]), np.asarray(boxes2[0])) and info[0].tobytes() == info2[0].tobytes()
    assert np.array_equal(plain.lines_debug(0, capi.LDBG_LEVELS, 384, 512), want_levels)
    with pytest.raises(RuntimeError, match="debug flag 64"):
        plain.lines_debug(0, capi.LDBG_GAUSS, 384, 512)
    plain.close()


def test_class_lookup_and_class_function_agree(capi, eng):
    """k_ccl_init computes a pixel's class from its level where the class table has the usual shape; debug flag 128 makes it
    look the class up instead: same labels, same boxes (odd pixel count: the scalar path of k_ccl_collect too)"""
    from image_analysis_amd.synth import synth_lines
    for im in (synth_lines(384, 512, 11, np.uint8), synth_lines(301, 517, 12, np.uint8)):
        H, W = im.shape
        boxes, info = eng.detect_lines(im[None])
        roots = eng.lines_debug(0, capi.LDBG_ROOTS, H, W)
        eng.set_debug_flags(64 | 128)
        try:
            boxes2, info2 = eng.detect_lines(im[None])
            roots2 = eng.lines_debug(0, capi.LDBG_ROOTS, H, W)
        finally:
            eng.set_debug_flags(64)
        assert len(boxes[0]) > 0 and np.array_equal(np.asarray(boxes[0]), np.asarray(boxes2[0]))
        assert info[0].tobytes() == info2[0].tobytes() and np.array_equal(roots, roots2)


def test_drop_in_detect_lines(capi):
    import image_analysis_amd.detection as detection
    from oracle import lines_oracle as L
    im = G["lines_200x264_u8__image"]
    assert detection.detect_lines(im) == L.detect_lines(im) == [[51, 55, 0, 264]]
    with pytest.raises(ValueError):
        detection.detect_lines(np.zeros((8, 8, 3), np.uint8))
    # a saturated frame: the Gaussian of all-ones exceeds 1 by rounding and img_as_uint refuses it, here as in the reference
    sat = np.full((32, 32), 255, np.uint8)
    try:
        want = L.detect_lines(sat)
    except ValueError as e:
        with pytest.raises(ValueError, match=str(e)[:20]):
            detection.detect_lines(sat)
    else:
        assert detection.detect_lines(sat) == want
