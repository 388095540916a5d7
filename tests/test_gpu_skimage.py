"""HIP path == the reference's own detect executed over its vendored scikit-image sources (tests/golden/ref_skimage_*.npz,
see tests/test_skimage_vectors.py for what the fixtures are).  No oracle in between: threshold bits, mask / opened mask /
label image hashes, bbox + area table and the returned boxes come straight from the fixture.  Through the C ABI."""
import os

import numpy as np
import pytest
from PIL import Image

from golden_util import GOLDEN, clipped, sha, skimage_detect_cases

pytestmark = pytest.mark.gpu

G, NAMES = skimage_detect_cases()


@pytest.fixture(scope="module")
def capi():
    import image_analysis_amd._capi as capi
    return capi


def _check(capi, eng, im, mm_r, mm_c, L, thr_want, mask_sha, opened_sha, labels_sha, n_labels, label00, table, boxes):
    H, W = im.shape[:2]
    strips, info = eng.detect(im[None], mm_r, mm_c, L)
    inf = info[0]
    assert (inf["status"] & 3) == 0 and inf["ambiguous"] == 0
    thr = capi.host_threshold(inf, im.ndim == 3 or im.dtype == np.float64)       # (float images take the colour path's float64 arithmetic)
    assert np.asarray(thr, dtype=thr_want.dtype).tobytes() == thr_want.tobytes(), (thr, thr_want)
    assert np.array_equal(sha(eng.debug(0, capi.DBG_MASK, H, W)), mask_sha)
    if opened_sha is not None:
        assert np.array_equal(sha(eng.debug(0, capi.DBG_OPENED, H, W)), opened_sha)
    assert np.array_equal(sha(eng.debug(0, capi.DBG_LABELS, H, W)), labels_sha)
    assert int(inf["n_labels"]) == n_labels
    if label00 is not None:
        assert int(inf["label00"]) == label00
    if table is not None:
        assert np.array_equal(eng.debug_table(0, n_labels).astype(np.int64), table)
    got = [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips[0]]
    assert clipped(got, H, W) == boxes


@pytest.mark.parametrize("name", NAMES)
def test_hip_detect_equals_the_reference_run_over_vendored_skimage(capi, name):
    im = G[f"{name}__image"]
    mm_r, mm_c, L = (int(v) for v in G[f"{name}__params"])
    eng = capi.Engine(device=0, max_h=im.shape[0], max_w=im.shape[1], max_batch=1)
    try:
        # default path choice, class plane forced (colour / uint16), class plane off, float64 compare path (colour)
        # colour uint8: the equal-channel mode forced on as well (grey fixtures take its one-pass path, the others fall back to
        # the ordinary passes inside it), alone and with the float64 compare path
        for flags in ((0, 4, 2, 1, 1024, 1024 | 1, 2048) if im.ndim == 3 else (0, 0, 4, 2)):
            eng.set_debug_flags(flags)
            _check(capi, eng, im, mm_r, mm_c, L, G[f"{name}__threshold"], G[f"{name}__mask_sha"], G[f"{name}__opened_sha"],
                   G[f"{name}__labels_sha"], int(G[f"{name}__n_labels"]), int(G[f"{name}__label00"]), G[f"{name}__table"],
                   G[f"{name}__boxes"].tolist())
    finally:
        eng.close()


def test_hip_excel2_frame_equals_the_reference_run(capi):
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    frame = np.array(Image.open(os.path.join(GOLDEN, "excel2_frame.png")))
    eng = capi.Engine(device=0, max_h=2048, max_w=2048, max_batch=1)
    try:
        _check(capi, eng, frame, 1020, 51, 10, v["excel2_threshold"], v["excel2_mask_sha"], None, v["excel2_labels_sha"],
               int(v["excel2_n_labels"]), None, None, v["excel2_boxes"].tolist())
    finally:
        eng.close()


def test_hip_point_operation_tables_equal_the_vendored_skimage(capi):
    """_capi.point_op_lut (the table lfa_profile_lut_batch applies per pixel) against adjust_gamma -> rescale_intensity ->
    invert of the vendored skimage on every value of uint8 / uint16; then one profile through the kernel with each table."""
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    eng = capi.Engine(device=0, max_h=256, max_w=256, max_c=1, max_batch=1)
    try:
        for k in range(int(v["n_pointops"])):
            gamma_level, lo, hi, inv = v[f"pointop{k}_params"]
            dt = np.uint8 if int(v[f"pointop{k}_dtype"]) == 1 else np.uint16
            lut = capi.point_op_lut(dt, gamma_level, None if lo < 0 else lo, None if hi < 0 else hi, bool(inv))
            want = v[f"pointop{k}_table"]
            assert lut.dtype == dt and np.array_equal(lut, want), k
            allv = np.arange(np.iinfo(dt).max + 1, dtype=dt).reshape(-1, 256)[:256]
            roi = np.zeros(1, dtype=capi.ROI_DT)
            roi["r1"], roi["c1"] = allv.shape
            prof, _pk = eng.profiles(allv[None], roi, luts=lut)
            assert np.array_equal(prof[0], want.reshape(-1, 256)[:256].mean(axis=1)), k
    finally:
        eng.close()


def test_hip_gaussian_equals_the_vendored_skimage(capi):
    v = np.load(os.path.join(GOLDEN, "ref_skimage_vectors.npz"))
    eng = capi.Engine(device=0, max_h=64, max_w=64, max_batch=1)
    try:
        for g in range(int(v["n_gauss"])):
            im = v[f"gauss{g}_image"]
            out = eng.gaussian(im[None], sigma=1.0)[0]
            assert out.dtype == np.float64 and np.array_equal(out, v[f"gauss{g}_out"]), g      # bit-exact (tolerance 0)
    finally:
        eng.close()
