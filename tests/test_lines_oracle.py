"""oracle/lines_oracle.py (detect_lines, detection.py:156-201) against what the reference holds.

* tests/golden/ref_detect_lines.npz: the reference's own lines :156-201 executed over its vendored scikit-image sources up to
  the label() call (tests/golden/make_golden.py --only-lines): SHA-256 of every intermediate, the percentiles, the image handed
  to label().
* label() of a non-boolean image is Cython in skimage and cannot run here: its rule is pinned by the known-answer arrays of the
  vendored test-suite (skimage/measure/tests/test_ccomp.py:14-24, :58-62, :76-99, and the docstring example _label.py:108-115),
  transcribed below as data.
CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lines_oracle as L  # noqa: E402
from tests.golden_util import GOLDEN, sha  # noqa: E402

G = np.load(os.path.join(GOLDEN, "ref_detect_lines.npz"))
NAMES = [str(n) for n in G["names"]]


@pytest.mark.parametrize("name", NAMES)
def test_every_stage_equals_the_reference_run(name):
    d = L.lines_debug(G[f"{name}__image"])
    for key in ("gauss", "local", "gamma", "enhanced"):
        assert np.array_equal(sha(d[key]), G[f"{name}__{key}_sha"]), key
    assert np.array_equal(np.array([d["p80"], d["p99"]]).view(np.uint64), G[f"{name}__percentiles"].view(np.uint64))
    assert np.array_equal(d["rescaled"].view(np.uint64), G[f"{name}__label_input"].view(np.uint64))


def test_some_fixture_yields_boxes_and_some_none():
    got = {name: L.detect_lines(G[f"{name}__image"]) for name in NAMES}
    assert got["lines_200x264_u8"] == [[51, 55, 0, 264]]
    assert got["constant_40x56_u8"] == []
    assert got["lines_384x512_u8"] == [[100, 107, 0, 512], [293, 300, 0, 512]]
    # (before round 3 the oracle took every distinct float level as a class of its own and found 4 and 3 boxes here; label()
    #  works on the image cast to intp, so only the pixels equal to 1.0 are foreground -- see test_label_truncates_floats)
    assert got["tiny_seed2"] == [[15, 17, 0, 23]]
    assert got["tiny_seed7"] == [[3, 4, 0, 23], [7, 9, 0, 23]]
    assert got["tiny_seed23"] == [[7, 9, 0, 23]]


# skimage/measure/tests/test_ccomp.py:14-24 (the image and its expected labels)
X = np.array([[0, 0, 3, 2, 1, 9],
              [0, 1, 1, 9, 2, 9],
              [0, 0, 1, 9, 9, 9],
              [3, 1, 1, 5, 3, 0]])
LABELS = np.array([[0, 0, 1, 2, 3, 4],
                   [0, 5, 5, 4, 2, 4],
                   [0, 0, 5, 4, 4, 4],
                   [6, 5, 5, 7, 8, 0]])


def test_label_rule_matches_the_vendored_suite_vectors():
    assert np.array_equal(L.label_equal(X), LABELS)                                            # test_basic
    d = np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]])
    assert np.array_equal(L.label_equal(d), d)                                                 # test_diag (:58-62)
    assert np.array_equal(L.label_equal(np.array([[0, 1], [1, 0]])), [[0, 1], [1, 0]])         # test_4_vs_8, connectivity 2
    assert np.array_equal(L.label_equal(np.array([[1, 0, 0], [1, 1, 5], [0, 0, 0]])),          # test_background (:76-83),
                          [[1, 0, 0], [1, 1, 2], [0, 0, 0]])                                   # _label.py:108-115
    assert np.array_equal(L.label_equal(np.array([[0, 0, 6], [0, 0, 6], [5, 5, 5]])),          # test_background_two_regions
                          [[0, 0, 1], [0, 0, 1], [2, 2, 2]])


def test_label_rule_on_random_images_against_per_value_scipy_labelling():
    from scipy import ndimage as ndi
    rng = np.random.default_rng(5)
    for _ in range(5):
        x = rng.integers(0, 5, size=(20, 30))
        lab = L.label_equal(x)
        first = {}
        for v in np.unique(x[x != 0]):
            comp, k = ndi.label(x == v, structure=np.ones((3, 3)))
            for c in range(1, k + 1):
                idx = np.flatnonzero((comp == c).ravel())
                first[idx[0]] = idx
        for rank, start in enumerate(sorted(first), 1):
            assert np.all(lab.ravel()[first[start]] == rank)
        assert lab.max() == len(first) and np.all((lab == 0) == (x == 0))


def test_label_truncates_floats():
    """skimage.measure.label hands non-boolean input to `_ccomp.label_cython`, whose data buffer is typed np.intp: a float
    image is converted to integers first.  detect_lines' image lies in [0, 1], so everything below 1.0 is background and the
    pixels equal to 1.0 form ONE class -- a wide plateau at a constant mid level must not become a component (and so cannot
    become a box), however many equal neighbours it has."""
    x = np.zeros((12, 40))
    x[2:5, :] = 0.5                                       # wide constant mid-level band: would pass the shape filter as a component
    x[8, 3:30] = 1.0                                      # a saturated line
    x[9, 30:35] = 1.0                                     # 8-connected continuation
    x[0, 0] = 0.999999
    lab = L.label_as_skimage(x)
    assert lab.max() == 1 and np.array_equal(lab != 0, x == 1.0)
    assert np.array_equal(L.label_as_skimage(X.astype(np.float64) / 9.0), L.label_equal((X == 9).astype(int)))
    # integers pass through unchanged
    assert np.array_equal(L.label_as_skimage(X), LABELS)
    # a whole frame: labels only where the rescaled image is exactly 1.0, and a mid-level plateau yields no box
    img = np.full((96, 128), 90, np.uint8)
    rng = np.random.default_rng(3)
    img += rng.integers(0, 6, size=img.shape).astype(np.uint8)
    img[30:38, :] = 150                                   # bright constant band, 8 rows x full width
    img[70, 5:120] = 255
    d = L.lines_debug(img)
    assert np.array_equal(d["labels"] != 0, d["rescaled"] == 1.0)
    assert int(d["labels"].max()) >= 1
