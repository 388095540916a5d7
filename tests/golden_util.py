"""Helpers shared by the tests that read tests/golden/ref_skimage_*.npz (made by tests/golden/make_golden.py from the
reference's detection.py run over its vendored scikit-image sources)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


def skimage_detect_cases():
    g = np.load(os.path.join(GOLDEN, "ref_skimage_detect.npz"))
    return g, [str(n) for n in g["names"]]


def clipped(strips, H, W):
    """[minr, maxr, minc, maxc] boxes as numpy's slicing clips them (detection.py:92)."""
    return [[a, min(b, H), c, min(d, W)] for a, b, c, d in strips]
