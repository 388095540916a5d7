"""Seeded synthetic LFA images: inputs for the parity tests and for bench.py.

Host numpy generators; image i of a set uses seed 1234 + i.  bench.py builds a pool of distinct images once,
uploads it and tiles it on the device to the batch size of the workload."""
from __future__ import annotations

import numpy as np


def synth_rgb(H=1536, W=2048, n_strips=8, seed=1234) -> np.ndarray:
    """(H,W,3) uint8, BGR-like: dark noisy background, `n_strips` bright vertical strips with two darker transverse
    bands each, salt noise, channels differing by a few counts."""
    rng = np.random.default_rng(seed)
    img = rng.normal(40, 6, size=(H, W))
    pitch = W / n_strips
    for s in range(n_strips):
        w = int(W / 40 * rng.uniform(0.9, 1.3))
        h = int(rng.uniform(0.55, 0.7) * H)
        c0 = int(s * pitch + (pitch - w) / 2 + rng.uniform(-0.1, 0.1) * pitch)
        c0 = max(1, min(W - w - 1, c0))
        r0 = int(rng.uniform(0.05, 0.95 - h / H) * H)
        body = rng.normal(200, 8, size=(h, w))
        rows = np.arange(h)[:, None]
        for frac in (0.35, 0.50):
            depth = rng.uniform(20, 80)
            body -= depth * np.exp(-0.5 * ((rows - frac * h) / 6.0) ** 2)
        img[r0:r0 + h, c0:c0 + w] = body
    salt = rng.random((H, W)) < 1e-4
    img[salt] = 255
    r = np.clip(img, 0, 255)
    out = np.empty((H, W, 3), dtype=np.uint8)
    out[..., 2] = r.astype(np.uint8)
    out[..., 1] = np.clip(r + rng.integers(-2, 3, size=(H, W)), 0, 255).astype(np.uint8)
    out[..., 0] = np.clip(r + rng.integers(-2, 3, size=(H, W)), 0, 255).astype(np.uint8)
    return out


def synth_u16(H=2048, W=2048, n_strips=40, seed=1234) -> np.ndarray:
    """(H,W) uint16 with 12-bit statistics like the frame embedded in the reference's images/excel2.xlsx
    (values ~70..3200): `n_strips` strips in two rows, Poisson-like noise so that labelling sees ~1000 small components."""
    rng = np.random.default_rng(seed)
    img = rng.poisson(90, size=(H, W)).astype(np.float64) + rng.normal(0, 12, size=(H, W))
    per_row = (n_strips + 1) // 2
    pitch = W / per_row
    w = int(W / 40)
    h = int(H * 0.42)
    k = 0
    for row in range(2):
        r0 = int(H * (0.04 + 0.5 * row))
        for s in range(per_row):
            if k >= n_strips:
                break
            k += 1
            c0 = int(s * pitch + (pitch - w) / 2 + rng.uniform(-0.15, 0.15) * (pitch - w))
            c0 = max(1, min(W - w - 1, c0))
            level = rng.uniform(900, 2600)
            body = rng.poisson(level, size=(h, w)).astype(np.float64)
            rows = np.arange(h)[:, None]
            for frac in (0.35, 0.50):
                body += rng.uniform(100, 600) * np.exp(-0.5 * ((rows - frac * h) / 7.0) ** 2)
            img[r0:r0 + h, c0:c0 + w] = body
    # faint blotches: many small components after the opening
    for _ in range(H * W // 4000):
        r, c = int(rng.integers(0, H - 16)), int(rng.integers(0, W - 4))
        img[r:r + int(rng.integers(8, 16)), c:c + int(rng.integers(1, 4))] += rng.uniform(600, 1500)
    return np.clip(img, 0, 4095).astype(np.uint16)


def synth_rois(n_images, H, W, per_image=256, seed=99) -> np.ndarray:
    """Dense select-region workload (BASELINE config 5): h~U{64..512}, w~U{16..64}, uniform position.
    Returns an array with fields image, r0, r1, c0, c1 (include/lfa.h LfaRoi)."""
    rng = np.random.default_rng(seed)
    n = n_images * per_image
    h = rng.integers(64, 513, size=n)
    w = rng.integers(16, 65, size=n)
    r0 = (rng.random(n) * (H - h)).astype(np.int64)
    c0 = (rng.random(n) * (W - w)).astype(np.int64)
    out = np.zeros(n, dtype=[("image", "<i4"), ("r0", "<i4"), ("r1", "<i4"), ("c0", "<i4"), ("c1", "<i4")])
    out["image"] = np.repeat(np.arange(n_images), per_image)
    out["r0"], out["r1"], out["c0"], out["c1"] = r0, r0 + h, c0, c0 + w
    return out


def synth_lines(H=384, W=512, seed=1234, dtype=np.uint8, n_lines=2) -> np.ndarray:
    """(H,W) uint8 / uint16 strip-like crop for detect_lines: noisy mid-grey body with a slow gradient and `n_lines` thin,
    slightly tilted, saturated transverse lines.  The function keeps components of EQUAL value after a percentile rescale
    that sets the top 1 % of the pixels to 1.0, whose bounding box is wider than tall and covers 0.65 % - 4 % of the image:
    thin tilted lines (few pixels, tall enough box) are what passes."""
    rng = np.random.default_rng(seed)
    y = np.arange(H)[:, None]
    x = np.arange(W)[None, :]
    img = 90 + 25 * np.sin(y / H * 3.1) + 10 * (x / W) + rng.normal(0, 5, size=(H, W))
    need = int(np.ceil(0.0065 * H)) + 1                       # rows a full-width box needs to pass min_area
    for b in range(n_lines):
        r0 = int((0.1 + 0.8 * (b + rng.uniform(0.3, 0.7)) / n_lines) * H)
        c0, c1 = int(rng.uniform(0.0, 0.04) * W), int(rng.uniform(0.96, 1.0) * W)
        tilt = need + int(rng.integers(0, 3))
        cols = np.arange(c0, c1)
        rows = np.clip(r0 + (cols - c0) * tilt // max(1, c1 - c0), 0, H - 1)
        img[rows, cols] = 255
    img = np.clip(img, 0, 255)
    if np.dtype(dtype) == np.uint16:
        return (img.astype(np.uint16) * 257)
    return img.astype(np.uint8)
