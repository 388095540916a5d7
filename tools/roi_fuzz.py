#!/usr/bin/env python3
"""Extended fuzz of lfa_profile_batch on the GPU box, not a pytest test: random images, random ROI sets -- dense and sorted
by image (the band kernel), shuffled (the per-ROI kernel), a handful, degenerate ones -- plain and inverse analysis, both
peak ranges, every profile and peak against the oracle, on ONE handle.  usage: python tools/roi_fuzz.py [rounds] [seed]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rounds, seed):
    import image_analysis_amd._capi as capi
    from oracle import lfa_oracle as O
    eng = capi.Engine(device=0, max_h=256, max_w=320, max_c=3, max_batch=4)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    total = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(rounds):
            n = int(rng.integers(1, 5))
            kind = int(rng.integers(0, 3))
            H, W = int(rng.integers(4, 257)), int(rng.integers(4, 321))
            if rng.random() < 0.6:
                W = max(16, (W // 16) * 16)                          # the band kernel wants aligned rows
            top = 255 if kind != 1 else int(rng.choice([255, 4095, 65535]))
            shape = (n, H, W, 3) if kind == 2 else (n, H, W)
            imgs = rng.integers(0, top + 1, size=shape)
            if rng.random() < 0.5:                                   # smooth columns with a few bands: real peaks, ties
                ramp = (np.sin(np.arange(H) / rng.uniform(2, 30)) * 0.4 + 0.5) * top
                imgs = (imgs * 0.05 + ramp.reshape((1, H, 1) + ((1,) if kind == 2 else ()))).astype(np.int64)
            imgs = np.ascontiguousarray(np.clip(imgs, 0, top).astype(np.uint16 if kind == 1 else np.uint8))
            per = int(rng.choice([1, 3, 40, 70]))
            m = n * per
            rois = np.zeros(m, dtype=capi.ROI_DT)
            rois["image"] = np.repeat(np.arange(n), per)
            h = rng.integers(1, H + 1, size=m); w = rng.integers(1, min(W, 64) + 1, size=m)
            rois["r0"] = (rng.random(m) * (H - h + 1)).astype(int); rois["r1"] = rois["r0"] + h
            rois["c0"] = (rng.random(m) * (W - w + 1)).astype(int); rois["c1"] = rois["c0"] + w
            if m > 4:
                rois["r1"][1] = rois["r0"][1]                        # no rows
                rois["c1"][2] = rois["c0"][2]                        # no columns
            if rng.random() < 0.5:
                rois = rois[rng.permutation(m)]
            inverse = kind == 2 and rng.random() < 0.5               # inverse_image_analysis takes colour regions only
            r = int(rng.choice([15, 1]))
            profs, pk = eng.profiles(imgs, rois, inverse=inverse, rng=r)
            for q, roi in enumerate(rois):
                reg = imgs[roi["image"], roi["r0"]:roi["r1"], roi["c0"]:roi["c1"]]
                try:
                    if reg.shape[0] == 0:
                        assert len(profs[q]) == 0 and pk[q]["n_peaks"] == 0
                        continue
                    wp, wk = (O.inverse_image_analysis(reg, True, r) if inverse else O.image_analysis(reg, True, r))
                    assert np.array_equal(profs[q], wp, equal_nan=True)
                    if reg.shape[1] == 0:
                        continue
                    assert pk[q]["n_peaks"] == len(wk)
                    for i, (s, e, a) in enumerate(wk):
                        assert (pk[q]["start"][i], pk[q]["end"][i]) == (s, e)
                        assert pk[q]["area"][i] == a or (np.isnan(a) and np.isnan(pk[q]["area"][i]))
                except AssertionError:
                    print("MISMATCH round", it, "roi", q, roi, imgs.shape, imgs.dtype, "inverse", inverse, "rng", r, flush=True)
                    raise
                total += 1
            if it % 50 == 0:
                print(f"round {it} ok, {total} ROIs, {time.time() - t0:.0f} s", flush=True)
    eng.close()
    print(f"roi fuzz ok: {rounds} rounds on one handle, {total} ROIs against the oracle, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
