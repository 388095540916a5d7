#!/usr/bin/env python3
"""Extended fuzz of the whole analyze call (detect + profile + peaks) and of the peak search on the GPU box, not a pytest
test: random batches of random shape and dtype through ONE handle, strips, profiles and peaks against the oracle
(tests/test_gpu_profile.py::_check_analyze), then the peak-logic fuzz of the same file over many more seeds.
usage: python tools/analyze_fuzz.py [batches] [peak_seeds]"""
import importlib.util
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def run(batches, peak_seeds):
    import image_analysis_amd._capi as capi
    tp = _load(os.path.join(ROOT, "tests", "test_gpu_profile.py"), "tp")
    eng = capi.Engine(device=0, max_h=320, max_w=320, max_c=3, max_batch=4)
    rng = np.random.default_rng(99)
    t0 = time.time()
    analysed = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(batches):
            n = int(rng.integers(1, 5))
            kind = int(rng.integers(0, 3))                           # 0: uint8 2-D, 1: uint16 2-D, 2: colour
            H, W = int(rng.integers(30, 320)), int(rng.integers(8, 320))
            if rng.random() < 0.5:
                W = max(16, (W // 16) * 16)
            top = 255 if kind != 1 else int(rng.choice([255, 4095, 65535]))
            shape = (n, H, W, 3) if kind == 2 else (n, H, W)
            batch = rng.integers(0, max(2, top // 3), size=shape).astype(np.int64)
            for i in range(n):
                for _b in range(int(rng.integers(1, 7))):            # tall bright strips with darker bands across them
                    c0, w = int(rng.integers(0, W)), int(rng.integers(1, max(2, W // 6)))
                    r0, h = int(rng.integers(0, H // 2)), int(rng.integers(H // 4, H))
                    batch[i, r0:r0 + h, c0:c0 + w] += int(rng.integers(top // 3, top // 2))
                    for _l in range(int(rng.integers(0, 3))):
                        rr = int(rng.integers(r0, min(H, r0 + h)))
                        batch[i, rr:rr + int(rng.integers(1, 6)), c0:c0 + w] -= int(rng.integers(0, top // 4))
            batch = np.ascontiguousarray(np.clip(batch, 0, top).astype(np.uint16 if kind == 1 else np.uint8))
            mm_r, mm_c = int(rng.integers(0, H // 2)), int(rng.integers(0, max(1, W // 8)))
            try:
                k, _ = tp._check_analyze(eng, batch, mm_r, mm_c)
            except AssertionError:
                np.save(os.path.join(ROOT, "gpurun_out", f"analyze_fuzz_fail_{it}.npy"), batch)
                print("MISMATCH at batch", it, batch.shape, batch.dtype, mm_r, mm_c, flush=True)
                raise
            analysed += k
            if it % 50 == 0:
                print(f"batch {it} ok, {analysed} strips analysed, {time.time() - t0:.0f} s", flush=True)
        eng.close()
        for seed in range(4, 4 + peak_seeds):
            tp.test_random_profiles_peak_logic_against_oracle(seed)
    print(f"analyze fuzz ok: {batches} batches on one handle, {analysed} strips analysed; peak logic seeds 4..{3 + peak_seeds}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
