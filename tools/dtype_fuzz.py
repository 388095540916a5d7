#!/usr/bin/env python3
"""Extended fuzz of the scalar float64 paths on the GPU box, not a pytest test: the generators of
tests/test_gpu_fuzz_round2.py (uint16 colour images, float64 images 2-D and colour with foreground corners, regions of the
other dtypes through image_analysis / inverse_image_analysis) over many more seeds, through the drop-in module.
usage: python tools/dtype_fuzz.py [seeds]"""
import importlib.util
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(seeds):
    spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz_round2.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    t0 = time.time()
    a = b = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in range(1, seeds + 1):
            try:
                a += fz.uint16_colour_detect_fuzz(seed, 12)
                b += fz.float64_image_detect_fuzz(seed, 16)
                fz.region_dtype_fuzz(seed, 42)
            except AssertionError:
                print("MISMATCH at seed", seed, flush=True)
                raise
            if seed % 10 == 0:
                print(f"seed {seed} ok, {time.time() - t0:.0f} s", flush=True)
    print(f"dtype fuzz ok: {seeds} seeds; {a} strips from uint16 colour images, {b} from float64 images, {42 * seeds} regions of other dtypes; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
