#!/usr/bin/env python3
"""Extended detect_lines fuzz on the GPU box (not a pytest test: minutes of oracle time): random frames of random shape --
rows of a multiple of 8 pixels and not, pixel counts that are / are not multiples of 4 and 8 (the 16-byte-load paths of the
labelling kernels), shapes below one CLAHE tile -- every intermediate (uint16 image, levels, roots) and the boxes against
oracle/lines_oracle.py.  usage: python tools/lines_fuzz.py [cases] [seed] [largest side]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def run(cases, seed, top=300):
    import image_analysis_amd._capi as capi
    from oracle import lines_oracle as L
    fz = _load(os.path.join(ROOT, "tests", "test_gpu_fuzz_round2.py"), "fz")
    tl = _load(os.path.join(ROOT, "tests", "test_gpu_lines.py"), "tl")
    eng = capi.Engine(device=0, max_h=top + 20, max_w=top + 20, max_c=1, max_batch=3)
    eng.set_debug_flags(64)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    refused = boxes = 0
    for it in range(cases):
        H, W = int(rng.integers(2, top)), int(rng.integers(2, top))
        if it % 3 == 0:
            W = max(8, W & ~7)
        if it % 5 == 0:
            H = max(4, H & ~3)
        im = fz._frame(rng, H, W, np.uint8 if it % 2 else np.uint16)
        if it % 7 == 0:                                             # a few long bright bars: components that span waves and rows
            for _ in range(int(rng.integers(1, 4))):
                r = int(rng.integers(0, H)); h = int(rng.integers(1, max(2, H // 8)))
                im[r:r + h, int(rng.integers(0, max(1, W // 4))):] = im.max()
        try:
            L.lines_debug(im)
        except ValueError:
            refused += 1
            try:
                eng.detect_lines(im[None])
            except Exception:
                pass
            continue
        if it % 11 == 0:
            eng.set_debug_flags(64 | 128)
        try:
            bx0, info0 = eng.detect_lines(im[None])
            want0 = L.lines_debug(im)["strips"]
            got0 = [[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in bx0[0]]
            if got0 != want0:
                print("FIRST CALL case", it, H, W, "got", got0[:8], "want", want0[:8], info0[0], flush=True)
            d = tl._against_oracle(capi, eng, im)
        except AssertionError:
            bx, info = eng.detect_lines(im[None])
            want = L.lines_debug(im)["strips"]
            print("MISMATCH case", it, H, W, im.dtype, "flags128" if it % 11 == 0 else "", flush=True)
            print(" got ", [[int(b["minr"]), int(b["maxr"])] for b in bx[0]][:12], len(bx[0]), info[0])
            print(" want", [w[:2] for w in want][:12], len(want), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.save(os.path.join(ROOT, "gpurun_out", f"lines_fuzz_fail_{seed}_{it}.npy"), im)
            raise
        eng.set_debug_flags(64)
        boxes += len(d["strips"])
        if it % 4 == 0:                                             # the same frame beside two variants of it in one call: boxes only
            batch = np.stack([im, np.ascontiguousarray(im[::-1]), np.ascontiguousarray(im[:, ::-1])])
            try:
                want = [L.detect_lines(b) for b in batch]
            except ValueError:
                want = None
            if want is not None:
                bx, _info = eng.detect_lines(batch)
                got = [[[int(b["minr"]), int(b["maxr"]), int(b["minc"]), int(b["maxc"])] for b in one] for one in bx]
                assert got == want, ("batch", it, H, W)
        if it % 25 == 0:
            print(f"case {it}: {H}x{W} ok, {boxes} boxes so far, {time.time() - t0:.0f} s", flush=True)
    print(f"lines fuzz ok: {cases} cases, {refused} refused by the reference as well, {boxes} boxes, {time.time() - t0:.0f} s")
    eng.close()
    return boxes


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
        int(sys.argv[3]) if len(sys.argv) > 3 else 300)
