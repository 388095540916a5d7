#!/usr/bin/env python3
"""Latency of ONE image through the whole call (the reference GUI's case: one photo, detect + profile + peaks), resident in
HBM and from host memory, on a warm handle.  usage: python tools/single_image_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_u16
    for name, im in (("1536x2048 RGB uint8", synth_rgb(1536, 2048, 8, 1)), ("2048x2048 uint16", synth_u16(2048, 2048, 40, 2))):
        H, W = im.shape[:2]
        Cc = 1 if im.ndim == 2 else 3
        eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=Cc, max_batch=1)
        mm_c = int(W / 40); mm_r = mm_c * 20
        stride = int(H * .6) + 1
        t = capi.to_device(im[None], eng.device)
        code = capi.np_dtype_code(im.dtype)
        for _ in range(5):
            eng.analyze_raw(t, code, 1, H, W, Cc, mm_r, mm_c, 10, 64, stride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.analyze_raw(t, code, 1, H, W, Cc, mm_r, mm_c, 10, 64, stride)
            torch.cuda.synchronize()
        resident = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(20):
            res, _ = eng.analyze(im[None], mm_r, mm_c)
        e2e = (time.perf_counter() - t0) / 20
        print(f"{name}: {resident * 1e3:.3f} ms resident in HBM (call + sync), {e2e * 1e3:.2f} ms from a numpy array to numpy results "
              f"({len(res[0][0])} strips)", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
