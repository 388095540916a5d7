#!/usr/bin/env python3
"""Extended fuzz of lfa_analyze_batch_static (the entry point that allocates nothing) against lfa_analyze_batch on the GPU
box, not a pytest test: a long random sequence of batches of every shape, dtype and size through one handle and the one
shared scratch array; every output byte of the two entry points compared (no oracle needed, so it is fast).
usage: python tools/static_fuzz.py [rounds] [seed]"""
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


def same(a, b):
    cn, st, ro, pk, pr, inf = a
    cn2, st2, ro2, pk2, pr2, inf2 = b
    assert np.array_equal(cn, cn2)
    for i in range(len(cn)):
        k = cn[i]
        assert st[i, :k].tobytes() == st2[i, :k].tobytes()
        assert ro[i, :k].tobytes() == ro2[i, :k].tobytes()
        assert pk[i, :k].tobytes() == pk2[i, :k].tobytes()
        for q in range(k):
            L = max(0, int(pk[i, q]["length"]))
            assert pr[i, q, :L].tobytes() == pr2[i, q, :L].tobytes()
    assert inf.tobytes() == inf2.tobytes()
    return int(cn.sum())


def run(rounds, seed):
    import torch
    import image_analysis_amd._capi as capi
    tz = _load(os.path.join(ROOT, "tests", "test_zz_no_alloc.py"), "tz")
    dev = torch.device("cuda", 0)
    eng = capi.Engine(device=0, max_h=512, max_w=512, max_c=3, max_batch=4)
    h = eng._h
    rng = np.random.default_rng(seed)
    t0 = time.time()
    strips = 0
    for it in range(rounds):
        n = int(rng.integers(1, 5))
        kind = int(rng.integers(0, 3))
        H, W = int(rng.integers(4, 513)), int(rng.integers(4, 513))
        if rng.random() < 0.5:
            W = max(16, (W // 16) * 16)
        top = 255 if kind != 1 else int(rng.choice([255, 4095, 65535]))
        shape = (n, H, W, 3) if kind == 2 else (n, H, W)
        batch = rng.integers(0, max(2, top // 3), size=shape).astype(np.int64)
        for i in range(n):
            for _b in range(int(rng.integers(0, 7))):
                c0, w = int(rng.integers(0, W)), int(rng.integers(1, max(2, W // 6)))
                r0, hh = int(rng.integers(0, max(1, H // 2))), int(rng.integers(max(1, H // 4), H + 1))
                batch[i, r0:r0 + hh, c0:c0 + w] += int(rng.integers(top // 3, top // 2))
        batch = np.ascontiguousarray(np.clip(batch, 0, top).astype(np.uint16 if kind == 1 else np.uint8))
        mm_r, mm_c = int(rng.integers(0, max(1, H // 2))), int(rng.integers(0, max(1, W // 8)))
        capi_gray = eng._gray_for(W, 3 if kind == 2 else 1)
        try:
            want = tz._call(capi, capi.lib.lfa_analyze_batch, h, dev, batch, mm_r, mm_c)
            got = tz._call(capi, capi.lib.lfa_analyze_batch_static, h, dev, batch, mm_r, mm_c)
            strips += same(want, got)
        except AssertionError:
            print("MISMATCH round", it, batch.shape, batch.dtype, mm_r, mm_c, flush=True)
            raise
        if it % 100 == 0:
            print(f"round {it} ok, {strips} strips, {time.time() - t0:.0f} s", flush=True)
    eng.close()
    print(f"static fuzz ok: {rounds} rounds, {strips} strips with profiles and peaks identical, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
