#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): BASELINE.json configs[2] (16-bit frames), configs[4] (dense
select-region ROIs) and the Gaussian stage, each with per-kernel times and GB/s against its algorithmic bytes.
    python benchmarks_configs.py [u16|rois|gauss] ..."""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__file__) or ".")
capi = importlib.import_module("image_analysis_amd._capi")
synth = importlib.import_module("image_analysis_amd.synth")
import ctypes as C

dev = torch.device("cuda", 0)


def timed(fn, steps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def bench_u16(B=1024, CH=256, pool=8, real=False):
    """configs[2]: 2048x2048 uint16 frames, chunks alternating on two streams / handles like bench.py.
    real=True: the reference's own 12-bit frame (tests/golden/excel2_frame.png, images/excel2.xlsx), rolled per pool slot."""
    H = W = 2048
    if real:
        import os
        from PIL import Image
        f = np.array(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "excel2_frame.png")))
        pool_np = np.stack([np.ascontiguousarray(np.roll(f, (17 * i, 64 * i), (0, 1))) for i in range(pool)])
    else:
        pool_np = np.stack([synth.synth_u16(H, W, 40, 1234 + i) for i in range(pool)])
    p = torch.from_numpy(pool_np.view(np.uint8).reshape(pool, -1)).to(dev)
    batch = p[torch.arange(B, device=dev) % pool].contiguous()
    ms, stride = 64, int(H * .6) + 1
    NS = int(__import__("os").environ.get("LFA_STREAMS", "2"))
    engs = [capi.Engine(device=0, max_h=H, max_w=W, max_c=1, max_batch=CH) for _ in range(NS)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    strips = capi.dev_bytes(B * ms * 24, dev); counts = torch.zeros(B, dtype=torch.int32, device=dev)
    info = capi.dev_bytes(B * 104, dev); rois = capi.dev_bytes(B * ms * 20, dev)
    prof = torch.empty(B * ms * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(B * ms * 40, dev)
    ib = H * W * 2
    torch.cuda.synchronize()

    STAGGER = int(__import__("os").environ.get("LFA_STAGGER", "0"))

    def step(ns=NS):
        for ci, b0 in enumerate(range(0, B, CH)):
            if STAGGER and ns > 1:              # this chunk's passes start when the previous chunk's histogram pass is done
                capi._check(capi.lib.lfa_stream_wait_passes(engs[(ci - 1) % ns]._h, C.c_void_p(streams[ci % ns].cuda_stream)))
            capi._check(capi.lib.lfa_analyze_batch(engs[ci % ns]._h, batch.data_ptr() + b0 * ib, capi.LFA_U16, CH, H, W, 1, 1020, 51, 10,
                        strips.data_ptr() + b0 * ms * 24, ms, counts.data_ptr() + b0 * 4, info.data_ptr() + b0 * 104,
                        rois.data_ptr() + b0 * ms * 20, prof.data_ptr() + b0 * ms * stride * 8, stride,
                        peaks.data_ptr() + b0 * ms * 40, C.c_void_p(streams[ci % ns].cuda_stream)))
    for _ in range(3):                                          # the handles settle their choices (one-pass histogram, class plane, LDS slots)
        step(); torch.cuda.synchronize()
    dt = timed(step)
    engs[0].timing("*"); step(1); torch.cuda.synchronize(); kt = engs[0].timing_read(); engs[0].timing(None)
    ns = int(counts.sum().item())
    inf = capi.from_device(info, capi.INFO_DT, B)
    print(json.dumps({"config": "the reference's 12-bit frame excel2, 2048x2048 uint16" if real else "configs[2] 2048x2048 uint16 frames", "streams": NS, "chunk": CH, "frames": B, "ms_per_step": round(dt * 1e3, 3),
                      "frames_per_s": round(B / dt, 1), "strips_per_s": round(ns / dt, 1), "labels_mean": float(inf["n_labels"].mean()), "runs_mean": float(inf["n_runs"].mean()),
                      "e2e_frac_hbm": round(B / dt * ib / 8e12, 4),
                      "kernel_ms_one_stream": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}}))
    for e in engs:
        e.close()


def bench_big(B=128, CH=64, pool=4):
    """configs[3]: 3072x4096 colour images (the reference's phone-camera size), 2 chunk streams like bench.py."""
    H, W = 3072, 4096
    pool_np = np.stack([synth.synth_rgb(H, W, 8, 77 + i) for i in range(pool)])
    p = torch.from_numpy(pool_np.reshape(pool, -1)).to(dev)
    batch = p[torch.arange(B, device=dev) % pool].contiguous()
    ms, stride = 16, int(H * .6) + 1
    mm_c = int(W / 40); mm_r = mm_c * 20
    engs = [capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=CH) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    strips = capi.dev_bytes(B * ms * 24, dev); counts = torch.zeros(B, dtype=torch.int32, device=dev)
    info = capi.dev_bytes(B * 104, dev); rois = capi.dev_bytes(B * ms * 20, dev)
    prof = torch.empty(B * ms * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(B * ms * 40, dev)
    ib = H * W * 3
    torch.cuda.synchronize()

    def step():
        for ci, b0 in enumerate(range(0, B, CH)):
            capi._check(capi.lib.lfa_analyze_batch(engs[ci % 2]._h, batch.data_ptr() + b0 * ib, capi.LFA_U8, CH, H, W, 3, mm_r, mm_c, 10,
                        strips.data_ptr() + b0 * ms * 24, ms, counts.data_ptr() + b0 * 4, info.data_ptr() + b0 * 104,
                        rois.data_ptr() + b0 * ms * 20, prof.data_ptr() + b0 * ms * stride * 8, stride,
                        peaks.data_ptr() + b0 * ms * 40, C.c_void_p(streams[ci % 2].cuda_stream)))
    dt = timed(step)
    ns = int(counts.sum().item())
    inf = capi.from_device(info, capi.INFO_DT, B)
    assert (inf["status"] == 0).all()
    print(json.dumps({"config": "configs[3] 3072x4096 RGB uint8 images", "images": B, "chunk": CH, "ms_per_step": round(dt * 1e3, 3),
                      "images_per_s": round(B / dt, 1), "strips_per_s": round(ns / dt, 1), "ambiguous": int(inf["ambiguous"].sum()),
                      "e2e_frac_hbm": round(B / dt * ib / 8e12, 4)}))
    for e in engs:
        e.close()


def bench_photo(B=1024, CH=256, pool=8):
    """configs[1]'s shape on photo-like content: the decoded reference sample tests/golden/image1_decoded.npz (512x512 grey,
    what cv2.imread turns into three equal channels), tiled to 1536x2048x3 with a different shift per image.  Many pixels
    share the Otsu threshold's class here, so the handles stop writing the class plane and the mask pass compares the
    image."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "image1_decoded.npz"))["gray"]
    H, W = 1536, 2048
    tile = np.tile(g, (3, 4))
    pool_np = np.stack([np.repeat(np.roll(tile, (37 * i, 91 * i), (0, 1))[:, :, None], 3, axis=2) for i in range(pool)])
    p = torch.from_numpy(pool_np.reshape(pool, -1)).to(dev)
    batch = p[torch.arange(B, device=dev) % pool].contiguous()
    ms, stride = 64, int(H * .6) + 1
    engs = [capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=CH) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    strips = capi.dev_bytes(B * ms * 24, dev); counts = torch.zeros(B, dtype=torch.int32, device=dev)
    info = capi.dev_bytes(B * 104, dev); rois = capi.dev_bytes(B * ms * 20, dev)
    prof = torch.empty(B * ms * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(B * ms * 40, dev)
    ib = H * W * 3
    torch.cuda.synchronize()

    def step(ns=2):
        for ci, b0 in enumerate(range(0, B, CH)):
            capi._check(capi.lib.lfa_analyze_batch(engs[ci % ns]._h, batch.data_ptr() + b0 * ib, capi.LFA_U8, CH, H, W, 3, 1020, 51, 10,
                        strips.data_ptr() + b0 * ms * 24, ms, counts.data_ptr() + b0 * 4, info.data_ptr() + b0 * 104,
                        rois.data_ptr() + b0 * ms * 20, prof.data_ptr() + b0 * ms * stride * 8, stride,
                        peaks.data_ptr() + b0 * ms * 40, C.c_void_p(streams[ci % ns].cuda_stream)))
    step(); torch.cuda.synchronize(); step(); torch.cuda.synchronize()      # the handles settle their class-plane choice
    dt = timed(step)
    engs[0].timing("*"); step(1); torch.cuda.synchronize(); kt = engs[0].timing_read(); engs[0].timing(None)
    inf = capi.from_device(info, capi.INFO_DT, B)
    assert (inf["status"] == 0).all()
    print(json.dumps({"config": "photo-like colour images 1536x2048x3 (reference sample image1, tiled)", "images": B,
                      "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1), "strips_per_s": round(int(counts.sum().item()) / dt, 1),
                      "labels_mean": float(inf["n_labels"].mean()), "e2e_frac_hbm": round(B / dt * ib / 8e12, 4),
                      "kernel_ms_one_stream": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:8]}}))
    for e in engs:
        e.close()


def bench_rois(B=64, per=256):
    H, W = 1536, 2048
    pool_np = np.stack([synth.synth_rgb(H, W, 8, 1234 + i) for i in range(8)])
    p = torch.from_numpy(pool_np.reshape(8, -1)).to(dev)
    batch = p[torch.arange(B, device=dev) % 8].contiguous()
    rois = synth.synth_rois(B, H, W, per).astype(capi.ROI_DT)
    rt = capi.to_device(rois, dev)
    n = len(rois)
    stride = 512
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=B)
    prof = torch.empty(n * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(n * 40, dev)
    st = capi.current_stream_ptr(dev)

    def step():
        capi._check(capi.lib.lfa_profile_batch(eng._h, batch.data_ptr(), capi.LFA_U8, B, H, W, 3, rt.data_ptr(), n, 0, 15,
                                               prof.data_ptr(), stride, peaks.data_ptr(), st))
    dt = timed(step)
    eng.timing("*"); step(); torch.cuda.synchronize(); kt = eng.timing_read(); eng.timing(None)
    byts = int(((rois["r1"] - rois["r0"]).astype(np.int64) * (rois["c1"] - rois["c0"]) * 3).sum())
    print(json.dumps({"config": "configs[4] dense select-region: 256 ROIs/image", "images": B, "rois": n,
                      "ms_per_step": round(dt * 1e3, 3), "rois_per_s": round(n / dt, 1), "window_bytes": byts,
                      "GBps_algorithmic": round(byts / dt / 1e9, 1), "frac_hbm": round(byts / dt / 8e12, 4),
                      "kernel_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}}))
    eng.close()


def bench_gauss(B=64):
    for (H, W, Cc, dt_np, f32) in ((1536, 2048, 3, np.uint8, 0), (2048, 2048, 1, np.uint16, 0), (1536, 2048, 3, np.uint8, 1)):
        rng = np.random.default_rng(0)
        img = rng.integers(0, 256, size=(4, H, W, Cc) if Cc == 3 else (4, H, W)).astype(dt_np)
        p = torch.from_numpy(img.view(np.uint8).reshape(4, -1)).to(dev)
        batch = p[torch.arange(B, device=dev) % 4].contiguous()
        eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=B)
        out = torch.empty(B * H * W * Cc, dtype=torch.float32 if f32 else torch.float64, device=dev)
        st = capi.current_stream_ptr(dev)
        code = capi.np_dtype_code(dt_np)

        def step():
            capi._check(capi.lib.lfa_gaussian_batch(eng._h, batch.data_ptr(), code, B, H, W, Cc, 1.0, None, 0, out.data_ptr(), f32, st))
        dt = timed(step)
        byts = B * H * W * Cc * (np.dtype(dt_np).itemsize + (4 if f32 else 8))
        print(json.dumps({"config": f"gaussian sigma=1 {H}x{W}x{Cc} {np.dtype(dt_np).name} -> {'f32' if f32 else 'f64'}", "images": B,
                          "ms_per_step": round(dt * 1e3, 3), "GBps_algorithmic": round(byts / dt / 1e9, 1),
                          "frac_hbm": round(byts / dt / 8e12, 4)}))
        eng.close()


def bench_peaks(n=16384, length=288):
    rng = np.random.default_rng(0)
    x = np.arange(length)
    prof = 180 - 40 * np.exp(-0.5 * ((x - 0.35 * length) / 6) ** 2) - 25 * np.exp(-0.5 * ((x - 0.5 * length) / 6) ** 2)
    buf = np.round((prof[None, :] + rng.normal(0, 2.0, size=(n, length))) * 60) / 60
    bt = capi.to_device(buf, dev); lt = capi.to_device(np.full(n, length, np.int32), dev)
    out = capi.dev_bytes(n * 40, dev)
    eng = capi.Engine(device=0, max_h=64, max_w=64, max_c=1, max_batch=1)
    st = capi.current_stream_ptr(dev)
    for two in (0, 1):
        def step():
            capi._check(capi.lib.lfa_peaks_batch(eng._h, bt.data_ptr(), lt.data_ptr(), n, length, 0, 25, two, 15, out.data_ptr(), st))
        dt = timed(step, 5)
        pk = capi.from_device(out, capi.PEAKS_DT, n)
        print(json.dumps({"config": f"k_peaks two={two}", "profiles": n, "length": length, "ms": round(dt * 1e3, 3),
                          "profiles_per_s": round(n / dt, 1), "mean_peak_width": float((pk["end"][:, 0] - pk["start"][:, 0]).mean())}))
    eng.close()


def bench_stream(n=1024, batch=64, pool=8):
    """PCIe-inclusive rate: frames start in pageable host memory (numpy), frames.analyze_stream packs them into pinned
    batches, uploads and analyses them double-buffered.  Never the headline number (bench.py's `value` has inputs in HBM)."""
    from image_analysis_amd import frames as F
    from image_analysis_amd.synth import synth_rgb
    imgs = [synth_rgb(1536, 2048, 8, 1234 + i) for i in range(pool)]
    list(F.analyze_stream((imgs[i % pool] for i in range(batch)), batch=batch))          # warm-up: handle, pinned buffers
    t0 = time.perf_counter()
    strips = sum(len(r[1]) for r in F.analyze_stream((imgs[i % pool] for i in range(n)), batch=batch))
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "host frames -> analyze_stream (pinned double-buffered upload)", "frames": n, "batch": batch,
                      "seconds": round(dt, 3), "frames_per_s": round(n / dt, 1), "strips_per_s": round(strips / dt, 1),
                      "host_to_device_GBps": round(n * imgs[0].nbytes / dt / 1e9, 2)}))


if __name__ == "__main__":
    what = sys.argv[1:] or ["u16", "rois", "gauss"]
    if "u16" in what: bench_u16(CH=int(__import__("os").environ.get("LFA_CHUNK", "256")))
    if "u16real" in what: bench_u16(real=True)
    if "big" in what: bench_big()
    if "photo" in what: bench_photo()
    if "rois" in what: bench_rois()
    if "gauss" in what: bench_gauss()
    if "peaks" in what: bench_peaks()
    if "stream" in what: bench_stream()
