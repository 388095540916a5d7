#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs[2] (16-bit frames), configs[3] (3072x4096 images), configs[4] (dense
select-region ROIs), photo-like content and the Gaussian stage, each with per-kernel HIP-event times and GB/s against
its algorithmic bytes.  bench.py runs `secondary()` after its headline timed region and puts the
results into the "configs" object of its JSON line, so they are on the driver's clock; by hand:
    python benchmarks_configs.py [u16|u16real|big|photo|rois|gauss|peaks|lines|stream] ..."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
capi = importlib.import_module("image_analysis_amd._capi")
synth = importlib.import_module("image_analysis_amd.synth")

HBM_PEAK_GBS = 8000.0


def DEV():
    """the current HIP device (bench.py selects it per rank before calling in here)"""
    return torch.device("cuda", torch.cuda.current_device())


def timed(fn, steps=10, barrier=None):
    """seconds per step; barrier (multi-rank runs): called on both sides of the timed steps, after the device sync"""
    fn(); torch.cuda.synchronize()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return (time.perf_counter() - t0) / steps


def dominant(kt, alg_bytes_of=None):
    """The kernel with the largest summed HIP-event time of a lfa_timing_read dict: name, ms per launch, and -- given its
    algorithmic bytes per launch -- GB/s and the fraction of the 8 TB/s HBM peak."""
    if not kt:
        return None
    k = max(kt, key=lambda q: kt[q][0])
    ms = kt[k][0] / max(kt[k][1], 1)
    d = {"kernel": k, "ms_per_launch": round(ms, 4), "launches": kt[k][1]}
    b = alg_bytes_of(k) if alg_bytes_of else None
    if b:
        d["algorithmic_bytes_per_launch"] = int(b)
        d["GBps"] = round(b / (ms * 1e-3) / 1e9, 1)
        d["frac_hbm"] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return d


def _kernel_table(kt, top=8):
    return {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:top]}


class _Pipeline:
    """B resident images in chunks of CH alternating over NS handles / streams through lfa_analyze_batch, like bench.py."""

    def __init__(self, batch, code, B, CH, H, W, Cc, mm_r, mm_c, ms, NS=None):
        dev = DEV()
        # bench.py's schedule: three lanes, the image passes of consecutive chunks chained (lfa_chain_passes), their tails
        # beside the next chunk's passes.  LFA_STREAMS / LFA_CHAIN=0 (lanes run free) / LFA_STAGGER=1 for comparisons.
        NS = NS or int(os.environ.get("LFA_STREAMS", "3"))
        NS = max(1, min(NS, (B + CH - 1) // CH))
        self.batch, self.code, self.B, self.CH, self.H, self.W, self.Cc = batch, code, B, CH, H, W, Cc
        self.mm_r, self.mm_c, self.ms, self.NS = mm_r, mm_c, ms, NS
        self.stride = int(H * .6) + 1
        self.ib = H * W * Cc * (2 if code == capi.LFA_U16 else 1)
        self.engs = [capi.Engine(device=dev.index, max_h=H, max_w=W, max_c=Cc, max_batch=CH) for _ in range(NS)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
        self.strips = capi.dev_bytes(B * ms * 24, dev); self.counts = torch.zeros(B, dtype=torch.int32, device=dev)
        self.info = capi.dev_bytes(B * 104, dev); self.rois = capi.dev_bytes(B * ms * 20, dev)
        self.prof = torch.empty(B * ms * self.stride, dtype=torch.float64, device=dev); self.peaks = capi.dev_bytes(B * ms * 40, dev)
        self.stagger = int(os.environ.get("LFA_STAGGER", "0"))
        # (two lanes: free-running measured 3 % faster than chained on configs[3]; three lanes chained is bench.py's schedule)
        self.chain = NS > 1 and not self.stagger and os.environ.get("LFA_CHAIN", "1" if NS >= 3 else "0") == "1"
        if self.chain:
            for i in range(NS):
                capi._check(capi.lib.lfa_chain_passes(self.engs[i]._h, self.engs[i - 1]._h))
        self.turn = 0                    # chunks take the lanes in turn across steps too
        torch.cuda.synchronize()

    def step(self, ns=None):
        ns = ns or self.NS
        ms, st = self.ms, self.stride
        for b0 in range(0, self.B, self.CH):
            ci, self.turn = self.turn, self.turn + 1
            nb = min(self.CH, self.B - b0)
            stream = C.c_void_p(self.streams[ci % ns].cuda_stream)
            if self.stagger and ns > 1:          # this chunk's passes start when the previous chunk's histogram pass is done
                capi._check(capi.lib.lfa_stream_wait_passes(self.engs[(ci - 1) % ns]._h, stream))
            capi._check(capi.lib.lfa_analyze_batch(
                self.engs[ci % ns]._h, self.batch.data_ptr() + b0 * self.ib, self.code, nb, self.H, self.W, self.Cc,
                self.mm_r, self.mm_c, 10, self.strips.data_ptr() + b0 * ms * 24, ms, self.counts.data_ptr() + b0 * 4,
                self.info.data_ptr() + b0 * 104, self.rois.data_ptr() + b0 * ms * 20, self.prof.data_ptr() + b0 * ms * st * 8, st,
                self.peaks.data_ptr() + b0 * ms * 40, stream))

    def kernel_times(self):
        """one extra untimed step on ONE stream with every kernel bracketed by HIP events"""
        self.engs[0].timing("*"); self.step(1); torch.cuda.synchronize()
        kt = self.engs[0].timing_read(); self.engs[0].timing(None)
        return kt

    def results(self):
        inf = capi.from_device(self.info, capi.INFO_DT, self.B)
        assert (inf["status"] == 0).all(), "strip capacity overflow"
        return int(self.counts.sum().item()), inf

    def close(self):
        for e in self.engs:
            capi._check(capi.lib.lfa_chain_passes(e._h, None))
        for e in self.engs:
            e.close()


def _front_end_bytes(CH, H, W, Cc, code):
    """algorithmic bytes per launch of the whole-image kernels, for `dominant`"""
    npx = H * W
    s = Cc * (2 if code == capi.LFA_U16 else 1)
    # S1+S2 are priced at the image bytes READ; the class plane the histogram kernels also write (N/2 bytes, this
    # implementation's own choice) is not counted
    table = {"k_minmax_rgb": npx * 3, "k_hist_rgb": npx * 3, "k_minmax_u16": npx * 2, "k_hist_u16": npx * 2,
             "k_hist_u8": npx, "k_mask": npx // 2 + npx // 8}
    return lambda k: CH * table.get(k, npx * s)


def bench_u16(B=1024, CH=256, pool=8, real=False, steps=10):
    """configs[2]: 2048x2048 uint16 frames, chunks alternating on two streams / handles like bench.py.
    real=True: the reference's own 12-bit frame (tests/golden/excel2_frame.png, images/excel2.xlsx), rolled per pool slot."""
    H = W = 2048
    if real:
        from PIL import Image
        f = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "excel2_frame.png")))
        pool_np = np.stack([np.ascontiguousarray(np.roll(f, (17 * i, 64 * i), (0, 1))) for i in range(pool)])
    else:
        pool_np = np.stack([synth.synth_u16(H, W, 40, 1234 + i) for i in range(pool)])
    p = torch.from_numpy(pool_np.view(np.uint8).reshape(pool, -1)).to(DEV())
    batch = p[torch.arange(B, device=DEV()) % pool].contiguous()
    pl = _Pipeline(batch, capi.LFA_U16, B, CH, H, W, 1, 1020, 51, 64)
    for _ in range(3):                  # the handles settle their choices (one-pass histogram, class plane, LDS slots)
        pl.step(); torch.cuda.synchronize()
    dt = timed(pl.step, steps)
    kt = pl.kernel_times()
    ns, inf = pl.results()
    out = {"config": f"{B} resident copies of the reference's 12-bit frame excel2 (2048x2048 uint16), rolled per pool slot" if real else f"configs[2]: {B} resident 2048x2048 uint16 frames (pool of {pool} synthetic frames, 40 strips each, replicated)",
           "frames": B, "chunk": CH, "streams": pl.NS, "chained": bool(pl.chain), "pool": pool, "steps": steps, "ms_per_step": round(dt * 1e3, 3),
           "frames_per_s": round(B / dt, 1), "strips_per_s": round(ns / dt, 1), "labels_mean": float(inf["n_labels"].mean()),
           "runs_mean": float(inf["n_runs"].mean()), "algorithmic_bytes_per_step": B * pl.ib,
           "e2e_frac_hbm": round(B / dt * pl.ib / (HBM_PEAK_GBS * 1e9), 4),
           "dominant": dominant(kt, _front_end_bytes(CH, H, W, 1, capi.LFA_U16)), "kernel_ms_one_stream": _kernel_table(kt)}
    pl.close()
    return out


def bench_big(B=128, CH=int(os.environ.get("LFA_BIG_CHUNK", "64")), pool=4, steps=10, seed0=77, keep_pipeline=False, barrier=None):
    """configs[3]: 3072x4096 colour images (the reference's phone-camera size), 2 chunk streams like bench.py."""
    H, W = 3072, 4096
    pool_np = np.stack([synth.synth_rgb(H, W, 8, seed0 + i) for i in range(pool)])
    p = torch.from_numpy(pool_np.reshape(pool, -1)).to(DEV())
    batch = p[torch.arange(B, device=DEV()) % pool].contiguous()
    del p
    mm_c = int(W / 40)
    pl = _Pipeline(batch, capi.LFA_U8, B, CH, H, W, 3, mm_c * 20, mm_c, 16)
    dt = timed(pl.step, steps, barrier)
    kt = pl.kernel_times()
    ns, inf = pl.results()
    out = {"config": f"configs[3]: 3072x4096 RGB uint8 images, synthetic, 8 strips each ({B} resident per GPU from a pool of {pool}, cycled)",
           "images": B, "chunk": CH, "streams": pl.NS, "chained": bool(pl.chain), "pool": pool, "steps": steps, "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1),
           "strips_per_s": round(ns / dt, 1), "strips_per_step": ns, "ambiguous": int(inf["ambiguous"].sum()),
           "algorithmic_bytes_per_step": B * pl.ib, "e2e_frac_hbm": round(B / dt * pl.ib / (HBM_PEAK_GBS * 1e9), 4),
           "dominant": dominant(kt, _front_end_bytes(CH, H, W, 3, capi.LFA_U8)), "kernel_ms_one_stream": _kernel_table(kt)}
    if keep_pipeline:
        out["_pipeline"] = pl            # bench.py gathers the per-strip records of this configuration, then closes it
    else:
        pl.close()
    return out


def bench_photo(B=1024, CH=256, pool=8, steps=10):
    """configs[1]'s shape on photo-like content: the decoded reference sample tests/golden/image1_decoded.npz (512x512 grey,
    what cv2.imread turns into three equal channels), tiled to 1536x2048x3 with a different shift per image.  Many pixels
    share the Otsu threshold's class here, so the handles stop writing the class plane and the mask pass compares the
    image."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "image1_decoded.npz"))["gray"]
    H, W = 1536, 2048
    tile = np.tile(g, (3, 4))
    pool_np = np.stack([np.repeat(np.roll(tile, (37 * i, 91 * i), (0, 1))[:, :, None], 3, axis=2) for i in range(pool)])
    p = torch.from_numpy(pool_np.reshape(pool, -1)).to(DEV())
    batch = p[torch.arange(B, device=DEV()) % pool].contiguous()
    pl = _Pipeline(batch, capi.LFA_U8, B, CH, H, W, 3, 1020, 51, 64)
    pl.step(); torch.cuda.synchronize(); pl.step(); torch.cuda.synchronize()      # the handles settle their class-plane choice
    dt = timed(pl.step, steps)
    kt = pl.kernel_times()
    ns, inf = pl.results()
    out = {"config": "photo-like colour images 1536x2048x3 (reference sample image1, tiled)", "images": B, "steps": steps,
           "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1), "strips_per_s": round(ns / dt, 1),
           "labels_mean": float(inf["n_labels"].mean()), "algorithmic_bytes_per_step": B * pl.ib,
           "e2e_frac_hbm": round(B / dt * pl.ib / (HBM_PEAK_GBS * 1e9), 4),
           "dominant": dominant(kt, _front_end_bytes(CH, H, W, 3, capi.LFA_U8)), "kernel_ms_one_stream": _kernel_table(kt)}
    pl.close()
    return out


def bench_rois(B=64, per=256, steps=10, CH=256):
    """configs[4]: dense select-region workload, 256 ROIs per image (h ~ U{64..512}, w ~ U{16..64}), profile + peaks.
    B = 4096 is BASELINE.json's size (4096 images x 256 ROIs, all resident: 38.7 GB of images, 1 048 576 ROIs); the images
    go through lfa_profile_batch in calls of CH images."""
    H, W = 1536, 2048
    CH = min(CH, B)
    pool_np = np.stack([synth.synth_rgb(H, W, 8, 1234 + i) for i in range(8)])
    p = torch.from_numpy(pool_np.reshape(8, -1)).to(DEV())
    batch = p[torch.arange(B, device=DEV()) % 8].contiguous()
    del p
    rois = synth.synth_rois(B, H, W, per).astype(capi.ROI_DT)
    n = len(rois)
    stride = 512
    ib = H * W * 3
    chunks = []                                           # (first image, images, device ROI array rebased to the chunk, first ROI, ROIs)
    for b0 in range(0, B, CH):
        sel = np.flatnonzero((rois["image"] >= b0) & (rois["image"] < b0 + CH))
        assert len(sel) and np.array_equal(sel, np.arange(sel[0], sel[0] + len(sel)))
        r = rois[sel].copy()
        r["image"] -= b0
        chunks.append((b0, min(CH, B - b0), capi.to_device(r, DEV()), int(sel[0]), len(sel)))
    eng = capi.Engine(device=DEV().index, max_h=H, max_w=W, max_c=3, max_batch=CH)
    prof = torch.empty(n * stride, dtype=torch.float64, device=DEV()); peaks = capi.dev_bytes(n * 40, DEV())
    st = capi.current_stream_ptr(DEV())

    def step():
        for b0, nb, rt, r0, nr in chunks:
            capi._check(capi.lib.lfa_profile_batch(eng._h, batch.data_ptr() + b0 * ib, capi.LFA_U8, nb, H, W, 3, rt.data_ptr(), nr, 0, 15,
                                                   prof.data_ptr() + r0 * stride * 8, stride, peaks.data_ptr() + r0 * 40, st))
    dt = timed(step, steps)
    eng.timing("*"); step(); torch.cuda.synchronize(); kt = eng.timing_read(); eng.timing(None)
    byts = int(((rois["r1"] - rois["r0"]).astype(np.int64) * (rois["c1"] - rois["c0"]) * 3).sum())
    pk = capi.from_device(peaks, capi.PEAKS_DT, n)
    assert (pk["n_peaks"] >= 1).all()
    out = {"config": f"configs[4]: dense select-region, 256 ROIs per image on {B} resident 1536x2048 RGB images (profile + peaks), {CH} images per call",
           "images": B, "rois": n, "steps": steps, "ms_per_step": round(dt * 1e3, 3), "rois_per_s": round(n / dt, 1),
           "algorithmic_bytes_per_step": byts, "GBps_algorithmic": round(byts / dt / 1e9, 1),
           "e2e_frac_hbm": round(byts / dt / (HBM_PEAK_GBS * 1e9), 4),
           "dominant": dominant(kt, lambda k: byts // len(chunks) if k in ("k_profile", "k_profile_band") else None), "kernel_ms": _kernel_table(kt)}
    eng.close()
    return out


def bench_gauss(B=64, steps=10):
    """S10: Gaussian sigma=1 over batches; bytes = input read once + output written once."""
    outs = []
    for (H, W, Cc, dt_np, f32) in ((1536, 2048, 3, np.uint8, 0), (2048, 2048, 1, np.uint16, 0), (1536, 2048, 3, np.uint8, 1)):
        rng = np.random.default_rng(0)
        img = rng.integers(0, 256, size=(4, H, W, Cc) if Cc == 3 else (4, H, W)).astype(dt_np)
        p = torch.from_numpy(img.view(np.uint8).reshape(4, -1)).to(DEV())
        batch = p[torch.arange(B, device=DEV()) % 4].contiguous()
        eng = capi.Engine(device=DEV().index, max_h=H, max_w=W, max_c=3, max_batch=B)
        out = torch.empty(B * H * W * Cc, dtype=torch.float32 if f32 else torch.float64, device=DEV())
        st = capi.current_stream_ptr(DEV())
        code = capi.np_dtype_code(dt_np)

        def step():
            capi._check(capi.lib.lfa_gaussian_batch(eng._h, batch.data_ptr(), code, B, H, W, Cc, 1.0, None, 0, out.data_ptr(), f32, st))
        dt = timed(step, steps)
        eng.timing("*"); step(); torch.cuda.synchronize(); kt = eng.timing_read(); eng.timing(None)
        byts = B * H * W * Cc * (np.dtype(dt_np).itemsize + (4 if f32 else 8))
        outs.append({"config": f"gaussian sigma=1 {H}x{W}x{Cc} {np.dtype(dt_np).name} -> {'f32' if f32 else 'f64'}", "images": B,
                     "steps": steps, "ms_per_step": round(dt * 1e3, 3), "algorithmic_bytes_per_step": byts,
                     "GBps_algorithmic": round(byts / dt / 1e9, 1), "e2e_frac_hbm": round(byts / dt / (HBM_PEAK_GBS * 1e9), 4),
                     "dominant": dominant(kt, lambda k: byts)})
        eng.close()
        del out, batch
    return outs


def bench_lines(B=16, steps=5, H=1536, W=2048):
    """detect_lines over a batch of 2-D uint8 frames; bytes = the frames read once."""
    pool_np = np.stack([synth.synth_lines(H, W, 100 + i, np.uint8) for i in range(4)])
    p = torch.from_numpy(pool_np.reshape(4, -1)).to(DEV())
    batch = p[torch.arange(B, device=DEV()) % 4].contiguous()
    eng = capi.Engine(device=DEV().index, max_h=H, max_w=W, max_c=1, max_batch=B)
    cap = 256
    boxes = capi.dev_bytes(B * cap * 16, DEV()); counts = torch.zeros(B, dtype=torch.int32, device=DEV())
    info = capi.dev_bytes(B * capi.LINESINFO_DT.itemsize, DEV())
    w9, w3 = eng.scipy_gaussian_taps(1.0, 4), eng.scipy_gaussian_taps(2 / 6.0, 1)
    st = capi.current_stream_ptr(DEV())

    def step():
        capi._check(capi.lib.lfa_detect_lines_batch(eng._h, batch.data_ptr(), capi.LFA_U8, B, H, W, w9.ctypes.data, w3.ctypes.data,
                                                    boxes.data_ptr(), cap, counts.data_ptr(), info.data_ptr(), st))
    dt = timed(step, steps)
    eng.timing("*"); step(); torch.cuda.synchronize(); kt = eng.timing_read(); eng.timing(None)
    inf = capi.from_device(info, capi.LINESINFO_DT, B)
    byts = B * H * W
    out = {"config": f"detect_lines: {B} frames {H}x{W} uint8 (synth_lines)", "frames": B, "steps": steps,
           "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1), "labels_mean": float(inf["n_labels"].mean()),
           "boxes_mean": float(inf["n_boxes"].mean()), "algorithmic_bytes_per_step": byts,
           "e2e_frac_hbm": round(byts / dt / (HBM_PEAK_GBS * 1e9), 5), "kernel_ms": _kernel_table(kt, top=12)}
    eng.close()
    return out


def bench_peaks(n=16384, length=288):
    rng = np.random.default_rng(0)
    x = np.arange(length)
    prof = 180 - 40 * np.exp(-0.5 * ((x - 0.35 * length) / 6) ** 2) - 25 * np.exp(-0.5 * ((x - 0.5 * length) / 6) ** 2)
    buf = np.round((prof[None, :] + rng.normal(0, 2.0, size=(n, length))) * 60) / 60
    bt = capi.to_device(buf, DEV()); lt = capi.to_device(np.full(n, length, np.int32), DEV())
    out = capi.dev_bytes(n * 40, DEV())
    eng = capi.Engine(device=DEV().index, max_h=64, max_w=64, max_c=1, max_batch=1)
    st = capi.current_stream_ptr(DEV())
    res = []
    for two in (0, 1):
        def step():
            capi._check(capi.lib.lfa_peaks_batch(eng._h, bt.data_ptr(), lt.data_ptr(), n, length, 0, 25, two, 15, out.data_ptr(), st))
        dt = timed(step, 5)
        pk = capi.from_device(out, capi.PEAKS_DT, n)
        res.append({"config": f"k_peaks two={two}", "profiles": n, "length": length, "ms": round(dt * 1e3, 3),
                    "profiles_per_s": round(n / dt, 1), "mean_peak_width": float((pk["end"][:, 0] - pk["start"][:, 0]).mean())})
    eng.close()
    return res


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
    return {"config": "host frames -> analyze_stream (pinned double-buffered upload)", "frames": n, "batch": batch,
            "seconds": round(dt, 3), "frames_per_s": round(n / dt, 1), "strips_per_s": round(strips / dt, 1),
            "host_to_device_GBps": round(n * imgs[0].nbytes / dt / 1e9, 2)}


def big_with_gather(steps=10):
    """configs[3] plus the gather of its per-strip records (one rank: the packing and the trivial gather)."""
    shard = importlib.import_module("image_analysis_amd.shard")
    r = bench_big(steps=steps, keep_pipeline=True)
    pl = r.pop("_pipeline")
    cn = pl.counts.cpu().numpy()
    t0 = time.perf_counter()
    rec = shard.pack_strip_records(capi.from_device(pl.strips, capi.STRIP_DT, pl.B * pl.ms).reshape(pl.B, pl.ms),
                                   capi.from_device(pl.peaks, capi.PEAKS_DT, pl.B * pl.ms).reshape(pl.B, pl.ms), cn)
    c3, rec3 = shard.gather_strip_records(cn, rec, pl.B)
    r.update({"n_gpus": 1, "scaling": "weak", "gather": {"records": len(rec3), "record_bytes": rec3.dtype.itemsize,
                                                            "seconds": round(time.perf_counter() - t0, 5), "backend": "none (one rank)"}})
    pl.close()
    return r


def named(name, steps=10):
    """{key in bench.py's `configs` object: result} of one configuration"""
    if name == "big":
        return {"configs[3]": big_with_gather(steps)}
    if name == "u16":
        return {"configs[2]": bench_u16(B=10240, steps=steps)}          # BASELINE.json: the stack replicated to 10k frames, all resident
    if name == "u16real":
        return {"configs[2]_excel2_frame": bench_u16(B=10240, real=True, steps=steps)}
    if name == "rois":
        return {"configs[4]": bench_rois(B=4096, steps=steps)}          # BASELINE.json: 256 ROIs per image x 4096 images
    if name == "gauss":
        g = bench_gauss(steps=steps)
        return {"gaussian_rgb_u8_f64": g[0], "gaussian_u16_f64": g[1], "gaussian_rgb_u8_f32": g[2]}
    if name == "photo":
        return {"photo_like_1536x2048x3": bench_photo(steps=steps)}
    if name == "lines":
        return {"detect_lines_1536x2048_u8": bench_lines(steps=min(steps, 5))}
    raise ValueError(name)


if __name__ == "__main__":
    what = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["u16", "rois", "gauss"]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
    torch.cuda.set_device(0)
    if "--json" in sys.argv:                            # bench.py's child: one JSON object {key: result}
        print(json.dumps(named(what[0], steps)), flush=True)
        sys.exit(0)

    def show(r):
        for d in (r if isinstance(r, list) else [r]):
            print(json.dumps(d), flush=True)
    if "u16" in what: show(bench_u16(CH=int(os.environ.get("LFA_CHUNK", "256"))))
    if "u16real" in what: show(bench_u16(real=True))
    if "big" in what: show(bench_big())
    if "photo" in what: show(bench_photo())
    if "rois" in what: show(bench_rois())
    if "gauss" in what: show(bench_gauss())
    if "peaks" in what: show(bench_peaks())
    if "lines" in what: show(bench_lines())
    if "stream" in what: show(bench_stream())
