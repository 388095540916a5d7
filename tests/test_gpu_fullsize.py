"""GPU parity at BASELINE.json's full image sizes (oracle still finishes in about a second per image) and
size-independent properties on a larger batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lfa_oracle as O


def _strips(st):
    return [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in st]


def _check_full(imgs, max_batch):
    import image_analysis_amd._capi as capi
    H, W = imgs.shape[1:3]
    mm_c = int(W / 40)
    mm_r = mm_c * 20
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=max_batch)
    res, info = eng.analyze(imgs, mm_r, mm_c)
    assert (info["ambiguous"] == 0).all()
    for i, (strips, profiles, peaks) in enumerate(res):
        d = O.detect_debug(imgs[i], mm_r, mm_c)
        assert capi.host_threshold(info[i], imgs.ndim == 4) == d["threshold"]
        assert info[i]["n_labels"] == d["n_labels"]
        assert _strips(strips) == d["strips"]
        for k, (a, b, c, e) in enumerate(d["strips"]):
            wp, wk = O.image_analysis(imgs[i][a:b, c:e], False)
            assert np.array_equal(profiles[k], wp)
            assert [(int(peaks[k]["start"][q]), int(peaks[k]["end"][q]), float(peaks[k]["area"][q]))
                    for q in range(int(peaks[k]["n_peaks"]))] == [(s0, e0, float(a0)) for s0, e0, a0 in wk]
    eng.close()
    return res, info


def test_config2_size_colour_1536x2048():
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(1536, 2048, 8, 1234 + i) for i in range(3)])
    res, info = _check_full(imgs, 3)
    assert all(len(r[0]) == 8 for r in res)


def test_config3_size_uint16_2048x2048():
    from image_analysis_amd.synth import synth_u16
    imgs = np.stack([synth_u16(2048, 2048, 40, 1234 + i) for i in range(2)])
    res, info = _check_full(imgs, 2)
    assert all(len(r[0]) >= 30 for r in res) and info["n_labels"].min() > 300


def test_config4_size_colour_3072x4096():
    from image_analysis_amd.synth import synth_rgb
    imgs = synth_rgb(3072, 4096, 8, 77)[None]
    res, info = _check_full(imgs, 1)
    assert len(res[0][0]) == 8


@pytest.mark.parametrize("kind", ["colour", "uint8", "uint16"])
def test_images_above_2_24_pixels(kind):
    """More than 2^24 pixels: skimage's float32 histogram weights (thresholding.py:309) are no longer exact and round in
    bin order; the colour path restates that order anyway, the integer path switches to the sequential float32 walk
    (lfa_algo.h otsu_seq_*, pinned on the host against the oracle with far larger counts)."""
    from image_analysis_amd.synth import synth_rgb, synth_u16
    H, W = 4104, 4112                                      # 16.88 M pixels
    if kind == "colour":
        img = synth_rgb(H, W, 8, 7)
    elif kind == "uint8":
        img = np.ascontiguousarray(synth_rgb(H, W, 8, 8)[:, :, 1])
    else:
        img = synth_u16(H, W, 40, 9)
    assert img.shape[0] * img.shape[1] > (1 << 24)
    res, info = _check_full(img[None], 1)
    assert len(res[0][0]) >= 8


def test_batch_properties_and_determinism():
    """A tiled batch must give every copy of an image the same answer, twice in a row (atomics / union-find order
    must not leak into results), and flipping an image left-right must mirror the strip columns."""
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    base = np.stack([synth_rgb(768, 1024, 6, 500 + i) for i in range(4)])
    imgs = np.concatenate([base] * 6)                        # 24 images
    mm_c = int(1024 / 40)
    mm_r = mm_c * 20
    eng = capi.Engine(device=0, max_h=768, max_w=1024, max_batch=24)
    r1, i1 = eng.analyze(imgs, mm_r, mm_c)
    r2, i2 = eng.analyze(imgs, mm_r, mm_c)
    for k in range(24):
        a, b, c = r1[k], r1[k % 4], r2[k]
        for x in (b, c):
            assert np.array_equal(a[0], x[0]) and np.array_equal(a[2], x[2])
            assert all(np.array_equal(p, q) for p, q in zip(a[1], x[1]))
    assert np.array_equal(i1["n_labels"], i2["n_labels"])
    flipped = np.ascontiguousarray(base[:, :, ::-1])
    rf, _ = eng.analyze(flipped, mm_r, mm_c)
    for k in range(4):
        want = O.detect_debug(flipped[k], mm_r, mm_c)["strips"]
        assert _strips(rf[k][0]) == want
    eng.close()


@pytest.mark.parametrize("rule,lanes", [("per call", 2), ("standing", 3)])
def test_handles_chained_by_their_histogram_passes(rule, lanes):
    """Chunks taking 2 / 3 handles + streams in turn, each chunk's passes chained to the previous chunk's histogram pass -- by a
    wait enqueued before every call (lfa_stream_wait_passes) or as a standing rule between the handles (lfa_chain_passes, the
    bench's schedule): same strips, profiles and peaks as one handle processing the chunks one after the other."""
    import ctypes as C
    import torch
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    H, W, n, ch = 384, 512, 12, 3
    imgs = np.stack([synth_rgb(H, W, 6, 900 + i) for i in range(n)])
    mm_c = int(W / 40); mm_r = mm_c * 20
    ref_eng = capi.Engine(device=0, max_h=H, max_w=W, max_batch=ch)
    want, _ = ref_eng.analyze(imgs, mm_r, mm_c, max_strips=16)
    ref_eng.close()
    dev = torch.device("cuda", 0)
    engs = [capi.Engine(device=0, max_h=H, max_w=W, max_batch=ch) for _ in range(lanes)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    if rule == "standing":
        for i in range(lanes):
            capi._check(capi.lib.lfa_chain_passes(engs[i]._h, engs[i - 1]._h))
        assert capi.lib.lfa_chain_passes(engs[0]._h, engs[0]._h) == -1     # LFA_ERR_ARG: a handle cannot follow itself
    batch = capi.to_device(imgs, dev)
    ms, stride = 16, int(H * .6) + 1
    strips = capi.dev_bytes(n * ms * 24, dev); counts = torch.zeros(n, dtype=torch.int32, device=dev)
    info = capi.dev_bytes(n * 104, dev); rois = capi.dev_bytes(n * ms * 20, dev)
    prof = torch.zeros(n * ms * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(n * ms * 40, dev)
    torch.cuda.synchronize()
    ib = H * W * 3
    turn = 0
    for rep in range(2):
        for b0 in range(0, n, ch):
            ci, turn = turn, turn + 1
            st = C.c_void_p(streams[ci % lanes].cuda_stream)
            if rule == "per call":
                capi._check(capi.lib.lfa_stream_wait_passes(engs[(ci - 1) % lanes]._h, st))
            capi._check(capi.lib.lfa_analyze_batch(engs[ci % lanes]._h, batch.data_ptr() + b0 * ib, capi.LFA_U8, ch, H, W, 3, mm_r, mm_c, 10,
                        strips.data_ptr() + b0 * ms * 24, ms, counts.data_ptr() + b0 * 4, info.data_ptr() + b0 * 104,
                        rois.data_ptr() + b0 * ms * 20, prof.data_ptr() + b0 * ms * stride * 8, stride,
                        peaks.data_ptr() + b0 * ms * 40, st))
    torch.cuda.synchronize()
    cn = counts.cpu().numpy()
    st_np = capi.from_device(strips, capi.STRIP_DT, n * ms).reshape(n, ms)
    pk_np = capi.from_device(peaks, capi.PEAKS_DT, n * ms).reshape(n, ms)
    pr_np = prof.cpu().numpy().reshape(n, ms, stride)
    for i in range(n):
        w_strips, w_profiles, w_peaks = want[i]
        assert cn[i] == len(w_strips) and np.array_equal(st_np[i, :cn[i]], w_strips)
        for k in range(cn[i]):
            assert np.array_equal(pr_np[i, k, :pk_np[i, k]["length"]], w_profiles[k])
            assert pk_np[i, k] == w_peaks[k]
    for e in engs:
        capi._check(capi.lib.lfa_chain_passes(e._h, None))       # the rule ends before a producer goes away
    for e in engs:
        e.close()


def _many_strips_image(seed=0, rows=6, cols=16):
    """rows x cols small bright blocks (14 x 12 px, area 168 >= 100) on a dark ground, far enough apart that no two
    boxes touch: with the module defaults min_maxr = min_maxc = 0 every one of them is an accepted strip."""
    rng = np.random.default_rng(seed)
    img = rng.integers(10, 30, size=(rows * 40 + 8, cols * 32 + 8), dtype=np.uint8)
    for r in range(rows):
        for c in range(cols):
            img[8 + r * 40: 22 + r * 40, 8 + c * 32: 20 + c * 32] = rng.integers(200, 250, size=(14, 12))
    return img


def test_more_than_64_strips_through_the_batch_api(monkeypatch):
    """An image with 96 accepted strips: Engine.analyze / detection.analyze_batch / frames.analyze_stream start with 64 result
    slots per image and re-run the batch with more when an image overflows them, like detect always did."""
    import image_analysis_amd.detection as det
    from image_analysis_amd import frames as F
    img = _many_strips_image()
    few = img.copy()
    few[:, 200:] = 15                                     # a batch-mate with fewer strips
    d = O.detect_debug(img, 0, 0)
    assert len(d["strips"]) == 96
    monkeypatch.setattr(det, "min_maxr", 0)
    monkeypatch.setattr(det, "min_maxc", 0)
    out = det.analyze_batch(np.stack([few, img]), max_batch=2)
    assert len(out[1]) == 96 and 0 < len(out[0]) < 64
    for i, im in enumerate((few, img)):
        want = O.detect_debug(im, 0, 0)["strips"]
        assert len(out[i]) == len(want)
        for (crop, prof, peaks), (a, b, c, e) in zip(out[i], want):
            assert np.array_equal(crop, im[a:b, c:e])
            wp, wk = O.image_analysis(im[a:b, c:e], False)
            assert np.array_equal(prof, wp) and [(p[0], p[1]) for p in peaks] == [(s0, e0) for s0, e0, _ in wk]
    assert len(det.analyze_batch(img[None], max_strips=8)[0]) == 96           # 8 -> 32 -> 128 slots
    got = list(F.analyze_stream(iter([few, img, few]), batch=2, min_maxr=0, min_maxc=0))
    assert [len(g[1]) for g in got] == [len(out[0]), 96, len(out[0])]


def test_bad_rois_are_refused_not_returned_as_garbage():
    import image_analysis_amd._capi as capi
    eng = capi.Engine(device=0, max_h=64, max_w=64, max_batch=1)
    try:
        img = np.zeros((1, 32, 32), np.uint8)
        roi = np.zeros(1, dtype=capi.ROI_DT)
        for r0, r1, c0, c1, im in ((0, 40, 0, 8, 0), (5, 4, 0, 8, 0), (0, 8, 0, 8, 1), (0, 8, -1, 8, 0)):
            roi["r0"], roi["r1"], roi["c0"], roi["c1"], roi["image"] = r0, r1, c0, c1, im
            with pytest.raises(ValueError):
                eng.profiles(img, roi)
        roi["r0"], roi["r1"], roi["c0"], roi["c1"], roi["image"] = 0, 8, 0, 8, 0
        with pytest.raises(ValueError):
            eng.profiles(img, roi, luts=np.arange(256, dtype=np.uint8), roi_lut=np.array([3], np.int32))
    finally:
        eng.close()


def test_sharded_analyze_batch_two_handles_equal_one(monkeypatch):
    """detection.analyze_batch(images, devices=[0, 0]): two contiguous shards, each with its own handle, stream pair and host
    thread (here both on GPU 0 -- the box has one), joined in image order; must equal the single-handle result and the oracle."""
    import image_analysis_amd.detection as det
    from image_analysis_amd.synth import synth_rgb
    imgs = np.stack([synth_rgb(384, 512, 6, 500 + i) for i in range(7)])            # 7 images: shards of 3 and 4
    mm_c = int(512 / 40)
    monkeypatch.setattr(det, "min_maxc", mm_c)
    monkeypatch.setattr(det, "min_maxr", mm_c * 20)
    one = det.analyze_batch(imgs, max_batch=4)
    for devices in ([0, 0], [0, 0, 0], [0]):
        two = det.analyze_batch(imgs, max_batch=2, devices=devices)
        assert len(two) == len(one) == 7
        for a, b in zip(one, two):
            assert len(a) == len(b) > 0
            for (c1, p1, k1), (c2, p2, k2) in zip(a, b):
                assert np.array_equal(c1, c2) and c1.base is not None and np.array_equal(p1, p2) and k1 == k2
    for i in (0, 6):
        want = O.detect_debug(imgs[i], mm_c * 20, mm_c)["strips"]
        assert [c.shape[:2] for c, _p, _k in one[i]] == [(min(b, 384) - a, min(e, 512) - c) for a, b, c, e in want]
    with pytest.raises(ValueError):
        det.analyze_batch(imgs, devices=[])
