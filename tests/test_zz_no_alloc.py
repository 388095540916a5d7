"""lfa_analyze_batch_static -- the entry point that allocates nothing (its per-call workspace is a file-scope device array
inside liblfa_hip.so) -- against lfa_analyze_batch, through the C ABI, bit for bit: two calls in a row on one GPU, the
second with a smaller input, so that whatever the first call left in the shared array must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(capi, fn, h, dev, img, mm_r, mm_c, max_strips=16):
    import torch
    n, H, W = img.shape[:3]
    Cc = 1 if img.ndim == 3 else 3
    stride = int(H * .6) + 1
    t = capi.to_device(img, dev)
    strips = torch.zeros(n * max_strips * capi.STRIP_DT.itemsize, dtype=torch.uint8, device=dev)
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    info = torch.zeros(n * capi.INFO_DT.itemsize, dtype=torch.uint8, device=dev)
    rois = torch.zeros(n * max_strips * capi.ROI_DT.itemsize, dtype=torch.uint8, device=dev)
    profiles = torch.zeros(n * max_strips * stride, dtype=torch.float64, device=dev)
    peaks = torch.zeros(n * max_strips * capi.PEAKS_DT.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    before = capi.lib.lfa_workspace_bytes(h)
    mem_before = torch.cuda.memory_allocated(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]                # hipMemGetInfo: sees a raw hipMalloc inside the library too
    capi._check(fn(h, t.data_ptr(), capi.np_dtype_code(img.dtype), n, H, W, Cc, mm_r, mm_c, 10, strips.data_ptr(), max_strips,
                   counts.data_ptr(), info.data_ptr(), rois.data_ptr(), profiles.data_ptr(), stride, peaks.data_ptr(),
                   capi.current_stream_ptr(dev)))
    torch.cuda.synchronize()
    assert capi.lib.lfa_workspace_bytes(h) == before and torch.cuda.memory_allocated(dev) == mem_before
    if fn is capi.lib.lfa_analyze_batch_static:
        assert torch.cuda.mem_get_info(dev)[0] == free_before
    cn = counts.cpu().numpy()
    st = capi.from_device(strips, capi.STRIP_DT, n * max_strips).reshape(n, max_strips)
    ro = capi.from_device(rois, capi.ROI_DT, n * max_strips).reshape(n, max_strips)
    pk = capi.from_device(peaks, capi.PEAKS_DT, n * max_strips).reshape(n, max_strips)
    pr = profiles.cpu().numpy().reshape(n, max_strips, stride)
    inf = capi.from_device(info, capi.INFO_DT, n)
    return cn, st, ro, pk, pr, inf


def _same(a, b):
    cn, st, ro, pk, pr, inf = a
    cn2, st2, ro2, pk2, pr2, inf2 = b
    assert np.array_equal(cn, cn2) and cn.sum() > 0
    for i in range(len(cn)):
        k = cn[i]
        assert st[i, :k].tobytes() == st2[i, :k].tobytes()
        assert ro[i, :k].tobytes() == ro2[i, :k].tobytes()
        assert pk[i, :k].tobytes() == pk2[i, :k].tobytes()
        for q in range(k):
            L = pk[i, q]["length"]
            assert L > 0 and pr[i, q, :L].tobytes() == pr2[i, q, :L].tobytes()
    for f in inf.dtype.names:
        if f in ("vmin", "vmax", "v_thr_f", "g_thr", "gmin", "gmax"):
            assert inf[f].tobytes() == inf2[f].tobytes(), f
        else:
            assert np.array_equal(inf[f], inf2[f]), f


def test_static_entry_point_matches_analyze_batch_twice_in_a_row():
    import torch
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb, synth_u16
    dev = torch.device("cuda", 0)
    eng = capi.Engine(device=0, max_h=1024, max_w=1024, max_c=3, max_batch=4)
    try:
        h = eng._h
        assert capi.lib.lfa_static_workspace_bytes() == 64 << 20
        big = np.stack([synth_rgb(768, 1024, 8, 40 + i) for i in range(3)])
        small = np.stack([synth_rgb(320, 384, 8, 3)])
        frames = np.stack([synth_u16(512, 512, 10, 7), synth_u16(512, 512, 10, 8)])
        for img in (big, small, frames, small[:, :200, :160].copy()):                   # every call smaller than / unlike the one before
            mm_c = int(img.shape[2] / 40)
            mm_r = mm_c * 20
            want = _call(capi, capi.lib.lfa_analyze_batch, h, dev, img, mm_r, mm_c)
            got = _call(capi, capi.lib.lfa_analyze_batch_static, h, dev, img, mm_r, mm_c)
            _same(want, got)
            got2 = _call(capi, capi.lib.lfa_analyze_batch_static, h, dev, img, mm_r, mm_c)      # and once more on its own leftovers
            _same(want, got2)
        # a batch beyond the static workspace is refused with a message, not run
        huge = torch.zeros(4 * 1024 * 1024 * 3, dtype=torch.uint8, device=dev)
        rc = capi.lib.lfa_analyze_batch_static(h, huge.data_ptr(), capi.LFA_U8, 4, 1024, 1024, 3, 500, 25, 1, huge.data_ptr(), 16,
                                               huge.data_ptr(), None, None, huge.data_ptr(), 700, huge.data_ptr(), None)
        assert rc == -1 and b"static workspace" in capi.lib.lfa_last_error()
    finally:
        eng.close()


def test_static_entry_point_under_hip_graph_capture():
    """include/lfa.h offers lfa_analyze_batch_static to code under HIP graph capture: one call is captured (global capture
    mode: a hipMalloc, a host wait or an event from outside the capture would fail it), the graph is replayed on the
    captured input and, after the input buffer's CONTENT is replaced, on another batch; both replays must equal
    lfa_analyze_batch bit for bit.  An ordinary call is made on the handle right before the capture, so its counter copy is
    still pending when the captured call is set up (the wait for it is what the static path must not do)."""
    import torch
    import image_analysis_amd._capi as capi
    from image_analysis_amd.synth import synth_rgb
    dev = torch.device("cuda", 0)
    n, H, W, ms = 2, 384, 512, 16
    stride = int(H * .6) + 1
    mm_c = int(W / 40); mm_r = mm_c * 20
    a = np.stack([synth_rgb(H, W, 6, 60 + i) for i in range(n)])
    b = np.stack([synth_rgb(H, W, 5, 80 + i) for i in range(n)])
    eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=n)
    try:
        h = eng._h
        want = [_call(capi, capi.lib.lfa_analyze_batch, h, dev, img, mm_r, mm_c, ms) for img in (a, b)]
        t = capi.to_device(a, dev)
        strips = torch.zeros(n * ms * capi.STRIP_DT.itemsize, dtype=torch.uint8, device=dev)
        counts = torch.zeros(n, dtype=torch.int32, device=dev)
        info = torch.zeros(n * capi.INFO_DT.itemsize, dtype=torch.uint8, device=dev)
        rois = torch.zeros(n * ms * capi.ROI_DT.itemsize, dtype=torch.uint8, device=dev)
        profiles = torch.zeros(n * ms * stride, dtype=torch.float64, device=dev)
        peaks = torch.zeros(n * ms * capi.PEAKS_DT.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="global"):
            rc = capi.lib.lfa_analyze_batch_static(h, t.data_ptr(), capi.LFA_U8, n, H, W, 3, mm_r, mm_c, 10, strips.data_ptr(), ms,
                                                   counts.data_ptr(), info.data_ptr(), rois.data_ptr(), profiles.data_ptr(), stride,
                                                   peaks.data_ptr(), capi.current_stream_ptr(dev))
        assert rc == 0, capi.lib.lfa_last_error()
        torch.cuda.synchronize()
        assert int(counts.sum().item()) == 0                        # captured, not run

        def fetch():
            torch.cuda.synchronize()
            return (counts.cpu().numpy(), capi.from_device(strips, capi.STRIP_DT, n * ms).reshape(n, ms),
                    capi.from_device(rois, capi.ROI_DT, n * ms).reshape(n, ms), capi.from_device(peaks, capi.PEAKS_DT, n * ms).reshape(n, ms),
                    profiles.cpu().numpy().reshape(n, ms, stride), capi.from_device(info, capi.INFO_DT, n))
        g.replay()
        _same(want[0], fetch())
        t.copy_(capi.to_device(b, dev))
        g.replay()
        _same(want[1], fetch())
        g.replay()
        _same(want[1], fetch())
    finally:
        eng.close()
