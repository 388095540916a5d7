import sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
import image_analysis_amd._capi as capi
dev = torch.device('cuda', 0)
def run(n, length):
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
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        print(n, length, 'two', two, round((time.perf_counter() - t0) / 20 * 1e6, 1), 'us')
    eng.close()
run(2048, 537)
run(16384, 288)
