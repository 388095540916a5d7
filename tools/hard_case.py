#!/usr/bin/env python3
"""How does the colour front end behave on a photo-like image, where many pixels share the threshold's class and the
class plane cannot decide them?  Input: the decoded reference sample tests/golden/image1_decoded.npz (512x512 grey, what
cv2.imread turns into three equal channels), tiled to 1536x2048x3 with a different shift per image.
Prints per-kernel times with the class plane (default) and without it (debug flag 2)."""
import importlib, json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("image_analysis_amd._capi")


dev = torch.device("cuda", 0)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "image1_decoded.npz"))["gray"]
H, W, B = 1536, 2048, 256
tile = np.tile(g, (3, 4))
pool = np.stack([np.repeat(np.roll(tile, (37 * i, 91 * i), (0, 1))[:, :, None], 3, axis=2) for i in range(8)])
p = torch.from_numpy(pool.reshape(8, -1)).to(dev)
batch = p[torch.arange(B, device=dev) % 8].contiguous()
ms, stride = 64, int(H * .6) + 1
eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=B)
strips = capi.dev_bytes(B * ms * 24, dev); counts = torch.zeros(B, dtype=torch.int32, device=dev)
info = capi.dev_bytes(B * 104, dev); rois = capi.dev_bytes(B * ms * 20, dev)
prof = torch.empty(B * ms * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(B * ms * 40, dev)
st = capi.current_stream_ptr(dev)

def step():
    capi._check(capi.lib.lfa_analyze_batch(eng._h, batch.data_ptr(), capi.LFA_U8, B, H, W, 3, 1020, 51, 10, strips.data_ptr(), ms,
                counts.data_ptr(), info.data_ptr(), rois.data_ptr(), prof.data_ptr(), stride, peaks.data_ptr(), st))

res = {}
for flags in (0, 2):
    capi._check(capi.lib.lfa_set_debug_flags(eng._h, flags))
    step(); torch.cuda.synchronize()
    eng.timing("*"); step(); torch.cuda.synchronize(); kt = eng.timing_read(); eng.timing(None)
    inf = capi.from_device(info, capi.INFO_DT, B)
    res[flags] = (counts.cpu().numpy().copy(), capi.from_device(strips, capi.STRIP_DT, B * ms).copy())
    print(json.dumps({"debug_flags": flags, "status_ok": bool((inf["status"] == 0).all()), "strips": int(counts.sum().item()),
                      "labels_mean": float(inf["n_labels"].mean()),
                      "kernel_ms": {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:6]}}))
assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1]), "plane and image paths disagree"
print("plane and image paths agree")
