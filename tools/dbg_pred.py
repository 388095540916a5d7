import numpy as np, sys
sys.path.insert(0, ".")
import image_analysis_amd._capi as capi
from image_analysis_amd.synth import synth_rgb
H, W = 1536, 2048
scene = np.stack([synth_rgb(H, W, 8, 1234 + k) for k in range(16)])
e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=16)
for it in range(4):
    s, info = e.detect(scene, 1020, 51)
    print(it, [int(e.debug_pred_path(i)) for i in range(16)], flush=True)
thr = (info["g_thr"] * 2.55e6).astype(np.int64)
print(thr, info["otsu_bin"])
lo, hi = thr.min() - (thr.max() - thr.min()) // 2 - 16384, thr.max() + (thr.max() - thr.min()) // 2 + 16384
lo &= ~1023; hi = (hi + 1023) & ~1023
print(lo, hi)
for i in range(4):
    d = scene[i].astype(np.int64)
    G = 2125 * d[..., 0] + 7154 * d[..., 1] + 721 * d[..., 2]
    inside = (G >= lo) & (G < hi)
    print(i, "pixels inside", int(inside.sum()), "groups", int(inside.reshape(H, W // 16, 16).any(axis=2).sum()), "of", H * W // 16, "cap", H * W // 256)
