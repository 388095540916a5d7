"""dev aid: per-kernel times and the way each frame of a uint16 batch is decided, call after call: python tools/dbg_pred_u16.py"""
import numpy as np, sys, torch
sys.path.insert(0, ".")
import image_analysis_amd._capi as capi
from image_analysis_amd.synth import synth_u16
H = W = 2048
pool = np.stack([synth_u16(H, W, 40, 1234 + k) for k in range(8)])
fr = np.ascontiguousarray(pool[np.arange(64) % 8])
e = capi.Engine(device=0, max_h=H, max_w=W, max_c=1, max_batch=64)
dev = torch.device("cuda", 0)
d = capi.to_device(fr, dev)
for it in range(6):
    e.timing("*")
    out = e.analyze_raw(d, capi.LFA_U16, 64, H, W, 1, 1020, 51, 10, 64, int(H * .6) + 1)
    torch.cuda.synchronize()
    kt = e.timing_read()
    print(it, [int(e.debug_pred_path(i)) for i in range(8)], [int(e.debug_used_plane(i)) for i in range(8)], {k: round(v[0], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:5]}, flush=True)
