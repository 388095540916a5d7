"""dev aid: which way the masks of a small batch are decided, call after call (predicted-threshold mode): python tools/dbg_pred.py [photo]"""
import numpy as np, sys, os
sys.path.insert(0, ".")
import image_analysis_amd._capi as capi
from image_analysis_amd.synth import synth_rgb
H, W = 1536, 2048
if len(sys.argv) > 1 and sys.argv[1] == "photo":
    g = np.load(os.path.join("tests", "golden", "image1_decoded.npz"))["gray"]
    tile = np.tile(g, (3, 4))
    scene = np.stack([np.repeat(np.roll(tile, (37 * i, 91 * i), (0, 1))[:, :, None], 3, axis=2) for i in range(8)])
else:
    scene = np.stack([synth_rgb(H, W, 8, 1234 + k) for k in range(8)])
e = capi.Engine(device=0, max_h=H, max_w=W, max_batch=8)
for it in range(14):
    s, info = e.detect(scene, 1020, 51)
    thr = (info["g_thr"] * 2.55e6)
    print(it, "pred", [int(e.debug_pred_path(i)) for i in range(8)], "grey", [int(e.debug_grey_path(i)) for i in range(8)], "G_thr", [round(float(t), 1) for t in thr[:3]], "amb", int(info["ambiguous"].sum()), flush=True)
