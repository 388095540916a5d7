"""Per-kernel times of one lfa_analyze_batch call on one stream (headline shapes), results NOT checked: for ablation builds
(tools/build_variant.py) whose outputs are wrong on purpose.  usage: [LFA_LIB_PATH=...] python tools/kernel_time.py [debug_flags]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("image_analysis_amd._capi")
synth = importlib.import_module("image_analysis_amd.synth")
H, W, B = 1536, 2048, 256
dev = torch.device("cuda", 0)
pool = torch.from_numpy(np.stack([synth.synth_rgb(H, W, 8, 1234 + i) for i in range(16)]).reshape(16, -1)).to(dev)
batch = pool[torch.arange(B, device=dev) % int(os.environ.get('LFA_KT_POOL', '16'))].contiguous()
e = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=B)
if len(sys.argv) > 1:
    e.set_debug_flags(int(sys.argv[1]))
ms, stride = 16, int(H * .6) + 1
strips = capi.dev_bytes(B * ms * capi.STRIP_DT.itemsize, dev)
counts = torch.zeros(B, dtype=torch.int32, device=dev)
info = capi.dev_bytes(B * capi.INFO_DT.itemsize, dev)
rois = capi.dev_bytes(B * ms * capi.ROI_DT.itemsize, dev)
prof = torch.empty(B * ms * stride, dtype=torch.float64, device=dev)
peaks = capi.dev_bytes(B * ms * capi.PEAKS_DT.itemsize, dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call():
    capi._check(capi.lib.lfa_analyze_batch(e._h, batch.data_ptr(), capi.LFA_U8, B, H, W, 3, 1020, 51, 10, strips.data_ptr(), ms,
                                           counts.data_ptr(), info.data_ptr(), rois.data_ptr(), prof.data_ptr(), stride, peaks.data_ptr(), st))


for _ in range(4):
    call()
torch.cuda.synchronize()
e.timing("*")
for _ in range(10):
    call()
torch.cuda.synchronize()
t = e.timing_read()
print(os.environ.get("LFA_LIB_PATH", "default"), sys.argv[1:], "pred" if e.debug_pred_path(0) else "",  " ".join(f"{k}={v[0] / v[1]:.4f}" for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])[:5]), flush=True)
