"""dev aid: phase clocks of k_peaks (variant build -DLFA_PEAKS_CLOCK): LFA_LIB_PATH=image-analysis_amd/variants/liblfa_clock.so python tools/r3_peaks_clock.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("image_analysis_amd._capi")
synth = importlib.import_module("image_analysis_amd.synth")
H, W, B = 1536, 2048, 64
dev = torch.device("cuda", 0)
pool = torch.from_numpy(np.stack([synth.synth_rgb(H, W, 8, 1234 + i) for i in range(8)]).reshape(8, -1)).to(dev)
batch = pool[torch.arange(B, device=dev) % 8].contiguous()
rois = synth.synth_rois(B, H, W, 256).astype(capi.ROI_DT)
rt = capi.to_device(rois, dev)
n = len(rois); stride = 512
eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=B)
prof = torch.empty(n * stride, dtype=torch.float64, device=dev); peaks = capi.dev_bytes(n * 40, dev)
st = capi.current_stream_ptr(dev)
for _ in range(3):
    capi._check(capi.lib.lfa_profile_batch(eng._h, batch.data_ptr(), capi.LFA_U8, B, H, W, 3, rt.data_ptr(), n, 0, 15, prof.data_ptr(), stride, peaks.data_ptr(), st))
torch.cuda.synchronize()
import ctypes
ck = (ctypes.c_ulonglong * 16)()
if hasattr(capi.lib, "lfa_debug_clocks"):
    capi.lib.lfa_debug_clocks(ck)
    capi._check(capi.lib.lfa_profile_batch(eng._h, batch.data_ptr(), capi.LFA_U8, B, H, W, 3, rt.data_ptr(), n, 0, 15, prof.data_ptr(), stride, peaks.data_ptr(), st))
    torch.cuda.synchronize()
    capi.lib.lfa_debug_clocks(ck)
    names = ["thr sum", "walks", "seg sum", "base sum", "argext full", "analyze #1 (incl. 0-3)", "argext left+right", "analyze #2 (incl. 0-3)"]
    waves = n / 2
    for k, nm in enumerate(names):
        print(f"  phase {k} {nm:28s} {ck[k] / waves:9.0f} clocks per wave")
pk = capi.from_device(peaks, capi.PEAKS_DT, n)
a0 = pk["area"][:, 0]; a1 = pk["area"][:, 1]
setup = np.floor(a0); stage = (a0 - setup) * 1e6
print("profiles", n, "two-peak share", float((pk["n_peaks"] == 2).mean()))
for name, v in (("setup(roi load)", setup), ("stage(global->LDS+sync)", stage), ("search", a1)):
    print(f"{name:26s} mean {v.mean():9.0f}  p50 {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f} clocks")
ln = pk["length"]
for lo, hi in ((64, 128), (128, 256), (256, 384), (384, 513)):
    m = (ln >= lo) & (ln < hi)
    print(f"len [{lo},{hi}): search mean {a1[m].mean():9.0f} width mean {(pk['end'][m, 0] - pk['start'][m, 0]).mean():6.1f}")
