"""Files and frame streams in front of the device path.

The reference reads ONE file per click -- `tifffile.imread` for .tif/.tiff, `cv2.imread` otherwise
(LFA-Image_Analysis/detection.py:454) -- and hands the array to detect_lateral_flow_tests (:462).  Batch work needs two
things the reference does not have: a way to walk the pages of a multi-page 16-bit TIFF stack, and a pipeline that keeps
the GPU fed while the host is still decoding and packing the next frames.

* `imread(path)` mirrors detection.py:454 with the decoders this image has (PIL): 2-D uint8/uint16 for grey TIFF/PNG,
  (H,W,3) uint8 in **BGR** order for colour files, and -- like cv2.imread's default IMREAD_COLOR -- three equal channels
  for a grey JPEG/PNG.  Neither tifffile nor cv2 is installed here, and libjpeg builds may differ by one count per
  pixel, so the DECODE step is "parity unpinned"; everything after it is bit-exact (tests/test_frames.py works on
  lossless files for that reason).
* `iter_tiff_pages(path)` yields the pages of a TIFF stack one by one (memory stays at one page).
* `analyze_stream(frames, ...)` packs equally shaped frames into PINNED host batches, uploads batch k+1 on a copy stream
  while batch k is in lfa_analyze_batch on the compute stream, and brings the (small) results back asynchronously.
  Nothing here touches pixels on the host beyond the one copy into the pinned buffer.
"""
import os

import numpy as np

from . import _capi


def _pil():
    from PIL import Image
    return Image


def _page_to_array(im) -> np.ndarray:
    mode = im.mode
    if mode in ("I;16", "I;16L", "I;16B", "I;16N"):
        return np.asarray(im).astype(np.uint16, copy=False)
    if mode == "L":
        return np.asarray(im)
    if mode in ("RGB", "RGBA", "P", "CMYK", "YCbCr", "LA", "1"):
        return np.asarray(im.convert("RGB"))
    if mode in ("I", "F"):
        a = np.asarray(im)
        if a.min() >= 0 and a.max() <= 65535 and np.all(a == np.floor(a)):
            return a.astype(np.uint16)
        raise TypeError(f"pixel values of mode {mode} do not fit uint16")
    raise TypeError(f"unsupported image mode {mode}")


def iter_tiff_pages(path):
    """Pages of a (multi-page) TIFF in order, each a 2-D uint8/uint16 or (H,W,3) uint8 RGB array."""
    Image = _pil()
    with Image.open(path) as im:
        n = getattr(im, "n_frames", 1)
        for k in range(n):
            im.seek(k)
            yield np.ascontiguousarray(_page_to_array(im))


def imread(path) -> np.ndarray:
    """detection.py:454 -- the array detect_lateral_flow_tests is given for `path`."""
    ext = os.path.splitext(path)[1]
    if ext in (".tif", ".tiff"):                              # tifffile.imread: pages stacked when there are several
        pages = list(iter_tiff_pages(path))
        return pages[0] if len(pages) == 1 else np.stack(pages)
    Image = _pil()
    with Image.open(path) as im:                              # cv2.imread: always 3 channels, blue first
        a = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(a[:, :, ::-1])


def analyze_stream(frames, batch=16, device=0, line_length=10, max_strips=64, min_maxr=None, min_maxc=None, pack_threads=16):
    """detect + image_analysis(strip, False) for every frame of an iterable of equally shaped uint8/uint16 frames.

    Yields, in input order, one `(frame_index, strips, profiles, peaks)` per frame: `strips` a STRIP_DT array, `profiles`
    a list of float64 arrays, `peaks` a PEAKS_DT array.  min_maxr / min_maxc default to the GUI's rule
    (detection.py:457-460: width/40 and 20 times that).  Double-buffered: while the device works on one batch the host
    packs and uploads the next."""
    torch = _capi._torch()
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the LFA path runs on the GPU only (no CPU fallback)")
    it = iter(frames)
    try:
        first = np.asarray(next(it))
    except StopIteration:
        return
    if first.ndim not in (2, 3) or (first.ndim == 3 and first.shape[-1] != 3):
        raise ValueError("frames must be (H,W) or (H,W,3)")
    code = _capi.np_dtype_code(first.dtype)
    H, W = first.shape[:2]
    Cc = 1 if first.ndim == 2 else 3
    mm_c = int(W / 40) if min_maxc is None else int(min_maxc)
    mm_r = int(mm_c * 20) if min_maxr is None else int(min_maxr)
    stride = max(1, int(H * .6) + 1)
    dev = torch.device("cuda", device)
    frame_bytes = first.nbytes
    eng = _capi.Engine(device=device, max_h=H, max_w=W, max_c=Cc, max_batch=batch)
    copy_s, comp_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    slots = []
    for _ in range(2):
        slots.append({
            "host": torch.empty((batch, frame_bytes), dtype=torch.uint8, pin_memory=True),
            "dev": torch.empty((batch, frame_bytes), dtype=torch.uint8, device=dev),
            "uploaded": torch.cuda.Event(), "done": torch.cuda.Event(), "busy": False,
        })

    def launch(slot, n, index0):
        with torch.cuda.stream(copy_s):
            slot["dev"][:n].copy_(slot["host"][:n], non_blocking=True)
            slot["uploaded"].record(copy_s)
        with torch.cuda.stream(comp_s):
            comp_s.wait_event(slot["uploaded"])
            out = eng.analyze_raw(slot["dev"], code, n, H, W, Cc, mm_r, mm_c, int(line_length), max_strips, stride)
            host = [t.to("cpu", non_blocking=True) for t in out]         # pinned by torch's caching host allocator
            slot["done"].record(comp_s)
        slot.update(n=n, index0=index0, out=out, host_out=host, busy=True)

    def collect(slot):
        slot["done"].synchronize()
        n, index0 = slot["n"], slot["index0"]
        strips, counts, info, _rois, profiles, peaks = [t.numpy() for t in slot["host_out"]]
        per_image, _inf = _capi.Engine.unpack_analysis(n, max_strips, stride, strips, counts, info, profiles, peaks)
        if per_image is None:
            # an image with more than max_strips accepted regions: this batch goes round again with more slots (the frames
            # are still in the slot's device buffer), like Engine.detect / Engine.analyze
            with torch.cuda.stream(comp_s):
                per_image, _inf = eng.analyze_device(slot["dev"], code, n, H, W, Cc, mm_r, mm_c, int(line_length),
                                                     min(1024, max_strips * 4), stride)
        elif len(eng._ambiguous(_inf)):
            # a pixel within 2^-40 of a histogram edge: those frames again with the host's pow() (Engine.detect)
            with torch.cuda.stream(comp_s):
                per_image, _inf = eng.resolve_ambiguous(slot["dev"], code, n, H, W, Cc, mm_r, mm_c, int(line_length), max_strips,
                                                        stride, per_image, _inf.copy())
        res = [(index0 + i, *per_image[i]) for i in range(n)]
        slot["busy"] = False
        slot["out"] = slot["host_out"] = None
        return res

    from concurrent.futures import ThreadPoolExecutor
    packers = ThreadPoolExecutor(max_workers=max(1, int(pack_threads)))

    def pack(slot, group, index0):
        """copy the frames of one batch into the slot's pinned buffer (numpy releases the GIL: the copies run in parallel)"""
        def one(q):
            a = np.asarray(group[q])
            if a.shape != first.shape or a.dtype != first.dtype:
                raise ValueError(f"frame {index0 + q}: shape/dtype {a.shape}/{a.dtype} differs from the first frame's")
            slot["host"][q].numpy()[:] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        list(packers.map(one, range(len(group))))

    try:
        cur, index0, nxt = 0, 0, first
        pending = []                                               # slots in flight, oldest first
        while nxt is not None:
            group = []
            while nxt is not None and len(group) < batch:
                group.append(nxt)
                nxt = next(it, None)
            slot = slots[cur]
            if slot["busy"]:                                       # its previous batch must be out before the buffer is reused
                pending.remove(slot)
                yield from collect(slot)
            pack(slot, group, index0)
            launch(slot, len(group), index0)
            pending.append(slot)
            index0 += len(group)
            cur ^= 1
        for slot in pending:
            yield from collect(slot)
    finally:
        packers.shutdown(wait=True)
        torch.cuda.synchronize(dev)
        eng.close()
