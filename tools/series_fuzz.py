#!/usr/bin/env python3
"""Series fuzz on the GPU box (not a pytest test; a short run is in tests/test_gpu_long_sequences.py): batches of frames of ONE scene
that drifts from call to call -- background level, strip brightness, edge softness, noise -- with a frame of another scene mixed in
now and then and a jump to a new scene every few dozen calls, through one handle per pixel type.  That is the input the predicted-
threshold mode (bit plane + list of undecided groups, window from the last call's thresholds, fallback per image, back-off per handle)
lives on: every call's thresholds, masks, labels and strips are checked against the oracle whichever way each image was decided, and
the run reports how often the bits held.  usage: python tools/series_fuzz.py [calls] [seed]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def scene_frame(rng, H, W, kind, p):
    """one frame of scene p: dark noisy ground, a few bright strips with soft edges; kind 0 colour uint8, 1 uint16, 2 colour uint8 with
    three equal channels (the handle's grey mode)"""
    top = 255.0 if kind != 1 else p["top"]
    g = rng.normal(p["bg"] * top, p["noise"] * top, size=(H, W))
    for c0, w in p["strips"]:
        g[p["r0"]:p["r1"], c0:c0 + w] = rng.normal(p["fg"] * top, p["noise"] * top, size=(p["r1"] - p["r0"], w))
    for _ in range(p["soft"]):                                     # soften the strip edges: pixels between ground and strip
        g = (g + np.roll(g, 1, axis=1)) / 2
    g = np.clip(g, 0, top)
    if kind == 1:
        return g.astype(np.uint16)
    if kind == 2:                                                  # three equal channels: what cv2.imread makes of a grey file
        q = np.round(g / p["step"]) * p["step"] if p.get("step", 1) > 1 else g       # (coarse levels: thresholds then repeat from call to call)
        return np.ascontiguousarray(np.repeat(np.clip(q, 0, 255).astype(np.uint8)[:, :, None], 3, axis=2))
    out = np.stack([g + rng.integers(-2, 3, size=g.shape) for _ in range(3)], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def new_scene(rng, H, W, kind):
    n = int(rng.integers(1, 5))
    strips = [(int(rng.integers(4, W - 24)), int(rng.integers(6, 20))) for _ in range(n)]
    r0 = int(rng.integers(0, H // 3))
    return {"bg": float(rng.uniform(0.03, 0.35)), "fg": float(rng.uniform(0.5, 0.95)), "noise": float(rng.uniform(0.002, 0.04)),
            "soft": int(rng.integers(0, 3)), "strips": strips, "r0": r0, "r1": int(rng.integers(r0 + 20, H)),
            "top": float(rng.choice([1023, 4095, 16383, 65535])) if kind == 1 else 255.0, "step": int(rng.choice([1, 1, 8, 16]))}


def drift(rng, p):
    q = dict(p)
    q["bg"] = float(np.clip(p["bg"] * rng.uniform(0.97, 1.03), 0.01, 0.4))
    q["fg"] = float(np.clip(p["fg"] * rng.uniform(0.98, 1.02), 0.45, 0.98))
    q["noise"] = float(np.clip(p["noise"] * rng.uniform(0.9, 1.1), 0.001, 0.05))
    return q


def run(calls, seed):
    import image_analysis_amd._capi as capi
    td = _load(os.path.join(ROOT, "tests", "test_gpu_detect.py"), "td")
    td.ALLOW_AMBIGUOUS = True
    rng = np.random.default_rng(seed)
    t0 = time.time()
    held = tried = frames = 0
    for kind in (0, 1, 2):
        H, W = int(rng.choice([96, 128, 160])), int(rng.choice([128, 192, 256]))
        eng = capi.Engine(device=0, max_h=H, max_w=W, max_c=3, max_batch=4)
        p = new_scene(rng, H, W, kind)
        for call in range(calls):
            if rng.random() < 0.03:
                p = new_scene(rng, H, W, kind)
                if kind == 1 and rng.random() < 0.5:
                    p["top"] = float(rng.choice([1023, 4095]))
            p = drift(rng, p)
            n = int(rng.integers(1, 5))
            batch = []
            for i in range(n):
                q = new_scene(rng, H, W, kind) if rng.random() < 0.08 else p       # a frame of another scene in the batch
                if kind == 1:
                    q = dict(q, top=p["top"])
                batch.append(scene_frame(rng, H, W, kind, q))
            batch = np.ascontiguousarray(np.stack(batch))
            mm_r, mm_c, L = int(rng.integers(0, H)), int(rng.integers(0, W // 4)), int(rng.choice([3, 10, 10, 10]))
            try:
                res = eng.detect(batch, mm_r, mm_c, L)
                for i in range(n):
                    td.check_detect(capi, eng, batch[i], mm_r, mm_c, L, batch_index=i, res=res)
                    if not (np.asarray(res[1]["status"]) & 4).any():          # (else the handle's last call was the host-pow run of a sub-batch)
                        held += int(eng.debug_pred_path(i))
                    frames += 1
            except AssertionError:
                print("MISMATCH kind", kind, "call", call, "seed", seed, flush=True)
                raise
            if call % 100 == 0:
                print(f"kind {kind} call {call} ok, {held} of {frames} frames decided from the bit plane, {time.time() - t0:.0f} s", flush=True)
        eng.close()
    print(f"series fuzz ok: 3 x {calls} calls (seed {seed}), {frames} frames, {held} decided from the predicted-threshold bit plane, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
