"""CPU baseline runner for bench.py (TEST / MEASUREMENT INFRASTRUCTURE ONLY, like the rest of oracle/).

Times oracle.lfa_oracle.analyze_image (the numpy/scipy restatement of detection.py:31-94, 205-316) on a bounded sample
of the bench workload, on one core and on a process pool, in a process of its own that never touches the GPU.  bench.py
starts it as a child before it initialises the device and reads the one JSON line it prints.

    python oracle/cpu_baseline.py --height 1536 --width 2048 --pool 16 --seed0 1234 --single 64 --parallel 512 --procs 16
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_IMAGES = None
_MM = None


def _init(h, w, pool, seed0):
    global _IMAGES, _MM
    from image_analysis_amd.synth import synth_rgb
    _IMAGES = [synth_rgb(h, w, 8, seed0 + i) for i in range(pool)]
    mm_c = int(w / 40)
    _MM = (int(mm_c * 20), mm_c)


def _one(i):
    from oracle import lfa_oracle as O
    return len(O.analyze_image(_IMAGES[i % len(_IMAGES)], _MM[0], _MM[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1536)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--pool", type=int, default=16)
    ap.add_argument("--seed0", type=int, default=1234)
    ap.add_argument("--single", type=int, default=64, help="images timed on one core")
    ap.add_argument("--parallel", type=int, default=512, help="images timed on the process pool (0 = skip)")
    ap.add_argument("--procs", type=int, default=16)
    a = ap.parse_args()
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"                                  # one core per process: the pool supplies the parallelism
    _init(a.height, a.width, a.pool, a.seed0)
    _one(0)                                                  # imports, page-in
    t0 = time.perf_counter()
    s1 = sum(_one(i) for i in range(a.single))
    t1 = time.perf_counter() - t0
    out = {"single": {"images": a.single, "strips": s1, "seconds": t1}}
    if a.parallel > 0 and a.procs > 1:
        with mp.get_context("fork").Pool(a.procs, initializer=_init, initargs=(a.height, a.width, a.pool, a.seed0)) as p:
            p.map(_one, range(a.procs * 2))                  # workers up, images generated
            t0 = time.perf_counter()
            sp = sum(p.map(_one, range(a.parallel), chunksize=4))
            tp = time.perf_counter() - t0
        out["parallel"] = {"images": a.parallel, "strips": sp, "seconds": tp, "procs": a.procs}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
