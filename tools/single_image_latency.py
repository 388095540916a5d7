#!/usr/bin/env python3
"""Latency of the drop-in calls on ONE host image (the GUI's case: one file per click, detection.py:462, :541-543):
detect_lateral_flow_tests + image_analysis of every strip, host array in, host results out."""
import importlib, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
det = importlib.import_module("image_analysis_amd.detection")
synth = importlib.import_module("image_analysis_amd.synth")

def run(img, name, reps=20):
    det.min_maxc = int(img.shape[1] / 40); det.min_maxr = det.min_maxc * 20
    strips = det.detect_lateral_flow_tests(img)                      # warm-up: handle, workspace
    t0 = time.perf_counter()
    for _ in range(reps):
        strips = det.detect_lateral_flow_tests(img)
    t1 = time.perf_counter()
    for _ in range(reps):
        res = [det.image_analysis(s, False) for s in strips]
    t2 = time.perf_counter()
    print(json.dumps({"image": name, "strips": len(strips), "detect_ms": round((t1 - t0) / reps * 1e3, 3),
                      "image_analysis_ms_per_strip": round((t2 - t1) / reps / max(1, len(strips)) * 1e3, 3)}))

run(synth.synth_rgb(1536, 2048, 8, 1234), "1536x2048x3 uint8")
run(synth.synth_u16(2048, 2048, 40, 1234), "2048x2048 uint16")
run(synth.synth_rgb(512, 512, 4, 7), "512x512x3 uint8")
