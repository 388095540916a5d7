"""Short runs of the long-sequence fuzzers under tools/ (one handle, many calls of every shape and dtype, every result
against the oracle): a handle carries state from call to call -- counters that pick the next call's paths, a workspace
that grows, a table that stays zero between calls -- and these sequences are its test.  The full-length runs (tens of
thousands of calls) are started by hand: `python tools/detect_fuzz.py 3000 300` and friends."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_detect_sequence_on_one_handle():
    _tool("detect_fuzz").run(80, 400)          # seeds 400..479: seed 408 is the one-pixel-wide colour image (rgb2gray's dot-product order)


def test_analyze_and_peaks_sequence_on_one_handle():
    _tool("analyze_fuzz").run(250, 12)


def test_roi_sets_sequence_on_one_handle():
    _tool("roi_fuzz").run(120, 3)


def test_static_entry_point_sequence_against_analyze_batch():
    _tool("static_fuzz").run(250, 7)


def test_scalar_float64_paths_over_more_seeds():
    _tool("dtype_fuzz").run(12)


def test_drifting_series_through_the_predicted_threshold_mode():
    _tool("series_fuzz").run(150, 3)           # colour and uint16 frames of a drifting scene, other scenes mixed in (bit plane / list / fallback / back-off)
