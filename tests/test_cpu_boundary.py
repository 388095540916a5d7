"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/lfa.h declares, struct
layouts match, argument errors are reported without a GPU, the shared sequential device code (compiled for the host by
tests/hostcheck) reproduces the golden vectors, and the multi-rank shard/gather plumbing is correct under gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import image_analysis_amd._capi as capi
    return capi


def test_every_declared_symbol_is_exported(capi):
    hdr = open(os.path.join(ROOT, "include", "lfa.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(lfa_\w+)\s*\(", hdr, flags=re.M))
    assert declared and declared == set(capi.EXPORTED), declared ^ set(capi.EXPORTED)
    for name in declared:
        assert getattr(capi.lib, name) is not None
    assert capi.lib.lfa_version() == 1


def test_struct_layouts_match_header(capi):
    assert capi.STRIP_DT.itemsize == 24 and capi.ROI_DT.itemsize == 20
    assert capi.PEAKS_DT.itemsize == 40 and capi.PEAKS_DT.fields["area"][1] == 24
    assert capi.INFO_DT.itemsize == 104 and capi.INFO_DT.fields["gmin"][1] == 16 and capi.INFO_DT.fields["otsu_bin"][1] == 64


def test_errors_without_gpu_are_reported_not_swallowed(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = capi.lib.lfa_create(0, 64, 64, 3, 1, C.byref(h))
    assert rc == -2 and b"device" in capi.lib.lfa_last_error().lower()
    assert capi.lib.lfa_create(0, 0, 64, 3, 1, C.byref(h)) == -1
    with pytest.raises(RuntimeError):
        capi.Engine(device=0)
    import image_analysis_amd.detection as det
    with pytest.raises(RuntimeError):                      # product path fails loudly: no CPU fallback
        det.detect_lateral_flow_tests(np.zeros((32, 32), np.uint8))
    with pytest.raises(ValueError):
        det.detect_lateral_flow_tests(np.zeros((2, 2, 2, 2), np.uint8))


def test_host_tables(capi):
    from oracle import lfa_oracle as O
    assert np.array_equal(capi.numpy_gamma_lut_u8(), O.gamma_lut_u8())
    l16 = capi.numpy_gamma_lut_u16()
    assert np.array_equal(l16, O.gamma_lut_u16()) and np.all(np.diff(l16.astype(np.int64)) >= 0)
    x = np.arange(65536, dtype=np.uint16)
    assert np.array_equal(O.adjust_gamma(x, 0.4), l16)
    assert np.array_equal(O.adjust_gamma(x, 1), x)           # gamma 1 is the identity (probed)
    v = capi.host_gray_variant()
    assert 0 <= v < capi.GRAY_NVARIANTS


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_point_operation_tables_equal_the_oracle_chain(capi, dtype):
    """point_op_lut(v) == invert(rescale_intensity(adjust_gamma(v))) of the oracle for every pixel value."""
    from oracle import lfa_oracle as O
    top = int(np.iinfo(dtype).max)
    x = np.arange(top + 1, dtype=dtype)
    cases = [(1, None, None, False), (0.4, None, None, False), (2.2, None, None, True), (1, 10, top // 2, False),
             (0.75, 3.5, top * 0.9, True), (1, 40, 40, False), (1.3, top, 0, False), (0, 5, 100, True)]
    for gamma, lo, hi, inv in cases:
        want = O.adjust_gamma(x, gamma)
        if lo is not None or hi is not None:
            want = O.rescale_intensity(want, (lo, hi))
        if inv:
            want = O.invert(want)
        got = capi.point_op_lut(dtype, gamma, lo, hi, inv)
        assert got.dtype == np.dtype(dtype) and np.array_equal(got, want), (gamma, lo, hi, inv)
    with pytest.raises(TypeError):
        capi.point_op_lut(dtype, 1, None, 7)
    with pytest.raises(ValueError):
        capi.point_op_lut(dtype, -1)


def test_dropin_list_logic_matches_reference_vectors(golden_dir):
    import image_analysis_amd.detection as det
    g = np.load(os.path.join(golden_dir, "ref_region_lists.npz"))
    for t in range(int(g["n_trials"])):
        det.min_maxr, det.min_maxc = (int(v) for v in g[f"t{t}_params"])
        acc = g[f"t{t}_accepted"].tolist()
        seen = []
        for a, b, c, d in acc:
            assert det.check_duplicate(a, c, b, d, seen)
            seen.append([a, b, c, d])
        assert det.remove_outliers(acc) == g[f"t{t}_kept"].tolist()
    det.min_maxr, det.min_maxc = 0, 0


# ---- the sequential code the kernels run (lfa_algo.h), compiled for the host ---------------------------------------
@pytest.fixture(scope="module")
def hc():
    d = os.path.join(ROOT, "tests", "hostcheck")
    so = os.path.join(d, "libhostcheck.so")
    src = os.path.join(d, "hostcheck.cpp")
    hdr = os.path.join(ROOT, "image-analysis_amd", "csrc", "lfa_algo.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.hc_pairwise.restype = C.c_double
    lib.hc_linspace_at.restype = C.c_double
    lib.hc_linspace_at.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    lib.hc_otsu_float256.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double)]
    return lib


def test_device_peak_code_matches_reference_vectors(hc, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_profile_peaks.npz"))
    dp = C.POINTER(C.c_double)
    for i in range(int(g["n_profiles"])):
        x = np.ascontiguousarray(g[f"x{i}"])
        for inv in (0, 1):
            s, e, a = C.c_int(), C.c_int(), C.c_double()
            hc.hc_analyze_peaks(x.ctypes.data_as(dp), len(x), inv, 25, C.byref(s), C.byref(e), C.byref(a))
            assert (s.value, e.value) == tuple(int(v) for v in g[f"p{i}_{inv}"]) and a.value == float(g[f"a{i}_{inv}"])
    for k in range(int(g["n_cases"])):
        prof = np.ascontiguousarray(g[f"c{k}_profile"])
        inv = int(g[f"c{k}_inverse"])
        st, en, ar = (C.c_int * 2)(), (C.c_int * 2)(), (C.c_double * 2)()
        n = hc.hc_two_peaks(prof.ctypes.data_as(dp), len(prof), inv, int(g[f"c{k}_range"]), 0 if inv else 1, 25, st, en, ar)
        assert n == int(g[f"c{k}_n"])
        for q in range(n):
            assert (st[q], en[q]) == tuple(int(v) for v in g[f"c{k}_idx"][q]) and ar[q] == g[f"c{k}_area"][q]


def test_device_sum_and_linspace_match_numpy(hc):
    rng = np.random.default_rng(0)
    dp = C.POINTER(C.c_double)
    for n in list(range(0, 300)) + [1000, 5000]:
        a = rng.normal(1, 100, n)
        assert hc.hc_pairwise(a.ctypes.data_as(dp), n) == (a.sum() if n else 0.0)
    for _ in range(100):
        a, b = rng.normal(0, 10, 2)
        n = int(rng.integers(1, 200))
        ref = np.linspace(a, b, n)
        assert all(hc.hc_linspace_at(a, b, n, i) == ref[i] for i in range(n))


def test_device_big_image_int_otsu_matches_oracle(hc):
    """Integer-image Otsu with float32-rounded weights (images above 2^24 pixels): the shared sequential code against the
    oracle's thresholding.py restatement on histograms whose counts reach 2^31 in total, with a gamma table that merges
    neighbouring input values into one bin and leaves gamma bins empty."""
    from oracle import lfa_oracle as O
    rng = np.random.default_rng(5)
    hc.hc_otsu_int_seq.restype = C.c_int
    hc.hc_otsu_int_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for trial in range(40):
        nb = int(rng.integers(2, 3000))
        scale = 10 ** int(rng.integers(3, 7))
        cnt = (rng.random(nb) ** 3 * scale).astype(np.uint32)
        if trial % 3 == 0:                                            # two far modes of very different weight
            cnt[: nb // 3] *= 50
        cnt[0] = max(cnt[0], 1); cnt[-1] = max(cnt[-1], 1)            # xmin and xmax are occupied by definition
        gamma = float(rng.choice([0.4, 1.0, 2.0]))
        x = np.arange(nb) + int(rng.integers(0, 100))
        lv = np.floor(((x / x.max()) ** gamma) * x.max()).astype(np.int32)   # non-decreasing, with ties and gaps
        # the reference's histogram: one bin per gamma value from lv.min() to lv.max()
        vmin, vmax = int(lv.min()), int(lv.max())
        if vmin == vmax:
            continue
        counts = np.bincount(lv - vmin, weights=cnt.astype(np.float64), minlength=vmax - vmin + 1).astype(np.int64)
        want_v, _ = O.otsu_from_hist(counts, np.arange(vmin, vmax + 1))
        b = hc.hc_otsu_int_seq(cnt.ctypes.data, lv.ctypes.data, nb)
        assert b >= 0 and int(lv[b]) == int(want_v), (trial, nb, scale, gamma)


def test_device_float_otsu_matches_oracle(hc):
    from oracle import lfa_oracle as O
    rng = np.random.default_rng(5)
    for t in range(20):
        img = rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
        if t % 2:
            img[:, :32] //= 4
        v = O.adjust_gamma(O.rgb2gray(img), 0.4)
        cnt = np.histogram(v.reshape(-1), bins=256)[0].astype(np.uint32)
        c = C.c_double()
        hc.hc_otsu_float256(cnt.ctypes.data, float(v.min()), float(v.max()), C.byref(c))
        assert c.value == O.threshold_otsu(v)


# ---- N > 1: shard + gather under gloo, world_size 2 ------------------------------------------------------------------
WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from image_analysis_amd.shard import shard_range, gather_strip_records
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
n = 7
lo, hi = shard_range(n, rank, 2)
dt = np.dtype([("image", "<i4"), ("minr", "<i4"), ("val", "<f8")])
counts = np.array([(i * 3) %% 4 for i in range(lo, hi)], dtype=np.int32)      # includes images with zero strips
recs = np.zeros(int(counts.sum()), dtype=dt)
k = 0
for i, c in zip(range(lo, hi), counts):
    for q in range(c):
        recs[k] = (i, 10 * i + q, i + q / 8); k += 1
ac, ar = gather_strip_records(counts, recs, n)
want_c = np.array([(i * 3) %% 4 for i in range(n)], dtype=np.int32)
assert np.array_equal(ac, want_c), (ac, want_c)
exp = [(i, 10 * i + q, i + q / 8) for i in range(n) for q in range(want_c[i])]
assert [tuple(r) for r in ar.tolist()] == exp
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_shard_and_gather_world_size_2_gloo(tmp_path):
    from image_analysis_amd.shard import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert shard_range(3, 0, 1) == (0, 3) and shard_range(1, 1, 2) == (0, 1) and shard_range(1, 0, 2) == (0, 0)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_numpy_float_box_sum_order_is_what_k_box_sums_restates():
    """k_regions.hip (float64 images, foreground corner): np.mean of a strided 2-D box = pairwise sums over C-order buffers of
    8192 elements added up in turn; of a colour box over axes (0, 1) = one running sum per channel in raster order.  Pure numpy:
    if a numpy upgrade changes this order, this test says so before the GPU parity tests do."""
    def pw(a):
        n = len(a)
        if n < 8:
            r = 0.0
            for x in a:
                r = r + x
            return r
        if n <= 128:
            r = [a[i] for i in range(8)]
            lim = n - n % 8
            for i in range(8, lim, 8):
                for k in range(8):
                    r[k] = r[k] + a[i + k]
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            for i in range(lim, n):
                res = res + a[i]
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])

    rng = np.random.default_rng(8192)
    img = rng.random((500, 300))
    for h, w in ((1, 1), (7, 5), (120, 12), (130, 51), (400, 40), (33, 250)):
        v = img[11:11 + h, 7:7 + w]
        flat = [float(x) for x in v.ravel()]
        acc = 0.0
        for i in range(0, len(flat), 8192):
            acc = acc + pw(flat[i:i + 8192])
        assert np.float64(acc / (h * w)).tobytes() == np.float64(np.mean(v)).tobytes(), (h, w)
    img3 = rng.random((200, 100, 3))
    for h, w in ((1, 1), (9, 4), (120, 12), (150, 51)):
        v = img3[5:5 + h, 3:3 + w, :]
        a = np.zeros(3)
        for y in range(h):
            for x in range(w):
                a = a + v[y, x]
        assert (a / (h * w)).tobytes() == np.mean(v, axis=(0, 1)).tobytes(), (h, w)


def test_gray_variant_is_probed_per_image_width():
    """`rgb @ coeffs` is a dot product for one-pixel-wide images and a matrix-vector product otherwise: the binding asks the
    host's numpy for the rounding order of each width it meets (and finds one of the known orders for every width)."""
    import importlib
    capi = importlib.import_module("image_analysis_amd._capi")
    got = {w: capi.host_gray_variant(w) for w in (1, 2, 3, 8, 17, 64, 250)}
    assert all(0 <= v <= 5 for v in got.values())
    assert capi.host_gray_variant(64) == capi.host_gray_variant()


def test_host_pow_tables_reproduce_the_reference_histogram_and_mask(capi):
    """_capi.host_pow_tables (the LfaHostPow record behind lfa_set_host_pow): bin(g) = the largest k with edges_g[k] <= g must
    be the bin np.histogram gives gray ** 0.4 (exposure.py:298-301, histograms.py:800-859), and gray >= thr_g[b] must be
    gray ** 0.4 > the centre of bin b (thresholding.py:378-381, detection.py:47) -- for EVERY gray value, including the ones
    that sit on an edge in gamma space.  Checked on dense random grays, on the 256 grays of grey pixels and on the exact
    tie gmax / 32 (gamma value vmax / 4 = edge 64 when gmin = 0)."""
    from oracle import lfa_oracle as O
    rng = np.random.default_rng(11)
    grey = O.rgb2gray(np.arange(256, dtype=np.uint8)[:, None, None].repeat(3, axis=2))[:, 0]
    cases = [(0.0, float(grey[128]), np.concatenate([grey[:129], [grey[128] / 32]])),
             (float(grey[3]), float(grey[250]), grey[3:251]),
             (0.02, 0.9, rng.uniform(0.02, 0.9, 200000)),
             (0.3, 0.3000001, rng.uniform(0.3, 0.3000001, 5000))]
    for gmin, gmax, g in cases:
        g = np.ascontiguousarray(np.concatenate([g, [gmin, gmax]]))
        tab = capi.host_pow_tables(gmin, gmax)
        v = g ** 0.4
        assert tab[0] == v.min() and tab[1] == v.max()
        want_counts, centres = O.histogram_image_range(v)
        edges_g, thr_g = tab[2:259], tab[259:]
        bins = np.clip(np.searchsorted(edges_g, g, side="right") - 1, 0, 255)
        assert np.array_equal(np.bincount(bins, minlength=256), want_counts)
        for b in (0, 1, 63, 64, 128, 254):
            assert np.array_equal(g >= thr_g[b], v > centres[b]), b
    const = capi.host_pow_tables(0.25, 0.25)                       # all-equal image: edge 0 = gmin, everything else unreachable
    assert const[0] == const[1] and const[2] == 0.25 and np.isinf(const[3:]).all()


def test_minmax_screen_bound_holds_for_every_byte_triple():
    """k_minmax_rgb screens pixels with g = 76*b0 + 255*b1 + 26*b2 (one v_dot4) instead of the exact integer gray
    G = 2125*b0 + 7154*b1 + 721*b2 and relies on 28*g - 2550 <= G <= 28*g + 3570: a pixel with G <= Wlo has g <= (Wlo + 2550) // 28,
    one with G >= Whi has g >= ceil((Whi - 3570) / 28).  Checked over all 2^24 triples (as the extreme values of a linear form
    and by brute force on a slice), together with the two screening inequalities on random bounds."""
    assert (2125 - 28 * 76, 7154 - 28 * 255, 721 - 28 * 26) == (-3, 14, -7)
    assert -3 * 255 - 7 * 255 == -2550 and 14 * 255 == 3570
    b = np.arange(256, dtype=np.int64)
    for b0 in (0, 1, 127, 254, 255):
        G = 2125 * b0 + 7154 * b[:, None] + 721 * b[None, :]
        g = 76 * b0 + 255 * b[:, None] + 26 * b[None, :]
        assert ((28 * g - 2550 <= G) & (G <= 28 * g + 3570)).all()
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, size=(200000, 3))
    G = 2125 * t[:, 0] + 7154 * t[:, 1] + 721 * t[:, 2]
    g = 76 * t[:, 0] + 255 * t[:, 1] + 26 * t[:, 2]
    for Wlo, Whi in rng.integers(0, 2550001, size=(50, 2)):
        Alo = (Wlo + 2550) // 28
        Ahi = (Whi - 3570 + 27) // 28 if Whi > 3570 else 0
        assert (g[G <= Wlo] <= Alo).all() and (g[G >= Whi] >= Ahi).all()
