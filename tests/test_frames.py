"""File readers and the frame pipeline in front of the device path.

CPU part: imread / iter_tiff_pages on files written here with PIL (lossless: TIFF, PNG), shapes, dtypes and channel
order as detection.py:454 would deliver them.  GPU part: analyze_stream == the oracle frame by frame, in order, across
batch boundaries."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import lfa_oracle as O


def _write_stack(path, frames):
    ims = [Image.fromarray(f) for f in frames]
    ims[0].save(path, save_all=True, append_images=ims[1:])


def test_tiff_stack_pages_round_trip(tmp_path):
    from image_analysis_amd import frames as F
    rng = np.random.default_rng(1)
    stack = [rng.integers(0, 4096, size=(48, 64), dtype=np.uint16) for _ in range(5)]
    p = str(tmp_path / "stack.tif")
    _write_stack(p, stack)
    pages = list(F.iter_tiff_pages(p))
    assert len(pages) == 5 and all(a.dtype == np.uint16 and a.flags.c_contiguous for a in pages)
    assert all(np.array_equal(a, b) for a, b in zip(pages, stack))
    assert np.array_equal(F.imread(p), np.stack(stack))                  # tifffile.imread stacks the pages
    one = str(tmp_path / "one.tiff")
    Image.fromarray(stack[0]).save(one)
    assert np.array_equal(F.imread(one), stack[0]) and F.imread(one).ndim == 2


def test_imread_colour_is_bgr_and_grey_is_three_channels(tmp_path):
    from image_analysis_amd import frames as F
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, size=(20, 30, 3), dtype=np.uint8)
    p = str(tmp_path / "c.png")
    Image.fromarray(rgb).save(p)
    got = F.imread(p)
    assert got.dtype == np.uint8 and got.flags.c_contiguous and np.array_equal(got, rgb[:, :, ::-1])
    g = rng.integers(0, 256, size=(20, 30), dtype=np.uint8)
    pg = str(tmp_path / "g.png")
    Image.fromarray(g).save(pg)
    got = F.imread(pg)                                                   # cv2.imread default flag: IMREAD_COLOR
    assert got.shape == (20, 30, 3) and all(np.array_equal(got[:, :, c], g) for c in range(3))


def _check_frame(frame, strips, profiles, peaks):
    mm_c = int(frame.shape[1] / 40)
    mm_r = mm_c * 20
    d = O.detect_debug(frame, mm_r, mm_c, 10)
    assert [[int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"])] for s in strips] == d["strips"]
    for k, box in enumerate(d["strips"]):
        wp, wk = O.image_analysis(frame[box[0]:box[1], box[2]:box[3]], False)
        assert np.array_equal(profiles[k], wp)
        assert peaks[k]["n_peaks"] == len(wk)
        for i, (s, e, a) in enumerate(wk):
            assert (peaks[k]["start"][i], peaks[k]["end"][i]) == (s, e) and peaks[k]["area"][i] == a
    return len(d["strips"])


@pytest.mark.gpu
def test_analyze_stream_tiff_stack_matches_oracle(tmp_path):
    from image_analysis_amd import frames as F
    from image_analysis_amd.synth import synth_u16
    stack = [synth_u16(512, 512, 8, 300 + i) for i in range(5)]
    p = str(tmp_path / "stack.tif")
    _write_stack(p, stack)
    seen, total = [], 0
    for idx, strips, profiles, peaks in F.analyze_stream(F.iter_tiff_pages(p), batch=2):     # batches of 2, 2, 1
        seen.append(idx)
        total += _check_frame(stack[idx], strips, profiles, peaks)
    assert seen == [0, 1, 2, 3, 4] and total >= 5


@pytest.mark.gpu
def test_analyze_stream_colour_files_and_errors(tmp_path):
    from image_analysis_amd import frames as F
    from image_analysis_amd.synth import synth_rgb
    paths = []
    for i in range(3):
        p = str(tmp_path / f"img{i}.png")
        Image.fromarray(synth_rgb(384, 512, 6, 40 + i)[:, :, ::-1].copy()).save(p)       # file holds RGB; imread returns BGR
        paths.append(p)
    frames = [F.imread(p) for p in paths]
    assert all(np.array_equal(f, synth_rgb(384, 512, 6, 40 + i)) for i, f in enumerate(frames))
    out = list(F.analyze_stream(iter(frames), batch=8))
    assert [o[0] for o in out] == [0, 1, 2]
    for idx, strips, profiles, peaks in out:
        assert _check_frame(frames[idx], strips, profiles, peaks) >= 3
    assert list(F.analyze_stream(iter([]))) == []
    with pytest.raises(ValueError):
        list(F.analyze_stream(iter([frames[0], frames[1][:100]]), batch=4))
    with pytest.raises(TypeError):
        list(F.analyze_stream(iter([frames[0].astype(np.float32)])))


def test_profile_csv_has_the_reference_sheet_layout_and_round_trips(tmp_path, golden_dir):
    import os
    from image_analysis_amd import export as E
    ka = np.load(os.path.join(golden_dir, "excel2_known_answer.npz"))
    prof = ka["profile_column_B"]                          # column B of the sheet the reference itself wrote
    p = str(tmp_path / "p.csv")
    E.write_profile_csv(p, prof)
    lines = open(p).read().splitlines()
    assert lines[0] == ",Intensity" and lines[1].startswith("0,") and len(lines) == len(prof) + 1
    assert np.array_equal(E.read_profile_csv(p), prof)     # bit-exact through text


def test_batch_csv_tables(tmp_path):
    import csv
    from image_analysis_amd import export as E
    import image_analysis_amd._capi as capi
    strips = np.zeros(2, capi.STRIP_DT)
    strips["minr"], strips["maxr"], strips["minc"], strips["maxc"], strips["area"] = [1, 2], [11, 12], [3, 4], [8, 9], [50, 60]
    peaks = np.zeros(2, capi.PEAKS_DT)
    peaks["n_peaks"] = [2, 1]
    peaks["start"][0], peaks["end"][0], peaks["area"][0] = [1, 5], [3, 7], [0.1, 2.5]
    peaks["start"][1][0], peaks["end"][1][0], peaks["area"][1][0] = 2, 4, 1 / 3
    profs = [np.array([1.5, 2.25, 1 / 3]), np.array([4.0])]
    n = E.write_batch_csv(str(tmp_path / "pr.csv"), str(tmp_path / "su.csv"), [(7, strips, profs, peaks)])
    assert n == 2
    pr = list(csv.reader(open(tmp_path / "pr.csv")))
    su = list(csv.reader(open(tmp_path / "su.csv")))
    assert len(pr) == 1 + 4 and pr[3] == ["7", "0", "2", repr(1 / 3)]
    assert su[1][:8] == ["7", "0", "1", "11", "3", "8", "50", "2"] and float(su[1][13]) == 2.5
    assert su[2][7] == "1" and su[2][11:] == ["", "", ""] and float(su[2][10]) == 1 / 3


def test_profile_xlsx_is_the_reference_workbook(tmp_path, golden_dir):
    """export.write_profile_xlsx against the reference's own export images/excel2.xlsx (download_data, detection.py:735-759):
    for the same profile and pictures of the same sizes, the worksheet part and the drawing part are byte-identical to the
    reference's (tests/golden/excel2_sheet1.xml, excel2_drawing1.xml, copied out of that workbook), and the file reads back."""
    import zipfile
    from image_analysis_amd import export as E
    ka = np.load(os.path.join(golden_dir, "excel2_known_answer.npz"))
    prof = ka["profile_column_B"]
    frame_png = open(os.path.join(golden_dir, "excel2_frame.png"), "rb").read()
    crop = np.array(Image.open(os.path.join(golden_dir, "excel2_crop.png")))
    plot = np.zeros((600, 600, 4), np.uint8)                      # the GUI's matplotlib canvas is 600 x 600 RGBA
    p = str(tmp_path / "out.xlsx")
    E.write_profile_xlsx(p, prof, images=[frame_png, crop, plot])
    with zipfile.ZipFile(p) as z:
        assert z.testzip() is None
        assert z.read("xl/worksheets/sheet1.xml") == open(os.path.join(golden_dir, "excel2_sheet1.xml"), "rb").read()
        assert z.read("xl/drawings/drawing1.xml") == open(os.path.join(golden_dir, "excel2_drawing1.xml"), "rb").read()
        assert z.read("xl/media/image1.png") == frame_png
        assert np.array_equal(np.array(Image.open(__import__("io").BytesIO(z.read("xl/media/image2.png")))), crop)
    assert np.array_equal(E.read_profile_xlsx(p), prof)
    # a value that %.16G would not bring back keeps its last bit; no pictures -> no drawing part
    odd = np.array([0.1 + 0.2, 1 / 3, 120.30000000000001, 5e-324, 1e300])
    E.write_profile_xlsx(p, odd)
    assert np.array_equal(E.read_profile_xlsx(p), odd)
    with zipfile.ZipFile(p) as z:
        assert "xl/drawings/drawing1.xml" not in z.namelist()
