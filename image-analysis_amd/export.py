"""Host-side writers for batched results.

The reference saves ONE profile per click: `pd.DataFrame(profile, columns=["Intensity"]).to_excel(...)`
(LFA-Image_Analysis/detection.py:741-744), i.e. a sheet whose column A is the row index, column B the header
"Intensity" and the float64 row means -- that is the layout of images/excel2.xlsx, from which this repository's
known-answer fixture was read.  xlsxwriter is not installed here, so the same table is written as CSV (`repr`-exact
float64, so a profile read back compares equal bit for bit); the three pictures the reference pastes into the sheet
(:749-758) are GUI artefacts and stay with the GUI.

    write_profile_csv(path, profile)                    one strip, the reference's sheet layout
    write_batch_csv(profiles_path, summary_path, results)   everything analyze_stream / analyze_batch returned
"""
import csv

import numpy as np


def write_profile_csv(path, profile):
    """Column A: index, column B: "Intensity" -- the sheet detection.py:741-744 writes, as CSV."""
    profile = np.asarray(profile, dtype=np.float64)
    with open(path, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["", "Intensity"])
        for i, v in enumerate(profile):
            w.writerow([i, repr(float(v))])


def read_profile_csv(path) -> np.ndarray:
    with open(path, newline="") as fi:
        rows = list(csv.reader(fi))
    if not rows or rows[0] != ["", "Intensity"]:
        raise ValueError(f"{path}: not a profile table")
    return np.array([float(r[1]) for r in rows[1:]], dtype=np.float64)


def write_batch_csv(profiles_path, summary_path, results):
    """`results`: iterable of (frame_index, strips, profiles, peaks) as frames.analyze_stream yields them.

    profiles file: frame, strip, row, Intensity (long format, one line per profile value);
    summary file:  frame, strip, minr, maxr, minc, maxc, area, n_peaks, start1, end1, area1, start2, end2, area2."""
    n = 0
    with open(profiles_path, "w", newline="") as fp, open(summary_path, "w", newline="") as fs:
        wp, ws = csv.writer(fp), csv.writer(fs)
        wp.writerow(["frame", "strip", "row", "Intensity"])
        ws.writerow(["frame", "strip", "minr", "maxr", "minc", "maxc", "area", "n_peaks",
                     "start1", "end1", "area1", "start2", "end2", "area2"])
        for frame, strips, profiles, peaks in results:
            for k, s in enumerate(strips):
                for r, v in enumerate(profiles[k]):
                    wp.writerow([frame, k, r, repr(float(v))])
                pk = peaks[k]
                np_ = int(pk["n_peaks"])
                cells = []
                for q in range(2):
                    cells += [int(pk["start"][q]), int(pk["end"][q]), repr(float(pk["area"][q]))] if q < np_ else ["", "", ""]
                ws.writerow([frame, k, int(s["minr"]), int(s["maxr"]), int(s["minc"]), int(s["maxc"]), int(s["area"]), np_] + cells)
                n += 1
    return n
