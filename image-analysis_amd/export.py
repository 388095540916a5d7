"""Host-side writers for batched results.

The reference saves ONE profile per click: `pd.DataFrame(profile, columns=["Intensity"]).to_excel(...)`
(LFA-Image_Analysis/detection.py:741-744), i.e. a sheet whose column A is the row index, column B the header
"Intensity" and the float64 row means -- that is the layout of images/excel2.xlsx, from which this repository's
known-answer fixture was read.  Neither pandas' Excel engines nor xlsxwriter are installed here, so the workbook is
written directly: an .xlsx file is a zip of a few XML parts, and `write_profile_xlsx` emits the parts xlsxwriter emits for
this sheet -- the worksheet and the drawing part byte for byte as in the reference's own export (pinned against
images/excel2.xlsx by tests/test_frames.py), the three pictures the reference pastes in at D3 / M3 / O3 (:749-758)
included when the caller passes them.  The same table as CSV (`repr`-exact float64) for batch work.

    write_profile_xlsx(path, profile, images=())        one strip, the reference's workbook
    write_profile_csv(path, profile)                    one strip, the reference's sheet layout as CSV
    write_batch_csv(profiles_path, summary_path, results)   everything analyze_stream / analyze_batch returned
"""
import csv
import io
import re
import zipfile

import numpy as np

_XML = '<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
_NS_MAIN = "http://schemas.openxmlformats.org/spreadsheetml/2006/main"
_NS_REL = "http://schemas.openxmlformats.org/officeDocument/2006/relationships"
_NS_PKG = "http://schemas.openxmlformats.org/package/2006/relationships"
_EMU = 9525                    # EMU per pixel at 96 dpi
_COL_PX, _ROW_PX = 64, 20      # default column width / row height in pixels


def _num(v):
    """xlsxwriter writes numbers with %.16G; when that does not read back to the same float64, 17 digits are used instead
    (the reference would lose the last bit there)."""
    v = float(v)
    t = "%.16G" % v
    return t if float(t) == v else repr(v)


def _sheet_xml(profile, with_drawing):
    n = len(profile)
    rows = ['<row r="1" spans="1:2"><c r="B1" s="1" t="s"><v>0</v></c></row>']
    for i, v in enumerate(profile):
        r = i + 2
        rows.append(f'<row r="{r}" spans="1:2"><c r="A{r}" s="1"><v>{i}</v></c><c r="B{r}"><v>{_num(v)}</v></c></row>')
    return (_XML + f'<worksheet xmlns="{_NS_MAIN}" xmlns:r="{_NS_REL}"><dimension ref="A1:B{n + 1}"/><sheetViews><sheetView tabSelected="1" '
            'workbookViewId="0"/></sheetViews><sheetFormatPr defaultRowHeight="15"/><sheetData>' + "".join(rows) +
            '</sheetData><pageMargins left="0.7" right="0.7" top="0.75" bottom="0.75" header="0.3" footer="0.3"/>' +
            ('<drawing r:id="rId1"/>' if with_drawing else "") + "</worksheet>")


def _drawing_xml(sizes, cells, names):
    """One twoCellAnchor per picture, as xlsxwriter's insert_image places an unscaled picture at a cell."""
    out = [_XML + '<xdr:wsDr xmlns:xdr="http://schemas.openxmlformats.org/drawingml/2006/spreadsheetDrawing" '
           'xmlns:a="http://schemas.openxmlformats.org/drawingml/2006/main">']
    for k, ((w, h), (col, row), name) in enumerate(zip(sizes, cells, names)):
        out.append(
            f'<xdr:twoCellAnchor editAs="oneCell"><xdr:from><xdr:col>{col}</xdr:col><xdr:colOff>0</xdr:colOff><xdr:row>{row}</xdr:row>'
            f'<xdr:rowOff>0</xdr:rowOff></xdr:from><xdr:to><xdr:col>{col + w // _COL_PX}</xdr:col><xdr:colOff>{(w % _COL_PX) * _EMU}</xdr:colOff>'
            f'<xdr:row>{row + h // _ROW_PX}</xdr:row><xdr:rowOff>{(h % _ROW_PX) * _EMU}</xdr:rowOff></xdr:to><xdr:pic><xdr:nvPicPr>'
            f'<xdr:cNvPr id="{k + 2}" name="Picture {k + 1}" descr="{name}"/><xdr:cNvPicPr><a:picLocks noChangeAspect="1"/></xdr:cNvPicPr>'
            f'</xdr:nvPicPr><xdr:blipFill><a:blip xmlns:r="{_NS_REL}" r:embed="rId{k + 1}"/><a:stretch><a:fillRect/></a:stretch></xdr:blipFill>'
            f'<xdr:spPr><a:xfrm><a:off x="{col * _COL_PX * _EMU}" y="{row * _ROW_PX * _EMU}"/><a:ext cx="{w * _EMU}" cy="{h * _EMU}"/></a:xfrm>'
            '<a:prstGeom prst="rect"><a:avLst/></a:prstGeom></xdr:spPr></xdr:pic><xdr:clientData/></xdr:twoCellAnchor>')
    out.append("</xdr:wsDr>")
    return "".join(out)


_STYLES = ('<styleSheet xmlns="' + _NS_MAIN + '"><fonts count="2"><font><sz val="11"/><color theme="1"/><name val="Calibri"/><family val="2"/>'
           '<scheme val="minor"/></font><font><b/><sz val="11"/><color theme="1"/><name val="Calibri"/><family val="2"/><scheme val="minor"/></font>'
           '</fonts><fills count="2"><fill><patternFill patternType="none"/></fill><fill><patternFill patternType="gray125"/></fill></fills>'
           '<borders count="2"><border><left/><right/><top/><bottom/><diagonal/></border><border><left style="thin"><color auto="1"/></left>'
           '<right style="thin"><color auto="1"/></right><top style="thin"><color auto="1"/></top><bottom style="thin"><color auto="1"/></bottom>'
           '<diagonal/></border></borders><cellStyleXfs count="1"><xf numFmtId="0" fontId="0" fillId="0" borderId="0"/></cellStyleXfs>'
           '<cellXfs count="2"><xf numFmtId="0" fontId="0" fillId="0" borderId="0" xfId="0"/><xf numFmtId="0" fontId="1" fillId="0" borderId="1" '
           'xfId="0" applyFont="1" applyBorder="1" applyAlignment="1"><alignment horizontal="center" vertical="top"/></xf></cellXfs>'
           '<cellStyles count="1"><cellStyle name="Normal" xfId="0" builtinId="0"/></cellStyles><dxfs count="0"/>'
           '<tableStyles count="0" defaultTableStyle="TableStyleMedium9" defaultPivotStyle="PivotStyleLight16"/></styleSheet>')


def _png_bytes(img):
    """PNG bytes and (width, height) of a picture given as PNG bytes or as an array (what cv2.imwrite is handed, :749-754)."""
    from PIL import Image
    if isinstance(img, (bytes, bytearray)):
        with Image.open(io.BytesIO(bytes(img))) as im:
            return bytes(img), im.size
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] >= 3:
        a = np.ascontiguousarray(a[:, :, [2, 1, 0] + list(range(3, a.shape[2]))])       # cv2 writes BGR(A) arrays
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    return buf.getvalue(), (a.shape[1], a.shape[0])


def write_profile_xlsx(path, profile, images=()):
    """The workbook MyGUI.download_data writes (detection.py:735-759): Sheet1 with the row index in column A and
    "Intensity" + the float64 profile in column B (pandas `DataFrame(profile, columns=["Intensity"]).to_excel`), and up to
    three pictures -- the frame, the strip crop, the plot -- at D3, M3 and O3 (`worksheet.insert_image`).  images: PNG
    byte strings or uint8 / uint16 arrays."""
    profile = np.asarray(profile, dtype=np.float64).reshape(-1)
    pics = [_png_bytes(im) for im in images][:3]
    cells = [(3, 2), (12, 2), (14, 2)]                                     # D3, M3, O3 (zero-based column, row)
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        ct = ['<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>',
              '<Default Extension="xml" ContentType="application/xml"/>']
        if pics:
            ct.append('<Default Extension="png" ContentType="image/png"/>')
        ct += ['<Override PartName="/xl/styles.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.styles+xml"/>',
               '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>',
               '<Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>']
        if pics:
            ct.append('<Override PartName="/xl/drawings/drawing1.xml" ContentType="application/vnd.openxmlformats-officedocument.drawing+xml"/>')
        ct.append('<Override PartName="/xl/sharedStrings.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sharedStrings+xml"/>')
        z.writestr("[Content_Types].xml", _XML + '<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">' + "".join(ct) + "</Types>")
        z.writestr("_rels/.rels", _XML + f'<Relationships xmlns="{_NS_PKG}"><Relationship Id="rId1" Type="{_NS_REL}/officeDocument" '
                   'Target="xl/workbook.xml"/></Relationships>')
        z.writestr("xl/workbook.xml", _XML + f'<workbook xmlns="{_NS_MAIN}" xmlns:r="{_NS_REL}"><sheets><sheet name="Sheet1" sheetId="1" r:id="rId1"/>'
                   '</sheets></workbook>')
        z.writestr("xl/_rels/workbook.xml.rels", _XML + f'<Relationships xmlns="{_NS_PKG}"><Relationship Id="rId1" Type="{_NS_REL}/worksheet" '
                   f'Target="worksheets/sheet1.xml"/><Relationship Id="rId3" Type="{_NS_REL}/styles" Target="styles.xml"/>'
                   f'<Relationship Id="rId4" Type="{_NS_REL}/sharedStrings" Target="sharedStrings.xml"/></Relationships>')
        z.writestr("xl/styles.xml", _XML + _STYLES)
        z.writestr("xl/sharedStrings.xml", _XML + f'<sst xmlns="{_NS_MAIN}" count="1" uniqueCount="1"><si><t>Intensity</t></si></sst>')
        z.writestr("xl/worksheets/sheet1.xml", _sheet_xml(profile, bool(pics)))
        if pics:
            z.writestr("xl/worksheets/_rels/sheet1.xml.rels", _XML + f'<Relationships xmlns="{_NS_PKG}"><Relationship Id="rId1" '
                       f'Type="{_NS_REL}/drawing" Target="../drawings/drawing1.xml"/></Relationships>')
            z.writestr("xl/drawings/drawing1.xml", _drawing_xml([p[1] for p in pics], cells, [f"temp{k + 1}.png" for k in range(len(pics))]))
            z.writestr("xl/drawings/_rels/drawing1.xml.rels", _XML + f'<Relationships xmlns="{_NS_PKG}">' + "".join(
                f'<Relationship Id="rId{k + 1}" Type="{_NS_REL}/image" Target="../media/image{k + 1}.png"/>' for k in range(len(pics))) + "</Relationships>")
            for k, (data, _size) in enumerate(pics):
                z.writestr(f"xl/media/image{k + 1}.png", data)


def read_profile_xlsx(path) -> np.ndarray:
    """Column B of Sheet1 of a workbook written by write_profile_xlsx -- or by the reference's download_data."""
    with zipfile.ZipFile(path) as z:
        sheet = z.read("xl/worksheets/sheet1.xml").decode()
    return np.array([float(v) for v in re.findall(r'<c r="B\d+"><v>([^<]+)</v>', sheet)], dtype=np.float64)


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
