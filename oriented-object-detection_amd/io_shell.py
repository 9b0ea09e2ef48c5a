"""I/O shell around the hot path: what Detect_OBB.py does before and after process_image's arithmetic --
reading the input image (:270), drawing the detections on a copy of it (:295-324), writing the table as an .xlsx file (:326-330) and
walking the input directory (:745-755).  Host-side Python only: no arithmetic of the hot path lives here, and nothing here is timed by
bench.py.  cv2 / pandas / openpyxl are replaced by Pillow and a minimal writer of the SpreadsheetML parts pandas' default writer emits
(same sheet layout: header row of inline strings, numeric cells as shortest round-trip decimal text), so the output opens in Excel
and reads back through `read_xlsx` / pandas."""
import os
import time
import zipfile
from xml.sax.saxutils import escape

import numpy as np

CLASS_COLORS = {0: (255, 0, 0), 1: (0, 255, 0), 2: (0, 0, 255), 3: (255, 255, 0), 4: (255, 0, 255), 5: (0, 255, 255), 6: (0, 0, 0),
                7: (240, 34, 0), 8: (50, 20, 60), 9: (60, 50, 20), 10: (200, 150, 80), 11: (100, 200, 150)}  # BGR, Detect_OBB.py:59-72
XLSX_COLUMNS = ["Class", "X1", "Y1", "X2", "Y2", "X3", "Y3", "X4", "Y4", "Confidence", "Angle"]  # Detect_OBB.py:328


def _col(i):
    s = ""
    i += 1
    while i:
        i, r = divmod(i - 1, 26)
        s = chr(65 + r) + s
    return s


def write_xlsx(path, rows, columns=XLSX_COLUMNS):
    """rows: lists [label, x1..y4, conf, angle] (Detect_OBB.py:322, 326-330: DataFrame(rows, columns).to_excel(path, index=False))."""
    def cell(r, c, v):
        ref = f"{_col(c)}{r}"
        if isinstance(v, str):
            return f'<c r="{ref}" t="inlineStr"><is><t>{escape(v)}</t></is></c>'
        return f'<c r="{ref}" t="n"><v>{repr(float(v)) if float(v) != int(float(v)) or abs(float(v)) >= 1e15 else int(float(v))}</v></c>'
    body = [f'<row r="1">' + "".join(cell(1, c, h) for c, h in enumerate(columns)) + "</row>"]
    for r, row in enumerate(rows, start=2):
        body.append(f'<row r="{r}">' + "".join(cell(r, c, v) for c, v in enumerate(row)) + "</row>")
    dim = f"A1:{_col(len(columns) - 1)}{len(rows) + 1}"
    sheet = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
             '<worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main">'
             f'<dimension ref="{dim}"/><sheetData>' + "".join(body) + "</sheetData></worksheet>")
    parts = {
        "[Content_Types].xml": '<?xml version="1.0" encoding="UTF-8" standalone="yes"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
                               '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
                               '<Default Extension="xml" ContentType="application/xml"/>'
                               '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>'
                               '<Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/></Types>',
        "_rels/.rels": '<?xml version="1.0" encoding="UTF-8" standalone="yes"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                       '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>',
        "xl/workbook.xml": '<?xml version="1.0" encoding="UTF-8" standalone="yes"?><workbook xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" '
                           'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets><sheet name="Sheet1" sheetId="1" r:id="rId1"/></sheets></workbook>',
        "xl/_rels/workbook.xml.rels": '<?xml version="1.0" encoding="UTF-8" standalone="yes"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                                      '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet1.xml"/></Relationships>',
        "xl/worksheets/sheet1.xml": sheet,
    }
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        for name, data in parts.items():
            z.writestr(name, data)


def read_xlsx(path):
    """-> list of rows (strings as written) of sheet1: reads both this writer's files and the reference's own Output/*.xlsx."""
    import re
    s = zipfile.ZipFile(path).read("xl/worksheets/sheet1.xml").decode()
    rows = []
    for r in re.findall(r"<row [^>]*>(.*?)</row>", s, re.S):
        vals = []
        for inner in re.findall(r"<c [^>]*>(.*?)</c>", r, re.S):
            m = re.search(r"<t[^>]*>(.*?)</t>", inner, re.S)
            vals.append(m.group(1) if m else re.search(r"<v>(.*?)</v>", inner).group(1))
        rows.append(vals)
    return rows


def draw_detections(image_bgr, dets, class_names, class_colors=CLASS_COLORS):
    """Copy of the image with every detection's polygon (2 px) and its "<label> <conf>" caption (Detect_OBB.py:295-324).  Pillow instead of
    cv2.polylines / cv2.putText: same geometry and colours, different rasteriser and font."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.ascontiguousarray(image_bgr[:, :, ::-1]))
    d = ImageDraw.Draw(im)
    H, W = image_bgr.shape[:2]
    for (x1, y1, x2, y2, x3, y3, x4, y4, cls_id, conf, angle) in dets:
        b, g, r = class_colors.get(cls_id, (0, 255, 255))
        pts = [(int(x1), int(y1)), (int(x2), int(y2)), (int(x3), int(y3)), (int(x4), int(y4))]  # np.int32 truncation, :310
        d.line(pts + [pts[0]], fill=(r, g, b), width=2)
        tx = int(max(0, min(W - 1, round(min(x1, x2, x3, x4)))))
        ty = int(max(0, min(H - 1, round(min(y1, y2, y3, y4) - 10))))
        d.text((tx, max(0, ty - 10)), f"{class_names.get(cls_id, f'Class{cls_id}')} {conf:.2f}", fill=(r, g, b))
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def save_outputs(image_bgr, image_path, output_dir, dets, class_names):
    """<name>_detected.jpg + <name>.xlsx next to each other in output_dir (Detect_OBB.py:297-330)."""
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    name = os.path.basename(image_path)
    # .jpg / .png names come out exactly as the reference's str.replace gives them (:299, :322); for the other extensions its loop accepts
    # (.jpeg / .tif / .tiff, :750) that replace is a no-op and pandas then rejects the ".tif" workbook name -- here the stem is used
    stem = os.path.splitext(name)[0]
    xlsx = os.path.join(output_dir, stem + ".xlsx")
    jpg = os.path.join(output_dir, stem + "_detected.jpg")
    Image.fromarray(draw_detections(image_bgr, dets, class_names)[:, :, ::-1]).save(jpg, quality=95)
    rows = [[class_names.get(d[8], f"Class{d[8]}")] + [float(v) for v in d[:8]] + [float(d[9]), float(d[10])] for d in dets]
    write_xlsx(xlsx, rows)
    return jpg, xlsx


def main(input_dir="Input", output_dir="Output", models=None, cfg=None):
    """The script's main loop (Detect_OBB.py:745-755): every .jpg / .png / .jpeg / .tif / .tiff of input_dir (the reference's extension
    tuple, :750; Pillow reads the first page of a TIFF) through process_image, outputs into output_dir."""
    from . import detect as D
    cfg = cfg or D.DEFAULT
    t0 = time.time()
    os.makedirs(output_dir, exist_ok=True)
    done = {}
    for fname in sorted(os.listdir(input_dir)):
        if not fname.lower().endswith((".jpg", ".png", ".jpeg", ".tif", ".tiff")):
            continue
        path = os.path.join(input_dir, fname)
        t1 = time.time()
        image = D.imread_bgr(path)
        if image is None:
            print(f"[Warn] Could not read image: {path}")
            continue
        rows = D.process_image(image, None, models, cfg)
        save_outputs(image, path, output_dir, rows, cfg.CLASS_NAMES)
        print(f"--- {time.time() - t1:.3f} seconds ---")
        done[path] = rows
    print(f"Total time: {time.time() - t0:.2f} s")
    return done
