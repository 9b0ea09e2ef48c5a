"""f4 I/O shell (host side, no GPU): xlsx writer / reader round trip on the reference's own golden rows, overlay drawing, and the
configs[0] plumbing on the REAL sample inputs (tests/golden/input/Test{1,2}.png, MIT-licensed data of the reference): tile rectangles in
visiting order and letterboxed network input shapes must equal SURVEY.md Appendix C."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_xlsx_csv

import oriented_object_detection_amd  # noqa: F401
from oriented_object_detection_amd import io_shell, ops


def test_xlsx_round_trip_of_the_reference_goldens(tmp_path):
    for name in ("Test1", "Test2"):
        names, boxes, conf, angle = load_xlsx_csv(name)
        rows = [[names[i]] + [float(v) for v in boxes[i]] + [float(conf[i]), float(angle[i])] for i in range(len(names))]
        path = str(tmp_path / f"{name}.xlsx")
        io_shell.write_xlsx(path, rows)
        back = io_shell.read_xlsx(path)
        assert back[0] == io_shell.XLSX_COLUMNS
        assert len(back) == len(rows) + 1
        for r, b in zip(rows, back[1:]):
            assert b[0] == r[0]
            assert [float(v) for v in b[1:]] == r[1:]  # shortest round-trip decimal text: bit-exact doubles
        pd = pytest.importorskip("pandas")
        try:
            df = pd.read_excel(path)  # needs openpyxl: absent offline -> skipped, present elsewhere -> the file must load
        except ImportError:
            continue
        assert list(df.columns) == io_shell.XLSX_COLUMNS and len(df) == len(rows)


def test_overlay_draws_polygons_and_labels():
    img = np.full((120, 160, 3), 200, np.uint8)
    dets = [(20.4, 30.2, 80.9, 30.2, 80.9, 70.7, 20.4, 70.7, 1, 0.91, 45.0), (100.0, 20.0, 140.0, 40.0, 130.0, 60.0, 90.0, 40.0, 7, 0.75, 0.0)]
    out = io_shell.draw_detections(img, dets, {1: "Strike", 7: "Bergsturz"})
    assert out.shape == img.shape and out.dtype == np.uint8
    assert tuple(out[30, 50]) == io_shell.CLASS_COLORS[1]  # on the top edge of the first box, in BGR
    assert np.array_equal(img, np.full((120, 160, 3), 200, np.uint8))  # the input is not modified (result_image = image.copy(), :295)
    assert (out != img).any(axis=2).sum() > 300


APPENDIX_C = {  # (image, tile, overlap) -> grid rows x cols, last origin, {crop (h, w): (count, network input (H, W))}
    ("Test1", 416, 100): ((3, 3), (632, 632), {(416, 416): (4, (416, 416)), (416, 263): (2, (416, 288)), (175, 416): (2, (192, 416)), (175, 263): (1, (288, 416))}),
    ("Test1", 128, 30): ((9, 10), (882, 784), {(128, 128): (56, (128, 128)), (128, 13): (7, (128, 32)), (23, 128): (8, (32, 128)), (23, 13): (1, (128, 96))}),
    ("Test2", 416, 100): ((4, 4), (948, 948), {(416, 416): (6, (416, 416)), (416, 108): (2, (416, 128)), (396, 416): (3, (416, 416)), (80, 416): (3, (96, 416)), (80, 108): (1, (320, 416))}),
    ("Test2", 128, 30): ((11, 11), (980, 980), {(128, 128): (100, (128, 128)), (128, 76): (10, (128, 96)), (48, 128): (10, (64, 128)), (48, 76): (1, (96, 128))}),
}


@pytest.mark.parametrize("key", sorted(APPENDIX_C))
def test_tile_grid_and_letterbox_shapes_on_the_real_inputs(key):
    from oriented_object_detection_amd import detect as D
    name, tile, overlap = key
    img = D.imread_bgr(os.path.join(GOLDEN, "input", name + ".png"))
    assert img is not None and img.dtype == np.uint8 and img.shape[2] == 3
    H, W = img.shape[:2]
    assert (W, H) == {"Test1": (895, 807), "Test2": (1056, 1028)}[name]
    rects = ops.tile_grid(H, W, tile, overlap)
    (rows, cols), last, shapes = APPENDIX_C[key]
    assert len(rects) == rows * cols
    step = tile - overlap
    exp = [(x, y, min(x + tile, W), min(y + tile, H)) for y in range(0, H, step) for x in range(0, W, step)]  # Detect_OBB.py:216-220
    assert [tuple(int(v) for v in r) for r in rects] == exp
    assert (int(rects[-1][0]), int(rects[-1][1])) == last
    seen = {}
    for x, y, x2, y2 in rects:
        seen.setdefault((int(y2 - y), int(x2 - x)), 0)
        seen[(int(y2 - y), int(x2 - x))] += 1
    for crop, (count, net) in shapes.items():
        assert seen.get(crop) == count, (crop, seen)
        p = ops.letterbox_shape(crop[0], crop[1], tile)
        assert (p["out_h"], p["out_w"]) == net, (crop, p)
    assert D.imread_bgr("/nonexistent/file.png") is None  # unreadable image -> None (the caller warns and returns, :271-273)
