"""Pins the torch-CPU oracle of the Ultralytics side of the path (no GPU): architecture checksum against the published
YOLO11-OBB table, graph/blob consistency, and the decode / Fast-NMS / result-construction restatements."""
import math
import struct

import numpy as np
import pytest
import torch

from conftest import CLASS_IDS, load_xlsx_csv
from oracle import geom as og
from oracle import postproc as pp
from oracle.yolo11_obb import Yolo11OBB, make_anchors
import synth


@pytest.fixture(scope="module")
def net():
    return Yolo11OBB("n", nc=12, ch=3, seed=0)


def test_architecture_checksum():
    # published table (Ultralytics docs, nc=80... OBB nc=15 for DOTA): yolo11n-obb 2.7 M params, 17.2 GFLOPs @1024; s: 9.7 M / 57.5
    n = Yolo11OBB("n", nc=15)
    assert abs(n.n_params() / 1e6 - 2.66) < 0.06
    assert abs(2 * n.macs(1024, 1024) / 1e9 - 17.2) < 0.4
    s = Yolo11OBB("s", nc=15)
    assert abs(s.n_params() / 1e6 - 9.7) < 0.1
    assert abs(2 * s.macs(1024, 1024) / 1e9 - 57.5) < 1.0


def test_work_per_tile_contract_figures(net):
    # BASELINE.md section 2 / SURVEY.md section 8(d)
    assert net.macs(416, 416) == 1392703312
    assert net.macs(128, 128) == 131383552
    assert Yolo11OBB("n", nc=12, ch=4).macs(416, 416) == 1398933328
    assert len(net.convs) == 96
    a, s = make_anchors(416, 416)
    assert a.shape == (3549, 2) and float(a[0, 0]) == 0.5 and float(s[-1]) == 32.0


def test_forward_shapes_and_precisions(net):
    x = np.random.default_rng(0).integers(0, 256, (1, 128, 160, 3), dtype=np.uint8)
    y32 = net.forward_raw(x, "fp32")
    assert y32.shape == (1, 16 * 20 + 8 * 10 + 4 * 5, 77) and torch.isfinite(y32).all()
    y16, yb = net.forward_raw(x, "f16"), net.forward_raw(x, "bf16")
    e16, eb = float((y16 - y32).abs().mean()), float((yb - y32).abs().mean())
    assert 0 < e16 < eb < 0.5 and e16 < eb / 3  # fp16 storage is several times closer to the reference's fp32 than bf16
    xb = np.concatenate([x, x[:, ::-1].copy()])
    yy = net.forward_raw(xb, "fp32")
    assert torch.allclose(yy[0], y32[0], atol=2e-3)  # batch independence (torch-CPU conv picks different blocking per batch size)


def test_blob_format(net):
    blob = net.to_blob()
    magic, ver, nrec, nc, ch, width, depth, max_ch, reg_max = struct.unpack_from("<4sIIiiffii", blob, 0)
    assert (magic, ver, nrec, nc, ch, reg_max) == (b"OBBW", 1, 96, 12, 3, 16)
    hdr = struct.calcsize("<4sIIiiffii") + 8
    name, c1, c2, k, s, g, act, w_off, b_off = struct.unpack_from("<64siiiiiiQQ", blob, hdr)
    assert name.rstrip(b"\0") == b"model.0" and (c1, c2, k, s, g, act) == (3, 16, 3, 2, 1, 1)
    w = np.frombuffer(blob, "<f4", c2 * c1 * k * k, w_off).reshape(c2, c1, k, k)
    assert np.array_equal(w, net.convs["model.0"].w.numpy())
    assert b_off == w_off + w.nbytes


def test_decode_known_answer():
    """all-zero logits: uniform DFL -> every side 7.5 cells, angle = (0.5 - 0.25) * pi, class scores 0.5"""
    head = torch.zeros((1, 16 + 4 + 1, 77))
    pred = pp.decode(head, 32, 32, 12)  # [1,17,21]
    assert pred.shape == (1, 17, 21)
    p = pred[0, :, 0]  # first P3 anchor at (0.5, 0.5), stride 8
    assert float(p[0]) == pytest.approx(4.0, abs=1e-4) and float(p[1]) == pytest.approx(4.0, abs=1e-4)
    assert float(p[2]) == pytest.approx(15 * 8, abs=1e-3) and float(p[3]) == pytest.approx(15 * 8, abs=1e-3)
    assert float(p[4]) == 0.5 and float(p[16]) == pytest.approx(math.pi / 4, abs=1e-6)
    assert float(pred[0, 2, 20]) == pytest.approx(15 * 32, abs=1e-3)  # P5 anchor uses stride 32


def test_fast_nms_c_oracle_equals_torch_restatement():
    for n in (0, 1, 50, 700):
        _, cls, conf, xywhr = synth.make_dets(9 + n, n, extent=300.0) if n else (None, np.zeros(0, np.int32), np.zeros(0), np.zeros((0, 5), np.float32))
        boxes = xywhr.copy()
        boxes[:, :2] += cls[:, None].astype(np.float32) * np.float32(7680)
        scores = conf.astype(np.float32)
        order, keep = og.fast_nms(boxes, scores, 0.7)
        if n == 0:
            assert len(order) == 0
            continue
        tk = pp.nms_rotated(torch.tensor(boxes), torch.tensor(scores), 0.7).numpy()
        assert np.array_equal(order[keep.astype(bool)], tk)  # same survivors in the same (score) order
    # Fast-NMS is not greedy: a suppressed box still suppresses (A kills B, B kills C although A does not touch C)
    b = np.array([[100, 100, 60, 20, 0.0], [112, 100, 60, 20, 0.0], [124, 100, 60, 20, 0.0]], np.float32)
    assert og.probiou(b[0], b[1]) >= 0.7 > og.probiou(b[0], b[2])
    order, keep = og.fast_nms(b, np.array([0.9, 0.8, 0.7], np.float32), 0.7)
    assert list(keep) == [1, 0, 0]


def test_nms_class_offset_and_max_det():
    pred = torch.zeros((1, 17, 400))
    pred[0, 0] = 200.0; pred[0, 1] = 200.0; pred[0, 2] = 50.0; pred[0, 3] = 20.0  # 400 identical boxes
    pred[0, 4 + 3, :200] = torch.linspace(0.9, 0.5, 200)  # class 3
    pred[0, 4 + 7, 200:] = torch.linspace(0.8, 0.3, 200)  # class 7
    out = pp.non_max_suppression(pred, 0.25, 0.7, 300, 12)[0]
    assert out.shape == (2, 7)  # one survivor per class: the 7680-px class offset keeps classes apart
    assert out[0, 5] == 3 and out[1, 5] == 7 and float(out[0, 4]) == pytest.approx(0.9)


@pytest.mark.parametrize("name", ["Test1", "Test2"])
def test_result_conventions_hold_on_reference_goldens(name):
    """Ultralytics geometry conventions checked against every row of the reference's own outputs:
    recover xywhr from the golden corners, rebuild the corners with the restated xywhr2xyxyxyxy -> identical boxes."""
    names, boxes, conf, angle = load_xlsx_csv(name)
    p = boxes.reshape(-1, 4, 2)
    ctr = p.mean(1)
    v1 = (p[:, 0] - p[:, 3]) / 2 + (p[:, 1] - p[:, 2]) / 2  # = 2*vec1 /2 ... p1 - p4 = 2 vec2? use definitions below
    w = np.linalg.norm(p[:, 0] - p[:, 3], axis=1)  # |p1 - p4| = |2 vec1| = w
    h = np.linalg.norm(p[:, 0] - p[:, 1], axis=1)  # |p1 - p2| = |2 vec2| = h
    vec1 = (p[:, 0] - p[:, 3]) / 2
    theta = np.arctan2(vec1[:, 1], vec1[:, 0])
    assert np.all(theta > -1e-6) and np.all(theta < math.pi / 2 + 1e-6)  # regularize_rboxes range [0, pi/2)
    rebuilt = pp.xywhr2xyxyxyxy(torch.tensor(np.stack([ctr[:, 0] % 1000, ctr[:, 1] % 1000, w, h, theta], 1), dtype=torch.float64))
    off = (ctr - np.stack([ctr[:, 0] % 1000, ctr[:, 1] % 1000], 1))[:, None, :]
    assert np.max(np.abs(rebuilt.numpy() + off - p)) < 5e-4
    assert (w < h).any() and (w > h).any()  # no "w >= h" convention (older Ultralytics) in effect


def test_letterbox_matches_survey_appendix_c():
    shapes = {(416, 416, 416): (416, 416), (416, 263, 416): (416, 288), (175, 416, 416): (192, 416), (175, 263, 416): (288, 416),
              (416, 108, 416): (416, 128), (396, 416, 416): (416, 416), (80, 416, 416): (96, 416), (80, 108, 416): (320, 416),
              (128, 13, 128): (128, 32), (23, 128, 128): (32, 128), (23, 13, 128): (128, 96), (48, 76, 128): (96, 128)}
    for (h, w, imgsz), out in shapes.items():
        p = pp.letterbox_params(h, w, imgsz)
        assert (p["out_h"], p["out_w"]) == out, ((h, w), p)
    img = np.random.default_rng(0).integers(0, 256, (175, 263, 3), dtype=np.uint8)
    lb, p = pp.letterbox(img, 416)
    assert lb.shape == (288, 416, 3) and p["resize"]
    assert (lb[: p["top"]] == 114).all() and (lb[p["top"] + p["new_h"]:] == 114).all()
    same, p2 = pp.letterbox(np.ascontiguousarray(img[:, :208]), 416)
    up = pp.resize_bilinear_u8(np.full((4, 4, 3), 77, np.uint8), 9, 7)
    assert (up == 77).all()  # constant images stay constant under the fixed-point bilinear resize
