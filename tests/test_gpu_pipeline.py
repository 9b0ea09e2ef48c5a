"""GPU parity of the assembled hot path (detect_symbols / process_image) behind the reference's own call surface."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pipeline as opl
from oracle.yolo11_obb import Yolo11OBB
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ood():
    assert torch.cuda.is_available()
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import detect, model, ops

    class NS:
        pass
    ns = NS()
    ns.detect, ns.model, ns.ops = detect, model, ops
    return ns


@pytest.fixture(scope="module")
def nets(ood):
    n416 = Yolo11OBB("n", nc=12, ch=3, seed=0)
    n128 = Yolo11OBB("n", nc=12, ch=3, seed=1)
    return {"n416": n416, "n128": n128, "m416": ood.model.YOLO(n416, imgsz=416), "m128": ood.model.YOLO(n128, imgsz=128)}


def test_reference_surface_with_foreign_model_matches_goldens(ood):
    """detect_symbols driven by a duck-typed Ultralytics-like model reproduces the reference's own outputs exactly."""
    cases = json.load(open(os.path.join(GOLDEN, "detect_symbols_cases.json")))
    for c in cases:
        img = synth.coord_image(c["H"], c["W"])
        dets = ood.detect.detect_symbols(img, synth.StubModel(c["seed"]), c["tile"], c["overlap"])
        exp = c["dets"]
        assert len(dets) == len(exp)
        for d, e in zip(dets, exp):
            assert list(d[:10]) == e[:10]
            assert abs(d[10] - e[10]) <= 1e-9


def test_list_api_matches_reference_goldens(ood, ref_vectors):
    rv = ref_vectors
    dets = synth.dets_to_tuples(rv["merge3_boxes"], rv["merge3_cls"], rv["merge3_conf"])
    kept = ood.detect.merge_detections(dets, float(rv["merge3_thr"]))
    assert [int(d[10]) for d in dets] == list(rv["merge3_sorted"])  # in-place sort side effect (Detect_OBB.py:183)
    assert [int(d[10]) for d in kept] == list(rv["merge3_kept"])
    assert ood.detect.merge_detections([], 0.4) == []
    ci = 3
    d1 = synth.dets_to_tuples(rv[f"cons{ci}_b1"].reshape(-1, 8), rv[f"cons{ci}_c1"], rv[f"cons{ci}_s1"], 0)
    d2 = synth.dets_to_tuples(rv[f"cons{ci}_b2"].reshape(-1, 8), rv[f"cons{ci}_c2"], rv[f"cons{ci}_s2"], 100000)
    kept = ood.detect.cross_scale_consensus_filter({128: d1, 416: d2})
    assert [int(d[10]) for d in kept] == list(rv[f"cons{ci}_kept"])
    assert ood.detect.cross_scale_consensus_filter({}) == []
    sq = [0, 0, 1, 0, 1, 1, 0, 1]
    assert ood.detect.compute_polygon_iou(sq, [0.5, 0, 1.5, 0, 1.5, 1, 0.5, 1]) == pytest.approx(1 / 3, abs=1e-15)
    assert ood.detect.compute_polygon_iou(sq, [0, 0, 1, 1, 1, 0, 0, 1]) == 0.0


def _compare_dets(got, exp, tol_px=2e-3):
    assert len(got) == len(exp), (len(got), len(exp))
    for g, e in zip(got, exp):
        assert g[8] == e[8]
        assert abs(g[9] - e[9]) < 1e-6  # sigmoid via device expf vs torch: <= 1 ulp apart
        assert max(abs(a - b) for a, b in zip(g[:8], e[:8])) < tol_px
        assert abs(g[10] - e[10]) < 1e-2


def test_yolo_call_surface(ood, nets):
    """model(crop, conf=...)[0].obb: iteration yields 1-row items with .xyxyxyxy / .cls / .conf like Detect_OBB.py:228-231."""
    rng = np.random.default_rng(0)
    crop = rng.integers(0, 256, (175, 263, 3), dtype=np.uint8)  # partial corner tile: resize + pad path
    res = nets["m416"](crop, conf=0.25)
    assert len(res) == 1
    head_fn = lambda t: ood.ops.forward(torch.as_tensor(t).cuda()).cpu()[..., :77]
    om = opl.OracleModel(nets["n416"], 416, head_fn=head_fn)
    pts, cls, cf = om.predict_rows(crop, 0.25)
    assert len(res[0].obb) == len(cls) and len(cls) > 0
    for k, det in enumerate(res[0].obb):
        p = [float(v) for v in det.xyxyxyxy[0].flatten().tolist()]
        assert int(det.cls[0]) == int(cls[k]) and abs(float(det.conf[0]) - float(cf[k])) < 1e-6
        assert max(abs(a - b) for a, b in zip(p, pts[k])) < 2e-3
    with pytest.raises(ValueError):
        nets["m416"](np.zeros((10, 10, 4), np.uint8))


@pytest.mark.parametrize("H,W,ts,ov,key", [(807, 895, 416, 100, "416"), (300, 500, 128, 30, "128")])
def test_detect_symbols_vs_oracle_pipeline(ood, nets, H, W, ts, ov, key):
    """Batched device detect_symbols == the tile-by-tile CPU restatement when both see the same network outputs."""
    img = np.random.default_rng(7).integers(0, 256, (H, W, 3), dtype=np.uint8)
    model, net = nets["m" + key], nets["n" + key]

    def head_fn(t):
        model._ensure_active()
        return ood.ops.forward(torch.as_tensor(t).cuda()).cpu()[..., :77]
    exp = opl.detect_symbols(img, opl.OracleModel(net, ts, head_fn=head_fn), ts, ov)
    got = ood.detect.detect_symbols(img, model, ts, ov)
    assert len(exp) > 5
    _compare_dets(got, exp)


def test_four_channel_checkpoint_end_to_end(ood):
    """BASELINE configs[3] (4-channel input): a 3-channel BGR image goes in, every crop gets its DT-edge channel on the device
    (build_multich, Detect_OBB.py:76-77 / :87-133) before the letterbox; == the CPU restatement fed with the same network outputs."""
    net = Yolo11OBB("n", nc=12, ch=4, seed=2)
    model = ood.model.YOLO(net, imgsz=416)
    assert model.ch == 4
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (600, 740, 3), dtype=np.uint8)
    img[100:130, 50:600] = 15
    img[200:520, 300:330] = (240, 10, 10)

    def head_fn(t):
        model._ensure_active()
        return ood.ops.forward(torch.as_tensor(t).cuda()).cpu()[..., :77]
    om = opl.OracleModel(net, 416, head_fn=head_fn)
    exp = opl.detect_symbols(img, om, 416, 100)
    got = ood.detect.detect_symbols(img, model, 416, 100)      # 4 tiles: one full, three partial (letterboxed after the channel is built)
    assert len(exp) > 5
    _compare_dets(got, exp)
    crop = np.ascontiguousarray(img[0:175, 0:263])
    res = model(crop, conf=0.25)                               # the per-crop call surface builds the channel too
    pts, cls, cf = om.predict_rows(crop, 0.25)
    assert len(res[0].obb) == len(cls)
    four = ood.detect.build_multich(crop, 4)
    from oracle import dtedge
    assert four.shape == (175, 263, 4) and np.abs(four.astype(np.int32) - dtedge.build_multich(crop, 4).astype(np.int32)).max() <= 1
    assert np.array_equal(ood.detect.build_multich(crop, 3), crop)


def test_process_image_dual_scale_vs_oracle(ood, nets):
    img = np.random.default_rng(9).integers(0, 256, (500, 640, 3), dtype=np.uint8)

    def hf(model):
        def f(t):
            model._ensure_active()
            return ood.ops.forward(torch.as_tensor(t).cuda()).cpu()[..., :77]
        return f
    oms = [opl.OracleModel(nets["n128"], 128, head_fn=hf(nets["m128"])), opl.OracleModel(nets["n416"], 416, head_fn=hf(nets["m416"]))]
    exp, exp_by_scale = opl.process_image(img, oms)
    got = ood.detect.process_image(img, None, [nets["m128"], nets["m416"]])
    assert sum(len(v) for v in exp_by_scale.values()) > 20
    _compare_dets(got, exp)
    cfg = ood.detect.Config(tile_sizes=(416,), overlaps=(100,))
    got1 = ood.detect.process_image(img, None, [nets["m416"]], cfg)
    exp1, _ = opl.process_image(img, oms[1:], (416,), (100,))
    _compare_dets(got1, exp1)


def test_process_image_on_the_real_sample_file(ood, nets, tmp_path):
    """BASELINE configs[0] plumbing: process_image(path, output_dir, ...) on the reference's own Input/Test1.png (committed fixture): image
    decode -> 9 + 90 tiles -> ... -> CSV rows; equals the CPU pipeline fed with the same network outputs; the I/O shell then writes the
    overlay JPEG and the xlsx table of Detect_OBB.py:295-330."""
    import os
    from conftest import GOLDEN
    from oriented_object_detection_amd import io_shell
    path = os.path.join(GOLDEN, "input", "Test1.png")
    img = ood.detect.imread_bgr(path)
    assert img.shape == (807, 895, 3)

    def hf(model):
        def f(t):
            model._ensure_active()
            return ood.ops.forward(torch.as_tensor(t).cuda()).cpu()[..., :77]
        return f
    oms = [opl.OracleModel(nets["n128"], 128, head_fn=hf(nets["m128"])), opl.OracleModel(nets["n416"], 416, head_fn=hf(nets["m416"]))]
    exp, by_scale = opl.process_image(img, oms)
    got = ood.detect.process_image(path, str(tmp_path), [nets["m128"], nets["m416"]])
    assert len(exp) > 3
    _compare_dets(got, exp)
    rows = [l.rstrip("\n").split(",") for l in open(tmp_path / "Test1.csv")]
    assert rows[0] == io_shell.XLSX_COLUMNS and len(rows) == len(got) + 1
    jpg, xlsx = io_shell.save_outputs(img, path, str(tmp_path), got, ood.detect.DEFAULT.CLASS_NAMES)
    assert os.path.getsize(jpg) > 10000
    back = io_shell.read_xlsx(xlsx)
    assert len(back) == len(got) + 1 and [float(v) for v in back[1][1:9]] == [float(v) for v in got[0][:8]]
    # the script's main loop over an input directory (Detect_OBB.py:745-755)
    os.makedirs(tmp_path / "in")
    os.symlink(path, tmp_path / "in" / "Test1.png")
    done = io_shell.main(str(tmp_path / "in"), str(tmp_path / "out"), [nets["m128"], nets["m416"]])
    assert len(done) == 1 and os.path.exists(tmp_path / "out" / "Test1.xlsx") and os.path.exists(tmp_path / "out" / "Test1_detected.jpg")


# ---------------------------------------------------------------------------------------------- device-side compaction (no torch glue)
def _records_equal(a, b):
    return len(a) == len(b) and torch.equal(a.tile, b.tile) and torch.equal(a.cls, b.cls) and torch.equal(a.conf, b.conf) and torch.equal(a.pts, b.pts)


@pytest.mark.parametrize("conf,md,lbox", [(0.25, 300, False), (0.02, 300, True), (0.25, 40, False), (0.001, 300, False), (1.0, 300, False)])
def test_tile_survivors_equals_the_glued_kernels(ood, nets, conf, md, lbox):
    """obb_tile_survivors (result construction + per-detection body + per-tile merge + compaction, counts kept on the device) against
    the stand-alone kernels glued by host-side compactions (obb_results -> obb_tile_postprocess -> obb_merge_segments): the same
    records, bit for bit, in the same order -- dense tiles (conf 0.001: saturated at max_det), sparse and empty ones, with and
    without a letterbox, max_det below the candidate count."""
    D, ops = ood.detect, ood.ops
    m = nets["m416"]
    rng = np.random.default_rng(5)
    B = 9
    tiles = torch.as_tensor(rng.integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
    tiles[3] = 0  # a tile without candidates at the usual thresholds
    rects = np.array([[(t % 4) * 316, (t // 4) * 316, (t % 4) * 316 + 416, (t // 4) * 316 + 416] for t in range(12)], np.int32)
    rects_dev = torch.as_tensor(rects).cuda()
    tile_ids = torch.tensor([0, 2, 3, 5, 6, 7, 9, 10, 11], dtype=torch.int32).cuda()
    lb = torch.tensor([[0.8, 3.0, 11.0]] * B, dtype=torch.float32).cuda() if lbox else None
    cfg = D.Config(max_det=md)
    outs = []
    for fused in (False, True):
        D.USE_TILE_SURVIVORS = fused
        try:
            outs.append(D.predict_tile_records(m, tiles, rects_dev, tile_ids, lb, cfg, 416, conf))
        finally:
            D.USE_TILE_SURVIVORS = True
    glue, fus = outs
    assert fus.packed is not None and glue.packed is None
    assert _records_equal(glue, fus), (len(glue), len(fus))
    if conf >= 1.0:  # nothing passes the confidence filter: every tile empty
        assert len(fus) == 0
    elif conf <= 0.25:
        assert len(fus) > 0
    # the consumer side: packed rows -> detections == column form -> detections
    a, b = D.records_to_detset(fus, rects_dev, cfg, 416), D.records_to_detset(glue, rects_dev, cfg, 416)
    assert len(a) == len(b)
    if len(a):
        assert torch.equal(a.boxes, b.boxes) and torch.equal(a.cls, b.cls) and torch.equal(a.conf, b.conf) and torch.equal(a.angle, b.angle)


def test_select_kept_and_lazy_counts(ood, ref_vectors):
    """merge_detections_device returns the kept rows through obb_select_kept (ordered compaction, count on the device): identical to
    indexing with the boolean mask on the host side, for the reference-generated merge cases and a 20 000-row set"""
    D, ops = ood.detect, ood.ops
    rv = ref_vectors
    rng = np.random.default_rng(11)
    cases = [(rv["merge3_boxes"], rv["merge3_cls"], rv["merge3_conf"], float(rv["merge3_thr"]))]
    n = 20000
    ctr = rng.uniform(0, 4000, (n, 2)); wh = rng.uniform(8, 60, (n, 2)); th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    corners = np.stack([ctr + np.stack([sx * wh[:, 0] / 2 * c - sy * wh[:, 1] / 2 * s, sx * wh[:, 0] / 2 * s + sy * wh[:, 1] / 2 * c], 1)
                        for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1))], 1).reshape(n, 8)
    cases.append((corners, rng.integers(0, 3, n), np.round(rng.uniform(0.25, 1, n), 2), 0.4))
    for boxes, cls, conf, thr in cases:
        ds = D.DetSet(torch.as_tensor(np.ascontiguousarray(boxes, np.float64)).cuda(), torch.as_tensor(np.asarray(cls, np.int32)).cuda(),
                      torch.as_tensor(np.asarray(conf, np.float64)).cuda(), torch.zeros(len(cls), dtype=torch.float64).cuda())
        order, keep, nk = ops.merge_detections(ds.boxes, ds.cls, ds.conf, thr)
        exp = ds.select(order[keep.bool()])
        got, order2 = D.merge_detections_device(ds, thr)
        assert got.count is not None and got._n is None  # nothing read yet
        assert len(got) == int(nk.item()) == len(exp) and torch.equal(order2, order)
        assert torch.equal(got.boxes, exp.boxes) and torch.equal(got.cls, exp.cls) and torch.equal(got.conf, exp.conf) and torch.equal(got.angle, exp.angle)


def test_gather_compact(ood):
    """the fixed-capacity all-gather buffer of 5 'ranks' (one empty, one over capacity) -> dense rows in rank order + counts"""
    ops = ood.ops
    rng = np.random.default_rng(2)
    world, cap = 5, 37
    counts = [12, 0, 37, 60, 5]  # rank 3 sent a count above the capacity: its first `cap` rows arrive, the count tells the caller to repeat
    recv = torch.as_tensor(rng.integers(-2**31, 2**31 - 1, (world, cap + 1, 12), dtype=np.int64).astype(np.int32))
    for r, c in enumerate(counts):
        recv[r, 0, 0] = c
    rows, cnt = ops.gather_compact(recv.cuda())
    cnt = cnt.cpu().tolist()
    exp = torch.cat([recv[r, 1:1 + min(c, cap)] for r, c in enumerate(counts)], 0)
    assert cnt[:world] == counts and cnt[world] == exp.shape[0]
    assert torch.equal(rows[:cnt[world]].cpu(), exp)
