"""GPU parity of decode / ProbIoU Fast-NMS / result construction / tiler crops vs the torch-CPU oracle (oracle/postproc.py)."""
import numpy as np
import pytest
import torch

from oracle import geom as og
from oracle import postproc as pp
from oracle.yolo11_obb import Yolo11OBB
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops as o
    return o


@pytest.fixture(scope="module")
def net(ops):
    m = Yolo11OBB("n", nc=12, ch=3, seed=0)
    ops.model_load(m.to_blob())
    return m


def _head(net, seed, B, h, w):
    x = np.random.default_rng(seed).integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    return net.forward_raw(x, "fp32")


@pytest.mark.parametrize("h,w", [(416, 416), (128, 128), (416, 288)])
def test_decode_matches_oracle(ops, net, h, w):
    head = _head(net, 1, 2, h, w)
    exp = pp.decode(head, h, w, 12).transpose(1, 2)  # [B,A,17]
    got = ops.decode(head.cuda(), h, w).cpu()
    d = (got - exp).abs()
    assert float(d[..., :4].max()) < 2e-3  # pixels (values up to ~450): fp32 exp/cos ulp differences only
    assert float(d[..., 4:16].max()) < 1e-6 and float(d[..., 16].max()) < 1e-6


@pytest.mark.parametrize("conf,B", [(0.25, 4), (0.001, 1)])
def test_decode_nms_matches_oracle(ops, net, conf, B):
    h = w = 416
    head = _head(net, 2, B, h, w)
    pred = ops.decode(head.cuda(), h, w).cpu()  # feed the oracle the GPU's own decode so only NMS is compared
    exp = pp.non_max_suppression(pred.transpose(1, 2).contiguous(), conf, 0.7, 300, 12)
    det, count = ops.decode_nms(head.cuda(), h, w, conf, 0.7, 300)
    det, count = det.cpu(), count.cpu()
    for b in range(B):
        n = int(count[b])
        assert n == exp[b].shape[0], (b, n, exp[b].shape)
        assert n > 0
        assert torch.equal(det[b, :n], exp[b])  # same rows, same order, bit-identical values (pure selection)
    if conf == 0.001:
        assert int(count[0]) == 300  # max_det reached in metrics mode


@pytest.mark.parametrize("conf,max_det", [(0.25, 300), (0.05, 300), (0.001, 300), (0.25, 5), (0.999999, 300)])
def test_candidate_first_path_equals_the_full_decode(ops, net, conf, max_det):
    """obb_decode_nms (conf filter on the class logits -> decode + NMS of the survivors only; tiles above 256 candidates finished by the
    row-parallel kernels) against obb_decode_nms_full (decode every anchor, one LDS-resident workgroup per tile): identical rows and
    counts, bit for bit, on tiles with a handful of candidates, with hundreds, and with every anchor a candidate."""
    h = w = 416
    B = 12
    head = _head(net, 5, B, h, w).cuda()
    head[3, :, 64:76] -= 6.0    # a tile with few candidates
    head[4, :, 64:76] -= 30.0   # and one with none
    head[5, :, 64:76] += 3.0    # a saturated tile: (almost) every anchor passes
    det, cnt = ops.decode_nms(head, h, w, conf, 0.7, max_det)
    det_f, cnt_f = ops.decode_nms(head, h, w, conf, 0.7, max_det, full=True)
    assert torch.equal(cnt, cnt_f), (cnt.tolist(), cnt_f.tolist())
    for b in range(B):
        n = int(cnt[b])
        assert torch.equal(det[b, :n], det_f[b, :n]), b
    c = cnt.cpu().numpy()
    print("conf", conf, "rows per tile", c.tolist())
    if conf == 0.25 and max_det == 300:
        pred = ops.decode(head, h, w)
        ncand = (pred[..., 4:16].amax(-1) > conf).sum(1).cpu().numpy()
        print("candidates per tile", ncand.tolist())
        assert ncand.min() == 0 and ncand.max() > 2048 and ((ncand > 0) & (ncand <= 256)).any() and ((ncand > 256) & (ncand < 2048)).any()


@pytest.mark.parametrize("h,w,B", [(416, 416, 150), (640, 640, 6), (128, 128, 70)])
def test_flagged_tiles_in_both_heavy_forms_equal_the_full_decode(ops, net, h, w, B):
    """Tiles above 256 candidates: the LDS-resident sort-free form (k_heavy_prep + k_heavy_rows: a 416-px tile's 3549 candidates fit; 150 tiles = block
    columns that walk several flagged tiles, shared by 1..32 workgroups each) and the three-kernel form that larger inputs fall back to
    (640 px: 8400 anchors) against obb_decode_nms_full, bit for bit, on random heads with 0 .. every anchor a candidate."""
    A = ops.model_info(h, w)["anchors"]
    g = torch.Generator(device="cuda").manual_seed(h + B)
    head = torch.randn((B, A, 80), generator=g, device="cuda")
    head[..., :64] *= 2.0
    shift = torch.linspace(-6.0, 3.0, B, device="cuda")[torch.randperm(B, generator=g, device="cuda")]
    head[..., 64:76] += shift[:, None, None]
    head[..., 76] *= 0.5
    for conf, max_det in ((0.25, 300), (0.6, 40)):
        det, cnt = ops.decode_nms(head, h, w, conf, 0.7, max_det)
        det_f, cnt_f = ops.decode_nms(head, h, w, conf, 0.7, max_det, full=True)
        assert torch.equal(cnt, cnt_f), (cnt.tolist(), cnt_f.tolist())
        for b in range(B):
            n = int(cnt[b])
            assert torch.equal(det[b, :n], det_f[b, :n]), (conf, b)
        ncand = (torch.sigmoid(head[..., 64:76]).amax(-1) > conf).sum(1).cpu().numpy()
        print(h, w, "conf", conf, "flagged tiles", int((ncand > 256).sum()), "largest", int(ncand.max()), "rows", int(cnt.sum()))
        if conf == 0.25:
            assert (ncand > 256).sum() >= (3 if A > 1000 else 1) and ncand.min() < 256


@pytest.mark.parametrize("precision,h,w,B", [("f32", 416, 416, 6), ("f32", 128, 128, 40), ("f32", 416, 288, 3), ("f16", 416, 416, 4)])
def test_forward_gate_emits_the_class_maximum_and_decode_nms_gate_is_identical(ops, net, precision, h, w, B):
    """obb_forward_gate leaves cmax[b, a] = max over the class logits of head[b, a] (written by the fused class tails of the fp32 head -- DWConv
    prologue + 1x1 + tail at the levels of >= 112 pixels, the plain tail form on the small maps -- or by the extra pass of plans without them:
    the 16-bit modes), EXACTLY the maximum of the stored logits; obb_decode_nms_gate, which gates the candidates on that dense tensor,
    returns the rows of obb_decode_nms bit for bit -- also at a confidence where hundreds of anchors pass."""
    import numpy as np
    ops.model_load(net.to_blob(), precision=precision)
    x = torch.as_tensor(np.random.default_rng(h + B).integers(0, 256, (B, h, w, 3), dtype=np.uint8)).cuda()
    A = ops.model_info(h, w)["anchors"]
    cmax = torch.full((B, A), float("nan"), device="cuda")
    head = ops.forward(x, cmax=cmax)
    assert torch.equal(head, ops.forward(x))                                  # the head itself is unchanged
    assert torch.equal(cmax, head[..., 64:76].amax(-1)), float((cmax - head[..., 64:76].amax(-1)).abs().max())
    for conf in (0.25, 0.001):
        det, cnt = ops.decode_nms(head, h, w, conf, 0.7, 300)
        det_g, cnt_g = ops.decode_nms(head, h, w, conf, 0.7, 300, cmax=cmax)
        assert torch.equal(cnt, cnt_g) and int(cnt.sum()) > 0
        for b in range(B):
            assert torch.equal(det[b, :int(cnt[b])], det_g[b, :int(cnt[b])]), (conf, b)
    ops.model_load(net.to_blob())  # (the module's default again)


def test_probiou_nms_keep_mask_vs_oracle(ops):
    rng = np.random.default_rng(0)
    for n in (1, 7, 300, 2000):
        _, cls, conf, xywhr = synth.make_dets(40 + n, n, extent=500.0)
        boxes = xywhr.copy()
        boxes[:, :2] += cls[:, None].astype(np.float32) * np.float32(7680)
        scores = conf.astype(np.float32)
        eo, ek = og.fast_nms(boxes, scores, 0.7)
        order, keep = ops.probiou_nms(torch.tensor(boxes).cuda(), torch.tensor(scores).cuda(), 0.7)
        assert np.array_equal(order.cpu().numpy(), eo)
        got = keep.cpu().numpy()
        if not np.array_equal(got, ek):  # logf/expf differ by ulps between libm and the device: only exact-threshold ties may flip
            bad = np.nonzero(got != ek)[0]
            for r in bad:
                ious = [og.probiou(boxes[eo[i]], boxes[eo[r]]) for i in range(r)]
                assert min(abs(v - 0.7) for v in ious) < 1e-5, (n, r)
        tk, tkeep = pp.nms_rotated(torch.tensor(boxes), torch.tensor(scores), 0.7), None
        assert np.array_equal(np.sort(eo[ek.astype(bool)]), np.sort(tk.numpy()))  # C oracle == torch restatement


def test_results_match_oracle(ops):
    rng = np.random.default_rng(3)
    n = 500
    det = np.zeros((n, 7), np.float32)
    det[:, 0:2] = rng.uniform(0, 416, (n, 2))
    det[:, 2:4] = rng.uniform(5, 120, (n, 2))
    det[:, 4] = rng.uniform(0.25, 1, n)
    det[:, 5] = rng.integers(0, 12, n)
    det[:, 6] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)
    det[:10, 6] = [0.0, np.pi / 2, np.float32(np.pi / 2), -np.pi / 4, 3 * np.pi / 4 - 1e-4, 1e-7, -1e-7, np.pi / 4, 1.5707, 1.5709]
    obb, corners = pp.construct_result(torch.tensor(det), (416, 416), (416, 416))
    xywhr, pts = ops.results(torch.tensor(det).cuda())
    assert float((xywhr.cpu() - obb[:, :5]).abs().max()) < 1e-5
    assert float((pts.cpu() - corners.reshape(n, 8)).abs().max()) < 1e-3
    assert float(xywhr[:, 4].min()) >= 0.0 and float(xywhr[:, 4].max()) < np.pi / 2 + 1e-6
    # letterboxed crop (175 x 263 -> gain 1.58, pad): un-letterbox parameters per row
    p = ops.letterbox_shape(175, 263, 416)
    assert (p["out_h"], p["out_w"]) == (288, 416)
    obb2, corners2 = pp.construct_result(torch.tensor(det), (p["out_h"], p["out_w"]), (175, 263))
    lb = torch.tensor([[p["gain"], p["pad_x"], p["pad_y"]]], dtype=torch.float32).repeat(n, 1).cuda()
    xywhr2, pts2 = ops.results(torch.tensor(det).cuda(), lb)
    assert float((xywhr2.cpu() - obb2[:, :5]).abs().max()) < 1e-4
    assert float((pts2.cpu() - corners2.reshape(n, 8)).abs().max()) < 1e-3


def test_gather_tiles_and_letterbox(ops):
    img = np.random.default_rng(5).integers(0, 256, (807, 895, 3), dtype=np.uint8)
    rects = ops.tile_grid(807, 895, 416, 100)
    full = np.array([r for r in rects if r[2] - r[0] == 416 and r[3] - r[1] == 416], np.int32)
    assert len(full) == 4
    dimg = torch.tensor(img).cuda()
    tiles = ops.gather_tiles(dimg, torch.tensor(full).cuda(), 416).cpu().numpy()
    for t, (x, y, x2, y2) in enumerate(full):
        assert np.array_equal(tiles[t], img[y:y2, x:x2])
    shapes = {}
    for (x, y, x2, y2) in rects:
        out, p = ops.letterbox(dimg, x, y, x2, y2, 416)
        exp, ep = pp.letterbox(img[y:y2, x:x2], 416)
        shapes[(int(y2 - y), int(x2 - x))] = tuple(out.shape[:2])
        assert tuple(out.shape[:2]) == (ep["out_h"], ep["out_w"])
        assert np.array_equal(out.cpu().numpy(), exp), (x, y)
    assert shapes == {(416, 416): (416, 416), (416, 263): (416, 288), (175, 416): (192, 416), (175, 263): (288, 416)}  # Appendix C
