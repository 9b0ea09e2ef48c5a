"""PyTorch-ROCm custom ops over the C-ABI of libobbhip.so (SURVEY.md section 8(b) "Registration").

Every device entry point of include/obbhip.h is registered with the torch dispatcher as `torch.ops.obbhip.<name>` through
`torch.library.custom_op` (device type "cuda" = HIP on ROCm: there is no CPU implementation, a CPU tensor raises).  The registered
form takes the caller's tensors only -- inputs, and outputs as pre-allocated tensors it mutates -- and launches the hand-written HIP
kernels on torch's CURRENT stream, so the ops compose with torch streams, events and `torch.cuda.graph` capture.  PyTorch supplies
device memory and streams, nothing else; no Triton, no multi-backend dispatch.

The plain functions below (`forward`, `decode_nms`, `merge_detections`, ...) are the convenience layer the rest of the package
uses: they allocate the outputs and call `torch.ops.obbhip.*`.
"""
import ctypes as C
import os
from typing import List, Optional

import torch

from . import _lib

_ctx = {}
T = torch.Tensor


def ctx(device=None):
    """One library context per device (created lazily)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if idx not in _ctx:
        h = C.c_void_p()
        _lib.check(_lib.lib().obb_ctx_create(idx, C.byref(h)))
        _ctx[idx] = h
    return _ctx[idx]


def destroy_contexts():
    for h in _ctx.values():
        _lib.lib().obb_ctx_destroy(h)
    _ctx.clear()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/HIP tensor (there is no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _call(name, c, *args):
    _lib.check(getattr(_lib.lib(), name)(c, *args), c)


def _op(name, mutates):
    return torch.library.custom_op(f"obbhip::{name}", mutates_args=mutates, device_types="cuda")


# ================================================================ registered ops (torch.ops.obbhip.*)
# ---------------------------------------------------------------- S2 polygon IoU (Detect_OBB.py:144-154)
@_op("poly_iou_pairs", ("out",))
def _poly_iou_pairs(a: T, b: T, out: T) -> None:
    _call("obb_poly_iou_pairs", ctx(a.device), _p(a), _p(b), a.shape[0], _p(out), _stream())


@_op("poly_iou_matrix", ("out",))
def _poly_iou_matrix(a: T, cls_a: Optional[T], b: T, cls_b: Optional[T], out: T) -> None:
    _call("obb_poly_iou_matrix", ctx(a.device), _p(a), _p(cls_a), a.shape[0], _p(b), _p(cls_b), b.shape[0], _p(out), _stream())


@_op("points_in_quads", ("out",))
def _points_in_quads(pts: T, cls_p: Optional[T], quads: T, cls_q: Optional[T], out: T) -> None:
    _call("obb_points_in_quads", ctx(pts.device), _p(pts), _p(cls_p), pts.shape[0], _p(quads), _p(cls_q), quads.shape[0], _p(out), _stream())


@_op("build_multich", ("out",))
def _build_multich(bgr: T, out: T) -> None:
    _call("obb_build_multich", ctx(bgr.device), _p(bgr), bgr.shape[0], bgr.shape[1], bgr.shape[2], _p(out), _stream())


# ---------------------------------------------------------------- S3 merge_detections (Detect_OBB.py:176-200)
@_op("sort_desc_stable", ("order",))
def _sort_desc_stable(key: T, order: T) -> None:
    _call("obb_sort_desc_stable", ctx(key.device), _p(key), key.shape[0], _p(order), _stream())


@_op("nms_mask", ("mask",))
def _nms_mask(boxes: T, cls: T, thr: float, mask: T) -> None:
    _call("obb_nms_mask", ctx(boxes.device), _p(boxes), _p(cls), boxes.shape[0], float(thr), _p(mask), _stream())


@_op("nms_reduce", ("keep", "n_keep"))
def _nms_reduce(mask: T, n: int, keep: T, n_keep: T) -> None:
    _call("obb_nms_reduce", ctx(mask.device), _p(mask), n, _p(keep), _p(n_keep), _stream())


@_op("merge_detections", ("order", "keep", "n_keep"))
def _merge_detections(boxes: T, cls: T, conf: T, thr: float, order: T, keep: T, n_keep: T) -> None:
    _call("obb_merge_detections", ctx(boxes.device), _p(boxes), _p(cls), _p(conf), boxes.shape[0], float(thr), _p(order), _p(keep), _p(n_keep), _stream())


@_op("merge_segments", ("order", "keep"))
def _merge_segments(boxes: T, cls: T, conf: T, seg_off: T, thr: float, order: T, keep: T) -> None:
    _call("obb_merge_segments", ctx(boxes.device), _p(boxes), _p(cls), _p(conf), _p(seg_off), seg_off.shape[0] - 1, boxes.shape[0], float(thr), _p(order),
          _p(keep), _stream())


# ---------------------------------------------------------------- S4 consensus (Detect_OBB.py:347-423)
@_op("consensus", ("out_idx", "n_out"))
def _consensus(boxes: T, cls: T, conf: T, offsets: List[int], iou_partner: float, cons_low: float, cons_high: float, out_idx: T, n_out: T) -> None:
    off = (C.c_int64 * len(offsets))(*[int(o) for o in offsets])
    _call("obb_consensus", ctx(boxes.device), _p(boxes), _p(cls), _p(conf), off, len(offsets) - 1, float(iou_partner), float(cons_low), float(cons_high),
          _p(out_idx), _p(n_out), _stream())


# ---------------------------------------------------------------- S5 detect_symbols pieces (Detect_OBB.py:202-266)
@_op("tile_postprocess", ("gboxes", "angle", "inside"))
def _tile_postprocess(local_pts: T, cls: T, det_tile: T, rects: T, margin: int, strike_cls: int, gboxes: T, angle: T, inside: T) -> None:
    _call("obb_tile_postprocess", ctx(local_pts.device), _p(local_pts), _p(cls), _p(det_tile), local_pts.shape[0], _p(rects), rects.shape[0], int(margin),
          int(strike_cls), _p(gboxes), _p(angle), _p(inside), _stream())


@_op("tile_survivors", ("records", "tile_off", "n_records"))
def _tile_survivors(det: T, count: T, lb: Optional[T], tile_ids: T, rects: T, margin: int, strike_cls: int, iou_thr: float, records: T, tile_off: T,
                    n_records: T) -> None:
    _call("obb_tile_survivors", ctx(det.device), _p(det), _p(count), det.shape[0], det.shape[1], _p(lb), _p(tile_ids), _p(rects), int(margin), int(strike_cls),
          float(iou_thr), _p(records), _p(tile_off), _p(n_records), _stream())


@_op("select_kept", ("out_boxes", "out_cls", "out_conf", "out_angle", "n_out"))
def _select_kept(order: T, keep: T, boxes: T, cls: T, conf: T, angle: T, out_boxes: T, out_cls: T, out_conf: T, out_angle: T, n_out: T) -> None:
    _call("obb_select_kept", ctx(order.device), _p(order), _p(keep), order.shape[0], _p(boxes), _p(cls), _p(conf), _p(angle), _p(out_boxes), _p(out_cls),
          _p(out_conf), _p(out_angle), _p(n_out), _stream())


@_op("records_to_dets", ("gboxes", "cls", "conf", "angle"))
def _records_to_dets(records: T, n: int, rects: T, strike_cls: int, gboxes: T, cls: T, conf: T, angle: T) -> None:
    _call("obb_records_to_dets", ctx(records.device), _p(records), int(n), _p(rects), int(strike_cls), _p(gboxes), _p(cls), _p(conf), _p(angle), _stream())


@_op("gather_compact", ("out", "counts"))
def _gather_compact(recv: T, out: T, counts: T) -> None:
    _call("obb_gather_compact", ctx(recv.device), _p(recv), recv.shape[0], recv.shape[1] - 1, _p(out), _p(counts), _stream())


@_op("gather_tiles", ("out",))
def _gather_tiles(image: T, rects: T, tile: int, out: T) -> None:
    H, W, Cc = image.shape
    _call("obb_gather_tiles", ctx(image.device), _p(image), H, W, Cc, _p(rects), rects.shape[0], tile, _p(out), _stream())


@_op("letterbox", ("out",))
def _letterbox(image: T, x: int, y: int, x2: int, y2: int, imgsz: int, out: T) -> None:
    H, W, Cc = image.shape
    _call("obb_letterbox", ctx(image.device), _p(image), H, W, Cc, int(x), int(y), int(x2), int(y2), int(imgsz), _p(out), out.shape[0], out.shape[1], _stream())


# ---------------------------------------------------------------- S1 model (Detect_OBB.py:26, 81-83, 228-231)
@_op("forward", ("head",))
def _forward(tiles: T, head: T) -> None:
    B, h, w, _ = tiles.shape
    _call("obb_forward", ctx(tiles.device), _p(tiles), B, h, w, _p(head), _stream())


@_op("forward_gate", ("head", "cmax"))
def _forward_gate(tiles: T, head: T, cmax: T) -> None:
    B, h, w, _ = tiles.shape
    _call("obb_forward_gate", ctx(tiles.device), _p(tiles), B, h, w, _p(head), _p(cmax), _stream())


@_op("decode", ("pred",))
def _decode(head: T, h: int, w: int, pred: T) -> None:
    _call("obb_decode", ctx(head.device), _p(head), head.shape[0], h, w, _p(pred), _stream())


@_op("decode_nms", ("det", "count"))
def _decode_nms(head: T, h: int, w: int, conf: float, iou: float, max_det: int, det: T, count: T) -> None:
    _call("obb_decode_nms", ctx(head.device), _p(head), head.shape[0], h, w, float(conf), float(iou), int(max_det), _p(det), _p(count), _stream())


@_op("decode_nms_gate", ("det", "count"))
def _decode_nms_gate(head: T, cmax: T, h: int, w: int, conf: float, iou: float, max_det: int, det: T, count: T) -> None:
    _call("obb_decode_nms_gate", ctx(head.device), _p(head), _p(cmax), head.shape[0], h, w, float(conf), float(iou), int(max_det), _p(det), _p(count), _stream())


@_op("decode_nms_full", ("det", "count"))
def _decode_nms_full(head: T, h: int, w: int, conf: float, iou: float, max_det: int, det: T, count: T) -> None:
    _call("obb_decode_nms_full", ctx(head.device), _p(head), head.shape[0], h, w, float(conf), float(iou), int(max_det), _p(det), _p(count), _stream())


@_op("probiou_nms", ("order", "keep"))
def _probiou_nms(boxes: T, scores: T, iou: float, order: T, keep: T) -> None:
    _call("obb_probiou_nms", ctx(boxes.device), _p(boxes), _p(scores), boxes.shape[0], float(iou), _p(order), _p(keep), _stream())


@_op("results", ("xywhr", "pts"))
def _results(det: T, lb: Optional[T], xywhr: T, pts: T) -> None:
    _call("obb_results", ctx(det.device), _p(det), _p(lb), det.shape[0], _p(xywhr), _p(pts), _stream())


@_op("probiou_loss", ("loss", "grad_pred"))
def _probiou_loss(pred: T, target: T, weight: Optional[T], target_scores_sum: float, loss: T, grad_pred: T) -> None:
    _call("obb_probiou_loss", ctx(pred.device), _p(pred), _p(target), _p(weight), pred.shape[0], float(target_scores_sum), _p(loss), _p(grad_pred), _stream())


@_op("tile_labels", ("mask", "out"))
def _tile_labels(labels: T, rects: T, min_fraction: float, mask: T, out: T) -> None:
    _call("obb_tile_labels", ctx(labels.device), _p(labels), labels.shape[0], _p(rects), rects.shape[0], float(min_fraction), _p(mask), _p(out), _stream())


@_op("rotated_tal_assign", ("target_labels", "target_bboxes", "target_scores", "fg_mask", "target_gt_idx"))
def _rotated_tal_assign(pd_scores: T, pd_bboxes: T, anc_points: T, gt_labels: T, gt_bboxes: T, mask_gt: T, topk: int, alpha: float, beta: float,
                        target_labels: T, target_bboxes: T, target_scores: T, fg_mask: T, target_gt_idx: T) -> None:
    bs, na, nc = pd_scores.shape
    _call("obb_rotated_tal_assign", ctx(pd_scores.device), _p(pd_scores), _p(pd_bboxes), _p(anc_points), _p(gt_labels), _p(gt_bboxes), _p(mask_gt), bs, na, nc,
          gt_bboxes.shape[1], int(topk), float(alpha), float(beta), _p(target_labels), _p(target_bboxes), _p(target_scores), _p(fg_mask), _p(target_gt_idx), _stream())


@_op("dfl_loss", ("loss", "grad_pred"))
def _dfl_loss(pred_dist: T, target_ltrb: T, weight: Optional[T], reg_max: int, target_scores_sum: float, loss: T, grad_pred: T) -> None:
    _call("obb_dfl_loss", ctx(pred_dist.device), _p(pred_dist), _p(target_ltrb), _p(weight), target_ltrb.shape[0], int(reg_max), float(target_scores_sum),
          _p(loss), _p(grad_pred), _stream())


@_op("bce_loss", ("loss", "grad_logits"))
def _bce_loss(logits: T, targets: T, target_scores_sum: float, loss: T, grad_logits: T) -> None:
    _call("obb_bce_loss", ctx(logits.device), _p(logits), _p(targets), logits.numel(), float(target_scores_sum), _p(loss), _p(grad_logits), _stream())


@_op("sgd_step", ("param", "momentum_buf"))
def _sgd_step(param: T, grad: T, momentum_buf: T, lr: float, momentum: float, weight_decay: float, nesterov: bool, first_step: bool) -> None:
    _call("obb_sgd_step", ctx(param.device), _p(param), _p(grad), _p(momentum_buf), param.numel(), float(lr), float(momentum), float(weight_decay),
          int(bool(nesterov)), int(bool(first_step)), _stream())


@_op("adamw_step", ("param", "exp_avg", "exp_avg_sq"))
def _adamw_step(param: T, grad: T, exp_avg: T, exp_avg_sq: T, step: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float) -> None:
    _call("obb_adamw_step", ctx(param.device), _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), int(step), float(lr), float(beta1), float(beta2),
          float(eps), float(weight_decay), _stream())


@_op("debug_activation", ("out",))
def _debug_activation(name: str, B: int, h: int, w: int, out: T) -> None:
    n = C.c_int64(0)
    shp = (C.c_int32 * 3)()
    _call("obb_debug_activation", ctx(out.device), h, w, B, name.encode(), _p(out), out.numel(), C.byref(n), shp, _stream())


_O = torch.ops.obbhip


# ================================================================ convenience layer (allocates the outputs)
def poly_iou_pairs(a, b):
    """a, b: [M,8] float64 -> [M] float64   (compute_polygon_iou per row, Detect_OBB.py:144)"""
    a = _chk(a, torch.float64, "a").reshape(-1, 8)
    b = _chk(b, torch.float64, "b").reshape(-1, 8)
    if a.shape != b.shape:
        raise ValueError("poly_iou_pairs: shape mismatch")
    out = torch.empty(a.shape[0], dtype=torch.float64, device=a.device)
    _O.poly_iou_pairs(a, b, out)
    return out


def poly_iou_matrix(a, b, cls_a=None, cls_b=None):
    a = _chk(a, torch.float64, "a").reshape(-1, 8)
    b = _chk(b, torch.float64, "b").reshape(-1, 8)
    if cls_a is not None:
        _chk(cls_a, torch.int32, "cls_a")
        _chk(cls_b, torch.int32, "cls_b")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=a.device)
    _O.poly_iou_matrix(a, cls_a, b, cls_b, out)
    return out


def build_multich(bgr_tiles, out=None):
    """uint8 [B,h,w,3] BGR crops -> uint8 [B,h,w,4] = RGB + distance-transform edge channel (Detect_OBB.py:87-133)"""
    t = _chk(bgr_tiles, torch.uint8, "bgr_tiles")
    assert t.dim() == 4 and t.shape[3] == 3
    if out is None:
        out = torch.empty((t.shape[0], t.shape[1], t.shape[2], 4), dtype=torch.uint8, device=t.device)
    else:
        _chk(out, torch.uint8, "out")
        assert tuple(out.shape) == (t.shape[0], t.shape[1], t.shape[2], 4)
    _O.build_multich(t, out)
    return out


def points_in_quads(pts, quads, cls_p=None, cls_q=None):
    """[np, nq] uint8: point i strictly inside the valid quad j (same class when both class vectors are given)"""
    pts = _chk(pts, torch.float64, "pts").reshape(-1, 2)
    quads = _chk(quads, torch.float64, "quads").reshape(-1, 8)
    if cls_p is not None:
        _chk(cls_p, torch.int32, "cls_p")
        _chk(cls_q, torch.int32, "cls_q")
    out = torch.zeros((pts.shape[0], quads.shape[0]), dtype=torch.uint8, device=pts.device)
    _O.points_in_quads(pts, cls_p, quads, cls_q, out)
    return out


def sort_desc_stable(key):
    key = _chk(key, torch.float64, "key")
    order = torch.empty(key.shape[0], dtype=torch.int32, device=key.device)
    _O.sort_desc_stable(key, order)
    return order


def nms_mask(boxes_sorted, cls_sorted, thr):
    b = _chk(boxes_sorted, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls_sorted, torch.int32, "cls")
    n = b.shape[0]
    w = (n + 63) // 64
    mask = torch.zeros((w, max(n, 1)), dtype=torch.int64, device=b.device)
    _O.nms_mask(b, c, float(thr), mask)
    return mask


def nms_reduce(mask, n):
    keep = torch.zeros(n, dtype=torch.uint8, device=mask.device)
    nk = torch.zeros(1, dtype=torch.int32, device=mask.device)
    _O.nms_reduce(mask, n, keep, nk)
    return keep, nk


def merge_detections(boxes, cls, conf, thr):
    """-> (order int32[n]: sorted position -> input row, keep uint8[n] in sorted order, n_keep int32[1])"""
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.empty(n, dtype=torch.uint8, device=b.device)  # (every position is written by the library)
    nk = torch.empty(1, dtype=torch.int32, device=b.device)
    _O.merge_detections(b, c, s, float(thr), order, keep, nk)
    return order, keep, nk


def merge_segments(boxes, cls, conf, seg_off, thr):
    """Batched per-tile merge (Detect_OBB.py:264).  seg_off int32[nseg+1] device.  Segments of any length (the library takes long
    ones through its dense path)."""
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    so = _chk(seg_off, torch.int32, "seg_off")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    _O.merge_segments(b, c, s, so, float(thr), order, keep)
    return order, keep


def consensus(boxes, cls, conf, offsets, iou_partner=0.40, cons_low=0.25, cons_high=0.70):
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    out = torch.empty(max(1, b.shape[0]), dtype=torch.int32, device=b.device)
    nout = torch.zeros(1, dtype=torch.int32, device=b.device)
    _O.consensus(b, c, s, [int(o) for o in offsets], float(iou_partner), float(cons_low), float(cons_high), out, nout)
    return out, nout


def tile_grid(H, W, tile, overlap):
    """Host helper -> int32 numpy [ntiles,4] (x, y, x2, y2) in reference visiting order."""
    import numpy as np
    n = C.c_int64(0)
    _lib.check(_lib.lib().obb_tile_grid(H, W, tile, overlap, None, 0, C.byref(n)))
    rects = np.zeros((max(1, n.value), 4), np.int32)
    _lib.check(_lib.lib().obb_tile_grid(H, W, tile, overlap, rects.ctypes.data_as(_lib.c_ip), n.value, C.byref(n)))
    return rects[:n.value]


def tile_postprocess(local_pts, cls, det_tile, rects, margin, strike_cls=1):
    lp = _chk(local_pts, torch.float32, "local_pts").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    dt = _chk(det_tile, torch.int32, "det_tile")
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    n = lp.shape[0]
    gb = torch.empty((n, 8), dtype=torch.float64, device=lp.device)
    ang = torch.empty(n, dtype=torch.float64, device=lp.device)
    ins = torch.empty(n, dtype=torch.uint8, device=lp.device)
    _O.tile_postprocess(lp, c, dt, r, int(margin), int(strike_cls), gb, ang, ins)
    return gb, ang, ins


def rotated_tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=10, alpha=0.5, beta=6.0):
    """RotatedTaskAlignedAssigner.forward -> (target_labels int32[bs,na], target_bboxes f32[bs,na,5], target_scores f32[bs,na,nc], fg_mask bool[bs,na],
    target_gt_idx int32[bs,na]); inputs as in ultralytics (pd_scores sigmoid probabilities [bs,na,nc], boxes xywhr in pixels, gt_labels [bs,n_max(,1)],
    mask_gt [bs,n_max(,1)])."""
    ps = _chk(pd_scores.float().contiguous(), torch.float32, "pd_scores")
    pb = _chk(pd_bboxes.float().contiguous(), torch.float32, "pd_bboxes")
    ap = _chk(anc_points.float().contiguous(), torch.float32, "anc_points")
    bs, na, nc = ps.shape
    gb = gt_bboxes.float().contiguous()
    n_max = gb.shape[1]
    gl = gt_labels.reshape(bs, n_max).to(torch.int32).contiguous()
    mg = mask_gt.reshape(bs, n_max).to(torch.uint8).contiguous()
    d = ps.device
    tl = torch.empty((bs, na), dtype=torch.int32, device=d)
    tb = torch.empty((bs, na, 5), dtype=torch.float32, device=d)
    ts = torch.empty((bs, na, nc), dtype=torch.float32, device=d)
    fg = torch.empty((bs, na), dtype=torch.uint8, device=d)
    ti = torch.empty((bs, na), dtype=torch.int32, device=d)
    _O.rotated_tal_assign(ps, pb, ap, gl, gb, mg, int(topk), float(alpha), float(beta), tl, tb, ts, fg, ti)
    return tl, tb, ts, fg.bool(), ti


def conv_dgrad_bf16(dy, weight):
    """dy bf16 [B,H,W,cout] (device, NHWC), weight fp32 [cout,cin,k,k] (any device; repacked on the host per call) -> dx bf16 [B,H,W,cin]: the
    input gradient of the stride-1 `same` convolution, on the forward's bf16 MFMA kernel with flipped / transposed weights."""
    d = _chk(dy, torch.bfloat16, "dy")
    B, H, W, cout = d.shape
    w = weight.detach().float().cpu().contiguous()
    assert w.shape[0] == cout and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3)
    cin, ks = int(w.shape[1]), int(w.shape[2])
    dx = torch.empty((B, H, W, cin), dtype=torch.bfloat16, device=d.device)
    _call("obb_conv_dgrad_bf16", ctx(d.device), _p(d), w.numpy().ctypes.data_as(_lib.c_fp), B, H, W, cin, cout, ks, _p(dx), _stream())
    return dx


def conv_wgrad_bf16(x, dy, ks):
    """x bf16 [B,H,W,cin], dy bf16 [B,H,W,cout] (NHWC, device) -> dw fp32 [cout,cin,ks,ks]: the weight gradient of the stride-1 `same`
    convolution, fp32 accumulation, deterministic."""
    xx, d = _chk(x, torch.bfloat16, "x"), _chk(dy, torch.bfloat16, "dy")
    B, H, W, cin = xx.shape
    cout = d.shape[3]
    dw = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=xx.device)
    _call("obb_conv_wgrad_bf16", ctx(xx.device), _p(xx), _p(d), B, H, W, cin, cout, int(ks), _p(dw), _stream())
    return dw


def conv_pack_bf16(w, H, W, dgrad_form=False):
    """fp32 OIHW master weights (device) -> the bf16 MFMA fragment order of the conv kernel for H x W maps, packed ON THE DEVICE (no host
    repack, no synchronisation); dgrad_form: the flipped / channel-transposed weights whose forward convolution is the input gradient."""
    ww = _chk(w, torch.float32, "w")
    cout, cin, ks, _ = ww.shape
    n = C.c_int64(0)
    _call("obb_conv_packed_elems", ctx(ww.device), cout, cin, ks, int(H), int(W), int(bool(dgrad_form)), C.byref(n))
    out = torch.empty(n.value, dtype=torch.bfloat16, device=ww.device)
    _call("obb_conv_pack_bf16", ctx(ww.device), _p(ww), cout, cin, ks, int(H), int(W), int(bool(dgrad_form)), _p(out), _stream())
    return out


def conv_fwd_bf16(x, packed, bias, cout, ks):
    """y = conv(x) + bias (stride 1, `same` padding, no activation): x bf16 [B,H,W,cin] NHWC, packed = conv_pack_bf16 of the [cout,cin,ks,ks]
    weights for this H x W, bias fp32 [cout] or None -> bf16 [B,H,W,cout]."""
    xx = _chk(x, torch.bfloat16, "x")
    B, H, W, cin = xx.shape
    y = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=xx.device)
    b = _chk(bias, torch.float32, "bias") if bias is not None else None
    _call("obb_conv_fwd_bf16", ctx(xx.device), _p(xx), _p(_chk(packed, torch.bfloat16, "packed")), _p(b), B, H, W, cin, int(cout), int(ks), _p(y), _stream())
    return y


def silu_bf16(z):
    zz = _chk(z, torch.bfloat16, "z")
    a = torch.empty_like(zz)
    _call("obb_silu_bf16", ctx(zz.device), _p(zz), _p(a), zz.numel(), _stream())
    return a


def silu_bwd_bf16(z, da):
    zz, d = _chk(z, torch.bfloat16, "z"), _chk(da, torch.bfloat16, "da")
    dz = torch.empty_like(zz)
    _call("obb_silu_bwd_bf16", ctx(zz.device), _p(zz), _p(d), _p(dz), zz.numel(), _stream())
    return dz


def bias_grad_bf16(dy, out=None):
    d = _chk(dy, torch.bfloat16, "dy")
    cout = d.shape[-1]
    db = out if out is not None else torch.empty(cout, dtype=torch.float32, device=d.device)
    _call("obb_bias_grad_bf16", ctx(d.device), _p(d), d.numel() // cout, cout, _p(_chk(db, torch.float32, "db")), _stream())
    return db


def tile_survivors(det, count, lb, tile_ids, rects, margin, iou_thr, strike_cls=1):
    """det float32[B, max_det, 7] + count int32[B] (decode_nms) -> (records int32[B * max_det, 12] (capacity), tile_off int32[B + 1],
    n_records int32[1]), all on the device: result construction, per-detection body (border filter `margin`), per-tile merge at `iou_thr`
    and compaction into exchange records in tile order, with no host-visible count in between."""
    d = _chk(det, torch.float32, "det")
    c = _chk(count, torch.int32, "count")
    ti = _chk(tile_ids, torch.int32, "tile_ids")
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    if lb is not None:
        lb = _chk(lb, torch.float32, "lb").reshape(-1, 3)
    B, md = d.shape[0], d.shape[1]
    rec = torch.empty((B * md, 12), dtype=torch.int32, device=d.device)
    off = torch.empty(B + 1, dtype=torch.int32, device=d.device)
    n = torch.empty(1, dtype=torch.int32, device=d.device)
    _O.tile_survivors(d, c, lb, ti, r, int(margin), int(strike_cls), float(iou_thr), rec, off, n)
    return rec, off, n


def select_kept(order, keep, boxes, cls, conf, angle):
    """rows order[i] of the sorted positions i with keep[i], in that order -> (boxes, cls, conf, angle) of capacity n and n_out int32[1] (device)"""
    n = order.shape[0]
    ob = torch.empty((n, 8), dtype=torch.float64, device=order.device)
    oc = torch.empty(n, dtype=torch.int32, device=order.device)
    of = torch.empty(n, dtype=torch.float64, device=order.device)
    oa = torch.empty(n, dtype=torch.float64, device=order.device)
    cnt = torch.empty(1, dtype=torch.int32, device=order.device)
    _O.select_kept(_chk(order, torch.int32, "order"), _chk(keep, torch.uint8, "keep"), _chk(boxes, torch.float64, "boxes"), _chk(cls, torch.int32, "cls"),
                   _chk(conf, torch.float64, "conf"), _chk(angle, torch.float64, "angle"), ob, oc, of, oa, cnt)
    return ob, oc, of, oa, cnt


def records_to_dets(records, n, rects, strike_cls=1):
    """the first n rows of packed records int32[*, 12] -> (global boxes float64[n, 8], cls int32[n], conf float64[n], angle float64[n])"""
    rec = _chk(records, torch.int32, "records")
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    n = int(n)
    gb = torch.empty((n, 8), dtype=torch.float64, device=rec.device)
    c = torch.empty(n, dtype=torch.int32, device=rec.device)
    f = torch.empty(n, dtype=torch.float64, device=rec.device)
    a = torch.empty(n, dtype=torch.float64, device=rec.device)
    if n:
        _O.records_to_dets(rec, n, r, int(strike_cls), gb, c, f, a)
    return gb, c, f, a


def gather_compact(recv):
    """recv int32[world, capacity + 1, 12] (all_gather of fixed-capacity record blocks, count in [w, 0, 0]) -> (rows int32[world * capacity, 12]
    dense in rank order, counts int32[world + 1]: the counts as sent, then the number of rows written)"""
    r = _chk(recv, torch.int32, "recv")
    world, cap = r.shape[0], r.shape[1] - 1
    out = torch.empty((world * cap, 12), dtype=torch.int32, device=r.device)
    counts = torch.empty(world + 1, dtype=torch.int32, device=r.device)
    _O.gather_compact(r, out, counts)
    return out, counts


PRECISIONS = {"f16": 16, "fp16": 16, "bf16": 1016, "f32": 32, "fp32": 32}
# engine switches of obb_set_option (each restores the separate launches of one fused form; used by the A/B parity tests)
MODEL_OPTIONS = ("fuse", "tail", "tail16", "bneck", "bneck_cv2", "c3kimg", "dwpw", "upfold", "stem", "front", "pair", "hmerge", "sppf_fuse", "attn_mfma", "xtile", "nitile", "nc2", "blk32", "c3k2f", "pw32", "graph")


def select_model(slot, device=None):
    """Several models can live in one context (dual-scale); choose which one load/forward/decode address."""
    _call("obb_set_option", ctx(device), b"model_slot", int(slot))


def model_load(blob, device=None, precision="f16", fuse=False, tail=True, **options):
    """blob: bytes of an "OBBW" weight blob (host), loaded into the active model slot.  precision: "f16" / "bf16" = 16-bit storage with
    fp32 accumulation, "f32" = fp32 arithmetic end to end (what the reference computes; one kernel per layer).
    tail=False keeps the final 1x1 conv of each head branch a separate launch and turns every other intermediate-swallowing fusion off
    (every layer observable).  Further keyword switches (MODEL_OPTIONS, all default on except `fuse`) disable single fused forms:
    tail16, bneck, bneck_cv2, c3kimg, dwpw, upfold, stem, front, pair, hmerge, sppf_fuse, attn_mfma, xtile, nitile, nc2, blk32, c3k2f, pw32."""
    c = ctx(device)
    _call("obb_set_option", c, b"precision", PRECISIONS[precision])
    _call("obb_set_option", c, b"fuse", 1 if fuse else 0)
    _call("obb_set_option", c, b"tail", 1 if tail else 0)
    for k in MODEL_OPTIONS[2:]:
        try:
            _call("obb_set_option", c, k.encode(), 1 if options.pop(k, True) else 0)
        except _lib.ObbHipError:
            if not os.environ.get("OBB_LIB"):  # (an older diagnostic build named by OBB_LIB may not know the newest switches: A/B timing only)
                raise
    if options:
        raise TypeError(f"model_load: unknown options {sorted(options)}")
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    _call("obb_model_load", c, buf, len(blob))


def model_unload(slot, device=None):
    """Frees the model of `slot` (weights, plans, activation slabs, captured graphs) after a device synchronisation."""
    _call("obb_model_unload", ctx(device), int(slot))


def model_info(h, w, device=None):
    c = ctx(device)
    nc, ch, a, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _call("obb_model_info", c, h, w, C.byref(nc), C.byref(ch), C.byref(a), C.byref(n))
    return {"nc": nc.value, "ch": ch.value, "anchors": a.value, "nconv": n.value}


def forward(tiles, out=None, cmax=None):
    """tiles uint8 [B,h,w,ch] NHWC (BGR for 3-channel input, as the reference passes crops) -> raw head [B,A,NO] f32.
    `out`: optional preallocated head tensor (stable input/output addresses let the library replay its captured hipGraph).
    `cmax`: optional float32 [B,A] that receives the largest class logit of every anchor (the dense candidate gate of
    decode_nms(..., cmax=...): obb_forward_gate)."""
    t = _chk(tiles, torch.uint8, "tiles")
    B, h, w, ch = t.shape
    info = model_info(h, w, t.device)
    if ch != info["ch"]:
        raise ValueError(f"forward: model expects {info['ch']} input channels, got {ch}")
    no = 64 + info["nc"] + 1
    shape = (B, info["anchors"], (no + 3) // 4 * 4)
    if out is not None:
        if tuple(out.shape) != shape:
            raise ValueError(f"forward: out must have shape {shape}")
        head = _chk(out, torch.float32, "out")
    else:
        head = torch.zeros(shape, dtype=torch.float32, device=t.device)
    if cmax is not None:
        cm = _chk(cmax, torch.float32, "cmax")
        if tuple(cm.shape) != shape[:2]:
            raise ValueError(f"forward: cmax must have shape {shape[:2]}")
        if B:
            _O.forward_gate(t, head, cm)
    elif B:
        _O.forward(t, head)
    return head  # rows padded to a multiple of 4 floats; [..., :64+nc+1] are the logits


def _padded_head(hd):
    """accept unpadded [B,A,64+nc+1] heads (e.g. from the oracle) by padding rows to the device layout"""
    no = hd.shape[-1]
    if no % 4 == 0:
        return hd
    out = torch.zeros(hd.shape[:-1] + ((no + 3) // 4 * 4,), dtype=hd.dtype, device=hd.device)
    out[..., :no] = hd
    return out


def debug_plan(h, w, device=None):
    """-> list of text lines, one per kernel launch of the lowered forward (layer, tiling, grid, LDS, MACs)"""
    c = ctx(device)
    need = C.c_int64(0)
    _call("obb_debug_plan", c, h, w, None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value)
    _call("obb_debug_plan", c, h, w, buf, need.value, C.byref(need))
    return buf.value.decode().strip().split("\n")


def debug_activation(name, B, h, w, device=None):
    c = ctx(device)
    n = C.c_int64(0)
    shp = (C.c_int32 * 3)()
    _call("obb_debug_activation", c, h, w, B, name.encode(), None, 0, C.byref(n), shp, _stream())
    out = torch.empty((B, shp[0], shp[1], shp[2]), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    _O.debug_activation(name, B, h, w, out)
    return out


def decode(head, h, w):
    """raw head [B,A,64+nc+1] -> predictions [B,A,4+nc+1] (x,y,w,h, class scores, theta); anchor-major."""
    hd = _padded_head(_chk(head, torch.float32, "head"))
    B, A, _ = hd.shape
    pred = torch.empty((B, A, 4 + model_info(h, w, hd.device)["nc"] + 1), dtype=torch.float32, device=hd.device)
    _O.decode(hd, h, w, pred)
    return pred


def decode_nms(head, h, w, conf=0.25, iou=0.7, max_det=300, full=False, zero=True, cmax=None):
    """-> (det [B,max_det,7] rows (x,y,w,h,conf,cls,theta) in score order, count int32[B]).  full=True runs the reference form (decode
    every anchor first) that the candidate-first default must reproduce bit for bit.  zero=False leaves the rows past count[b]
    uninitialised (no fill launch: what the device-side consumers, which only look at rows below the count, ask for)."""
    hd = _padded_head(_chk(head, torch.float32, "head"))
    B = hd.shape[0]
    det = (torch.zeros if zero or not B else torch.empty)((B, max_det, 7), dtype=torch.float32, device=hd.device)
    count = (torch.zeros if zero or not B else torch.empty)(B, dtype=torch.int32, device=hd.device)
    if B and cmax is not None and not full:
        cm = _chk(cmax, torch.float32, "cmax")
        if tuple(cm.shape) != tuple(hd.shape[:2]):
            raise ValueError("decode_nms: cmax must be [B, A] of the same head")
        _O.decode_nms_gate(hd, cm, h, w, float(conf), float(iou), int(max_det), det, count)
    elif B:
        (_O.decode_nms_full if full else _O.decode_nms)(hd, h, w, float(conf), float(iou), int(max_det), det, count)
    return det, count


def probiou_nms(boxes, scores, iou):
    b = _chk(boxes, torch.float32, "boxes").reshape(-1, 5)
    s = _chk(scores, torch.float32, "scores")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    _O.probiou_nms(b, s, float(iou), order, keep)
    return order, keep


def results(det, lb=None):
    """det [n,7] rows from decode_nms -> (xywhr [n,5] regularised + un-letterboxed, corners [n,8])"""
    d = _chk(det, torch.float32, "det").reshape(-1, 7)
    n = d.shape[0]
    if lb is not None:
        lb = _chk(lb, torch.float32, "lb").reshape(-1, 3)
    xywhr = torch.empty((n, 5), dtype=torch.float32, device=d.device)
    pts = torch.empty((n, 8), dtype=torch.float32, device=d.device)
    _O.results(d, lb, xywhr, pts)
    return xywhr, pts


def gather_tiles(image, rects, tile):
    """image uint8 [H,W,C] (device), rects int32 [n,4] (device, all tile x tile) -> uint8 [n,tile,tile,C]"""
    img = _chk(image, torch.uint8, "image")
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    H, W, Cc = img.shape
    out = torch.empty((r.shape[0], tile, tile, Cc), dtype=torch.uint8, device=img.device)
    _O.gather_tiles(img, r, int(tile), out)
    return out


def letterbox_shape(ch, cw, imgsz, stride=32):
    """Host arithmetic of LetterBox(auto=True): -> dict(out_h, out_w, gain, pad_x, pad_y)"""
    r = min(imgsz / ch, imgsz / cw)
    new_w, new_h = int(round(cw * r)), int(round(ch * r))
    dw, dh = ((imgsz - new_w) % stride) / 2, ((imgsz - new_h) % stride) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out_h, out_w = new_h + top + bottom, new_w + left + right
    # un-letterbox parameters exactly as scale_boxes computes them from the two shapes
    gain = min(out_h / ch, out_w / cw)
    pad_x = round((out_w - cw * gain) / 2 - 0.1)
    pad_y = round((out_h - ch * gain) / 2 - 0.1)
    return {"out_h": out_h, "out_w": out_w, "gain": gain, "pad_x": pad_x, "pad_y": pad_y}


def letterbox(image, x, y, x2, y2, imgsz):
    img = _chk(image, torch.uint8, "image")
    H, W, Cc = img.shape
    p = letterbox_shape(y2 - y, x2 - x, imgsz)
    out = torch.empty((p["out_h"], p["out_w"], Cc), dtype=torch.uint8, device=img.device)
    _O.letterbox(img, int(x), int(y), int(x2), int(y2), int(imgsz), out)
    return out, p


def tile_labels(labels, rects, min_fraction=0.1):
    """Training-set tiler, label assignment (Train_OBB.py:87-112): labels float64 [n,8] pixel corners, rects int32 [T,4] square tiles ->
    (mask uint8 [T,n], out float64 [T,n,8] = shifted / clipped / normalised corners where mask is set, zeros elsewhere)."""
    lab = _chk(labels, torch.float64, "labels").reshape(-1, 8)
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    mask = torch.zeros((r.shape[0], lab.shape[0]), dtype=torch.uint8, device=lab.device)
    out = torch.zeros((r.shape[0], lab.shape[0], 8), dtype=torch.float64, device=lab.device)
    if lab.shape[0] and r.shape[0]:
        _O.tile_labels(lab, r, float(min_fraction), mask, out)
    return mask, out


def dfl_loss(pred_dist, target_ltrb, weight=None, target_scores_sum=1.0, reg_max=16):
    """DFL term of the OBB loss for n matched anchors, forward + backward in one launch: pred_dist [n, 4*reg_max] (or [n*4, reg_max]) logits,
    target_ltrb [n,4] distances in bins, weight [n] or None -> (loss float32[1], grad [n, 4*reg_max]).  ultralytics DFLoss / RotatedBboxLoss."""
    t = _chk(target_ltrb, torch.float32, "target_ltrb").reshape(-1, 4)
    p = _chk(pred_dist, torch.float32, "pred_dist").reshape(t.shape[0], 4 * reg_max)
    if weight is not None:
        weight = _chk(weight, torch.float32, "weight").reshape(-1)
        if weight.shape[0] != t.shape[0]:
            raise ValueError("dfl_loss: weight length mismatch")
    loss = torch.zeros(1, dtype=torch.float32, device=p.device)
    grad = torch.empty_like(p)
    _O.dfl_loss(p, t, weight, int(reg_max), float(target_scores_sum), loss, grad)
    return loss, grad


def bce_loss(logits, targets, target_scores_sum=1.0):
    """Classification term of the OBB loss: BCEWithLogits(reduction none)(logits, targets).sum() / target_scores_sum, forward + backward in
    one launch: logits, targets float32 of equal shape -> (loss float32[1], grad like logits)."""
    x = _chk(logits, torch.float32, "logits")
    t = _chk(targets, torch.float32, "targets")
    if x.shape != t.shape:
        raise ValueError("bce_loss: shape mismatch")
    loss = torch.zeros(1, dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x)
    _O.bce_loss(x, t, float(target_scores_sum), loss, grad)
    return loss, grad


def sgd_step(param, grad, momentum_buf, lr, momentum=0.9, weight_decay=0.0, nesterov=True, first_step=False):
    """In place, one launch: torch.optim.SGD's update (dampening 0) of a flat fp32 parameter group; `first_step` initialises the momentum
    buffer with the (decayed) gradient as torch does."""
    p, g, b = _chk(param, torch.float32, "param"), _chk(grad, torch.float32, "grad"), _chk(momentum_buf, torch.float32, "momentum_buf")
    if not (p.numel() == g.numel() == b.numel()):
        raise ValueError("sgd_step: size mismatch")
    _O.sgd_step(p, g, b, float(lr), float(momentum), float(weight_decay), bool(nesterov), bool(first_step))


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """In place, one launch: torch.optim.AdamW's update of a flat fp32 parameter group; `step` counts from 1."""
    p, g = _chk(param, torch.float32, "param"), _chk(grad, torch.float32, "grad")
    m, v = _chk(exp_avg, torch.float32, "exp_avg"), _chk(exp_avg_sq, torch.float32, "exp_avg_sq")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adamw_step: size mismatch")
    _O.adamw_step(p, g, m, v, int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))


def probiou_loss(pred, target, weight=None, target_scores_sum=1.0):
    """ProbIoU rotated-box loss of n matched pairs, forward + backward in one launch: pred, target [n,5] (x,y,w,h,theta) float32,
    weight [n] or None -> (loss float32[1], grad_pred [n,5] = d loss / d pred).  Train_OBB.py:796-841 -> v8OBBLoss / RotatedBboxLoss."""
    p = _chk(pred, torch.float32, "pred").reshape(-1, 5)
    t = _chk(target, torch.float32, "target").reshape(-1, 5)
    if p.shape != t.shape:
        raise ValueError("probiou_loss: shape mismatch")
    if weight is not None:
        weight = _chk(weight, torch.float32, "weight").reshape(-1)
    loss = torch.zeros(1, dtype=torch.float32, device=p.device)
    grad = torch.empty_like(p)
    _O.probiou_loss(p, t, weight, float(target_scores_sum), loss, grad)
    return loss, grad
