"""Tensor-level bindings of the C-ABI: torch tensors supply device memory and the current HIP stream, nothing else.

Every function here launches hand-written HIP kernels from libobbhip.so; inputs must be CUDA(=HIP) tensors.
"""
import ctypes as C

import torch

from . import _lib

_ctx = {}


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


# ---------------------------------------------------------------- S2 polygon IoU

def poly_iou_pairs(a, b):
    """a, b: [M,8] float64 -> [M] float64   (compute_polygon_iou per row, Detect_OBB.py:144)"""
    a = _chk(a, torch.float64, "a").reshape(-1, 8)
    b = _chk(b, torch.float64, "b").reshape(-1, 8)
    if a.shape != b.shape:
        raise ValueError("poly_iou_pairs: shape mismatch")
    out = torch.empty(a.shape[0], dtype=torch.float64, device=a.device)
    _call("obb_poly_iou_pairs", ctx(a.device), _p(a), _p(b), a.shape[0], _p(out), _stream())
    return out


def poly_iou_matrix(a, b, cls_a=None, cls_b=None):
    a = _chk(a, torch.float64, "a").reshape(-1, 8)
    b = _chk(b, torch.float64, "b").reshape(-1, 8)
    if cls_a is not None:
        _chk(cls_a, torch.int32, "cls_a")
        _chk(cls_b, torch.int32, "cls_b")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=a.device)
    _call("obb_poly_iou_matrix", ctx(a.device), _p(a), _p(cls_a), a.shape[0], _p(b), _p(cls_b), b.shape[0], _p(out), _stream())
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
    _call("obb_build_multich", ctx(t.device), _p(t), t.shape[0], t.shape[1], t.shape[2], _p(out), _stream())
    return out


def points_in_quads(pts, quads, cls_p=None, cls_q=None):
    """[np, nq] uint8: point i strictly inside the valid quad j (same class when both class vectors are given)"""
    pts = _chk(pts, torch.float64, "pts").reshape(-1, 2)
    quads = _chk(quads, torch.float64, "quads").reshape(-1, 8)
    if cls_p is not None:
        _chk(cls_p, torch.int32, "cls_p")
        _chk(cls_q, torch.int32, "cls_q")
    out = torch.zeros((pts.shape[0], quads.shape[0]), dtype=torch.uint8, device=pts.device)
    _call("obb_points_in_quads", ctx(pts.device), _p(pts), _p(cls_p), pts.shape[0], _p(quads), _p(cls_q), quads.shape[0], _p(out), _stream())
    return out


# ---------------------------------------------------------------- S3 merge_detections

def sort_desc_stable(key):
    key = _chk(key, torch.float64, "key")
    order = torch.empty(key.shape[0], dtype=torch.int32, device=key.device)
    _call("obb_sort_desc_stable", ctx(key.device), _p(key), key.shape[0], _p(order), _stream())
    return order


def nms_mask(boxes_sorted, cls_sorted, thr):
    b = _chk(boxes_sorted, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls_sorted, torch.int32, "cls")
    n = b.shape[0]
    w = (n + 63) // 64
    mask = torch.zeros((w, max(n, 1)), dtype=torch.int64, device=b.device)
    _call("obb_nms_mask", ctx(b.device), _p(b), _p(c), n, float(thr), _p(mask), _stream())
    return mask


def nms_reduce(mask, n):
    keep = torch.zeros(n, dtype=torch.uint8, device=mask.device)
    nk = torch.zeros(1, dtype=torch.int32, device=mask.device)
    _call("obb_nms_reduce", ctx(mask.device), _p(mask), n, _p(keep), _p(nk), _stream())
    return keep, nk


def merge_detections(boxes, cls, conf, thr):
    """-> (order int32[n]: sorted position -> input row, keep uint8[n] in sorted order, n_keep int32[1])"""
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    nk = torch.zeros(1, dtype=torch.int32, device=b.device)
    _call("obb_merge_detections", ctx(b.device), _p(b), _p(c), _p(s), n, float(thr), _p(order), _p(keep), _p(nk), _stream())
    return order, keep, nk


def merge_segments(boxes, cls, conf, seg_off, thr):
    """Batched per-tile merge (Detect_OBB.py:264).  seg_off int32[nseg+1] device.  Segments must be <= 512 rows."""
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    so = _chk(seg_off, torch.int32, "seg_off")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    _call("obb_merge_segments", ctx(b.device), _p(b), _p(c), _p(s), _p(so), so.shape[0] - 1, n, float(thr), _p(order), _p(keep),
          _stream())
    return order, keep


# ---------------------------------------------------------------- S4 consensus

def consensus(boxes, cls, conf, offsets, iou_partner=0.40, cons_low=0.25, cons_high=0.70):
    b = _chk(boxes, torch.float64, "boxes").reshape(-1, 8)
    c = _chk(cls, torch.int32, "cls")
    s = _chk(conf, torch.float64, "conf")
    off = (C.c_int64 * len(offsets))(*[int(o) for o in offsets])
    out = torch.empty(max(1, b.shape[0]), dtype=torch.int32, device=b.device)
    nout = torch.zeros(1, dtype=torch.int32, device=b.device)
    _call("obb_consensus", ctx(b.device), _p(b), _p(c), _p(s), off, len(offsets) - 1, float(iou_partner), float(cons_low),
          float(cons_high), _p(out), _p(nout), _stream())
    return out, nout


# ---------------------------------------------------------------- S5 detect_symbols pieces

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
    _call("obb_tile_postprocess", ctx(lp.device), _p(lp), _p(c), _p(dt), n, _p(r), r.shape[0], int(margin), int(strike_cls),
          _p(gb), _p(ang), _p(ins), _stream())
    return gb, ang, ins


# ---------------------------------------------------------------- S1 model

PRECISIONS = {"f16": 16, "fp16": 16, "bf16": 1016, "f32": 32, "fp32": 32}


def select_model(slot, device=None):
    """Several models can live in one context (dual-scale); choose which one load/forward/decode address."""
    _call("obb_set_option", ctx(device), b"model_slot", int(slot))


def model_load(blob, device=None, precision="f16", fuse=False, tail=True):
    """blob: bytes of an "OBBW" weight blob (host), loaded into the active model slot.  precision: 16-bit storage type.
    fuse=True runs the 104x104 C3k2 block and the class / angle branches of the head as LDS-resident layer chains (fused.hip)
    instead of one kernel per layer (the default, which is currently faster and keeps every activation observable).
    tail=False keeps the final 1x1 conv of each head branch a separate launch (default: fused behind its producer, whose own
    output then never reaches HBM)."""
    c = ctx(device)
    _call("obb_set_option", c, b"precision", PRECISIONS[precision])
    _call("obb_set_option", c, b"fuse", 1 if fuse else 0)
    _call("obb_set_option", c, b"tail", 1 if tail else 0)
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    _call("obb_model_load", c, buf, len(blob))


def model_info(h, w, device=None):
    c = ctx(device)
    nc, ch, a, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _call("obb_model_info", c, h, w, C.byref(nc), C.byref(ch), C.byref(a), C.byref(n))
    return {"nc": nc.value, "ch": ch.value, "anchors": a.value, "nconv": n.value}


def forward(tiles, out=None):
    """tiles uint8 [B,h,w,ch] NHWC (BGR for 3-channel input, as the reference passes crops) -> raw head [B,A,NO] f32.
    `out`: optional preallocated head tensor (stable input/output addresses let the library replay its captured hipGraph)."""
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
    _call("obb_forward", ctx(t.device), _p(t), B, h, w, _p(head), _stream())
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
    _call("obb_debug_activation", c, h, w, B, name.encode(), _p(out), out.numel(), C.byref(n), shp, _stream())
    return out


def decode(head, h, w):
    """raw head [B,A,64+nc+1] -> predictions [B,A,4+nc+1] (x,y,w,h, class scores, theta); anchor-major."""
    hd = _padded_head(_chk(head, torch.float32, "head"))
    B, A, _ = hd.shape
    pred = torch.empty((B, A, 4 + model_info(h, w, hd.device)["nc"] + 1), dtype=torch.float32, device=hd.device)
    _call("obb_decode", ctx(hd.device), _p(hd), B, h, w, _p(pred), _stream())
    return pred


def decode_nms(head, h, w, conf=0.25, iou=0.7, max_det=300):
    """-> (det [B,max_det,7] rows (x,y,w,h,conf,cls,theta) in score order, count int32[B])"""
    hd = _padded_head(_chk(head, torch.float32, "head"))
    B = hd.shape[0]
    det = torch.zeros((B, max_det, 7), dtype=torch.float32, device=hd.device)
    count = torch.zeros(B, dtype=torch.int32, device=hd.device)
    _call("obb_decode_nms", ctx(hd.device), _p(hd), B, h, w, float(conf), float(iou), int(max_det), _p(det), _p(count), _stream())
    return det, count


def probiou_nms(boxes, scores, iou):
    b = _chk(boxes, torch.float32, "boxes").reshape(-1, 5)
    s = _chk(scores, torch.float32, "scores")
    n = b.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=b.device)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    _call("obb_probiou_nms", ctx(b.device), _p(b), _p(s), n, float(iou), _p(order), _p(keep), _stream())
    return order, keep


def results(det, lb=None):
    """det [n,7] rows from decode_nms -> (xywhr [n,5] regularised + un-letterboxed, corners [n,8])"""
    d = _chk(det, torch.float32, "det").reshape(-1, 7)
    n = d.shape[0]
    if lb is not None:
        lb = _chk(lb, torch.float32, "lb").reshape(-1, 3)
    xywhr = torch.empty((n, 5), dtype=torch.float32, device=d.device)
    pts = torch.empty((n, 8), dtype=torch.float32, device=d.device)
    _call("obb_results", ctx(d.device), _p(d), _p(lb), n, _p(xywhr), _p(pts), _stream())
    return xywhr, pts


def gather_tiles(image, rects, tile):
    """image uint8 [H,W,C] (device), rects int32 [n,4] (device, all tile x tile) -> uint8 [n,tile,tile,C]"""
    img = _chk(image, torch.uint8, "image")
    r = _chk(rects, torch.int32, "rects").reshape(-1, 4)
    H, W, Cc = img.shape
    out = torch.empty((r.shape[0], tile, tile, Cc), dtype=torch.uint8, device=img.device)
    _call("obb_gather_tiles", ctx(img.device), _p(img), H, W, Cc, _p(r), r.shape[0], tile, _p(out), _stream())
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
    _call("obb_letterbox", ctx(img.device), _p(img), H, W, Cc, int(x), int(y), int(x2), int(y2), int(imgsz), _p(out), p["out_h"],
          p["out_w"], _stream())
    return out, p
