"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/obb_oracle.c (the CPU restatement).

Mirrors the reference's Python call surface so parity tests read like calls into Detect_OBB.py:
  compute_polygon_iou (Detect_OBB.py:144), merge_detections (:176), cross_scale_consensus_filter (:347),
  center_inside_safe_region (:167), compute_angle_from_bbox (:135), tile grid of detect_symbols (:210-223),
  compute_pr_for_class/compute_ap_from_pr (:512/:489).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libobb_oracle.so")
    src = os.path.join(_HERE, "obb_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libobb_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip, bp, lp, fp = (C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                              C.POINTER(C.c_int64), C.POINTER(C.c_float))
        L.ora_poly_iou.restype = C.c_double
        L.ora_poly_iou.argtypes = [dp, dp]
        L.ora_poly_iou_pairs.restype = None
        L.ora_poly_iou_pairs.argtypes = [dp, dp, C.c_int64, dp]
        L.ora_sort_desc_stable.restype = None
        L.ora_sort_desc_stable.argtypes = [dp, C.c_int64, ip]
        L.ora_merge_detections.restype = C.c_int64
        L.ora_merge_detections.argtypes = [dp, ip, dp, C.c_int64, C.c_double, ip, bp]
        L.ora_consensus.restype = C.c_int64
        L.ora_consensus.argtypes = [dp, ip, dp, lp, C.c_int32, C.c_double, C.c_double, C.c_double, ip]
        L.ora_center_inside.restype = C.c_int
        L.ora_center_inside.argtypes = [dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
        L.ora_point_in_quad.restype = C.c_int
        L.ora_point_in_quad.argtypes = [dp, C.c_double, C.c_double]
        L.ora_strike_angle.restype = C.c_double
        L.ora_strike_angle.argtypes = [dp]
        L.ora_tile_grid.restype = C.c_int64
        L.ora_tile_grid.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, ip, C.c_int64]
        L.ora_ap_for_class.restype = C.c_double
        L.ora_ap_for_class.argtypes = [dp, dp, ip, C.c_int64, dp, ip, C.c_int64, C.c_double, bp, lp]
        L.ora_probiou.restype = C.c_float
        L.ora_probiou.argtypes = [fp, fp]
        L.ora_fast_nms.restype = C.c_int64
        L.ora_fast_nms.argtypes = [fp, fp, C.c_int64, C.c_float, ip, bp]
        _LIB = L
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def compute_polygon_iou(box1, box2):
    a, b = _d(box1[:8]), _d(box2[:8])
    return float(lib().ora_poly_iou(_p(a, C.c_double), _p(b, C.c_double)))


def poly_iou_pairs(a, b):
    a, b = _d(a).reshape(-1, 8), _d(b).reshape(-1, 8)
    out = np.empty(a.shape[0], np.float64)
    lib().ora_poly_iou_pairs(_p(a, C.c_double), _p(b, C.c_double), a.shape[0], _p(out, C.c_double))
    return out


def sort_desc_stable(key):
    key = _d(key)
    order = np.empty(key.shape[0], np.int32)
    lib().ora_sort_desc_stable(_p(key, C.c_double), key.shape[0], _p(order, C.c_int32))
    return order


def merge_arrays(boxes, cls, conf, thr):
    """-> (order[n] stable-desc permutation, keep[n] flags in sorted order)"""
    boxes, conf = _d(boxes).reshape(-1, 8), _d(conf)
    cls = np.ascontiguousarray(cls, np.int32)
    n = boxes.shape[0]
    order = np.empty(n, np.int32)
    keep = np.zeros(n, np.uint8)
    lib().ora_merge_detections(_p(boxes, C.c_double), _p(cls, C.c_int32), _p(conf, C.c_double), n, float(thr),
                               _p(order, C.c_int32), _p(keep, C.c_uint8))
    return order, keep


def merge_detections(detections, iou_threshold=0.5):
    """Reference-shaped: list of 11-tuples in, kept sub-list out (sorted by conf desc); sorts caller's list in place."""
    if not detections:
        return []
    arr = np.array([d[:10] for d in detections], np.float64)
    order, keep = merge_arrays(arr[:, :8], arr[:, 8].astype(np.int32), arr[:, 9], iou_threshold)
    srt = [detections[i] for i in order]
    detections[:] = srt
    return [d for d, k in zip(srt, keep) if k]


def consensus_arrays(boxes, cls, conf, offsets, iou_partner=0.40, cons_low=0.25, cons_high=0.70):
    boxes, conf = _d(boxes).reshape(-1, 8), _d(conf)
    cls = np.ascontiguousarray(cls, np.int32)
    off = np.ascontiguousarray(offsets, np.int64)
    out = np.empty(max(1, boxes.shape[0]), np.int32)
    n = lib().ora_consensus(_p(boxes, C.c_double), _p(cls, C.c_int32), _p(conf, C.c_double), _p(off, C.c_int64),
                            len(off) - 1, iou_partner, cons_low, cons_high, _p(out, C.c_int32))
    return out[:n].copy()


def cross_scale_consensus_filter(dets_by_scale):
    scales = sorted(dets_by_scale.keys())
    flat, off = [], [0]
    for s in scales:
        flat.extend(dets_by_scale[s])
        off.append(len(flat))
    if not flat:
        return []
    arr = np.array([d[:10] for d in flat], np.float64)
    idx = consensus_arrays(arr[:, :8], arr[:, 8].astype(np.int32), arr[:, 9], off)
    return [flat[i] for i in idx]


def center_inside_safe_region(points8, crop_x0, crop_y0, crop_w, crop_h, margin_px):
    p = _d(points8)
    return bool(lib().ora_center_inside(_p(p, C.c_double), crop_x0, crop_y0, crop_w, crop_h, margin_px))


def point_in_quad(pts8, x, y):
    """Polygon(pts).is_valid and Polygon(pts).contains(Point(x, y))  (Detect_OBB.py:631-634)"""
    a = _d(pts8)
    return bool(lib().ora_point_in_quad(_p(a, C.c_double), float(x), float(y)))


def compute_angle_from_bbox(points):
    p = _d(points)
    return float(lib().ora_strike_angle(_p(p, C.c_double)))


def tile_grid(H, W, tile, overlap):
    step = max(1, tile - overlap)
    cap = ((H + step - 1) // step) * ((W + step - 1) // step) + 1
    rects = np.zeros((cap, 4), np.int32)
    n = lib().ora_tile_grid(H, W, tile, overlap, _p(rects, C.c_int32), cap)
    return rects[:n].copy()


def ap_for_class(det_boxes, det_score, det_img, gt_boxes, gt_img, iou_thr):
    db, ds = _d(det_boxes).reshape(-1, 8), _d(det_score)
    di = np.ascontiguousarray(det_img, np.int32)
    gb = _d(gt_boxes).reshape(-1, 8)
    gi = np.ascontiguousarray(gt_img, np.int32)
    tp = np.zeros(max(1, db.shape[0]), np.uint8)
    tot = np.zeros(3, np.int64)
    ap = lib().ora_ap_for_class(_p(db, C.c_double), _p(ds, C.c_double), _p(di, C.c_int32), db.shape[0],
                                _p(gb, C.c_double), _p(gi, C.c_int32), gb.shape[0], float(iou_thr),
                                _p(tp, C.c_uint8), _p(tot, C.c_int64))
    return float(ap), tp[:db.shape[0]].copy(), tot


def probiou(o1, o2):
    a = np.ascontiguousarray(o1, np.float32)
    b = np.ascontiguousarray(o2, np.float32)
    return float(lib().ora_probiou(_p(a, C.c_float), _p(b, C.c_float)))


def fast_nms(boxes_xywhr, scores, thr):
    b = np.ascontiguousarray(boxes_xywhr, np.float32).reshape(-1, 5)
    s = np.ascontiguousarray(scores, np.float32)
    n = b.shape[0]
    order = np.empty(n, np.int32)
    keep = np.zeros(n, np.uint8)
    lib().ora_fast_nms(_p(b, C.c_float), _p(s, C.c_float), n, float(thr), _p(order, C.c_int32), _p(keep, C.c_uint8))
    return order, keep
