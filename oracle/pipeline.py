"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the whole Detect_OBB.py hot path (detect_symbols :202-266,
process_image :268-291) on top of the torch oracle model + oracle post-processing + the C geometry oracle.
One tile per model call, exactly like the reference (batch = 1), so it also serves as bench.py's cpu_baseline."""
import numpy as np
import torch

from . import geom as og
from . import postproc as pp

MARGIN_128, MARGIN_416 = 10, 20
STRIKE = 1


class OracleModel:
    """model(crop_bgr_uint8, conf=...) -> [obj] with obj.obb rows, Ultralytics-shaped (Detect_OBB.py:81-83, 228-231)."""

    def __init__(self, net, imgsz=416, precision="fp32", head_fn=None, iou=0.7, max_det=300, multich_fn=None):
        self.net, self.imgsz, self.precision, self.head_fn, self.iou, self.max_det = net, imgsz, precision, head_fn, iou, max_det
        self.multich_fn = multich_fn  # tests: the 4-channel input builder under test instead of the numpy restatement (isolates the forward)

    def predict_rows(self, crop, conf):
        if self.net.ch == 4 and crop.shape[2] == 3:  # run_inference_on_crop :76-77: net_input = build_multich(crop_bgr, channels)
            from . import dtedge
            crop = self.multich_fn(crop) if self.multich_fn is not None else dtedge.build_multich(crop, 4)
        lb, p = pp.letterbox(crop, self.imgsz)
        h, w = lb.shape[:2]
        head = self.head_fn(lb[None]) if self.head_fn is not None else self.net.forward_raw(lb[None], self.precision)
        pred = pp.decode(head, h, w, self.net.nc)
        det = pp.non_max_suppression(pred, conf, self.iou, self.max_det, self.net.nc)[0]
        if det.shape[0] == 0:
            return np.zeros((0, 8), np.float32), np.zeros(0, np.int64), np.zeros(0, np.float32)
        obb, corners = pp.construct_result(det, (h, w), crop.shape[:2])
        return corners.reshape(-1, 8).numpy(), obb[:, 6].numpy().astype(np.int64), obb[:, 5].numpy()


def detect_symbols(image, model, tile_size, overlap, conf=0.25, iou_threshold=0.4, apply_border=True):
    H, W = image.shape[:2]
    margin = MARGIN_128 if tile_size <= 128 else MARGIN_416
    out = []
    for (x, y, x2, y2) in og.tile_grid(H, W, tile_size, overlap):
        crop = np.ascontiguousarray(image[y:y2, x:x2])
        pts, cls, cf = model.predict_rows(crop, conf)
        crop_dets = []
        for k in range(len(cls)):
            p = [float(v) for v in pts[k]]
            g = [p[0] + x, p[1] + y, p[2] + x, p[3] + y, p[4] + x, p[5] + y, p[6] + x, p[7] + y]
            if apply_border and margin > 0 and not og.center_inside_safe_region(g, x, y, x2 - x, y2 - y, margin):
                continue
            ang = og.compute_angle_from_bbox(p) if int(cls[k]) == STRIKE else 0.0
            crop_dets.append(tuple(g) + (int(cls[k]), float(cf[k]), ang))
        out.extend(og.merge_detections(crop_dets, iou_threshold))
    return out


def process_image(image, models, tile_sizes=(128, 416), overlaps=(30, 100), conf=0.25, iou_threshold=0.4):
    dets_by_scale = {}
    for ts, ov, m in zip(tile_sizes, overlaps, models):
        dets_by_scale[ts] = detect_symbols(image, m, ts, ov, conf, iou_threshold)
    consensus = og.cross_scale_consensus_filter(dets_by_scale)
    return og.merge_detections(consensus, iou_threshold), dets_by_scale
