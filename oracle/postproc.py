"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU fp32 restatement of what the Ultralytics OBB predictor does after the
network forward (call site Detect_OBB.py:81-83, consumed at :228-231), ultralytics==8.3.196:

  decode      nn/modules/head.py  OBB.forward/_inference, DFL, utils/tal.py dist2rbox, make_anchors
  nms         utils/ops.py        non_max_suppression(rotated=True) + nms_rotated (Fast-NMS), utils/metrics.py batch_probiou
  results     models/yolo/obb/predict.py construct_result, utils/ops.py regularize_rboxes / scale_boxes / xywhr2xyxyxyxy
  letterbox   data/augment.py     LetterBox(auto=True, stride=32) as the predictor's pre_transform for one image

ultralytics is not vendored under /root/reference and not installable offline: these follow the published
algorithms.  **Parity unpinned** except for the geometry conventions, which hold on all
44 rows of the reference's own goldens (tests/test_oracle_postproc.py).
"""
import math

import numpy as np
import torch

from .yolo11_obb import REG_MAX, make_anchors


@torch.no_grad()
def decode(head, h, w, nc):
    """head [B,A,64+nc+1] raw logits -> prediction [B, 4+nc+1, A] (xywh in letterboxed-input pixels, class scores, theta)"""
    B, A, _ = head.shape
    anchors, strides = make_anchors(h, w)
    box = head[..., : 4 * REG_MAX].view(B, A, 4, REG_MAX)
    dist = (box.softmax(-1) * torch.arange(REG_MAX, dtype=torch.float32)).sum(-1)  # DFL expectation [B,A,4]
    angle = (head[..., 4 * REG_MAX + nc].sigmoid() - 0.25) * math.pi  # [B,A]
    lt, rb = dist[..., :2], dist[..., 2:]
    cos, sin = torch.cos(angle), torch.sin(angle)
    xf, yf = ((rb - lt) / 2).unbind(-1)
    x, y = xf * cos - yf * sin, xf * sin + yf * cos
    xy = torch.stack([x, y], -1) + anchors
    dbox = torch.cat([xy, lt + rb], -1) * strides[:, None]
    cls = head[..., 4 * REG_MAX: 4 * REG_MAX + nc].sigmoid()
    return torch.cat([dbox, cls, angle[..., None]], -1).transpose(1, 2).contiguous()


def _cov(boxes):
    a = boxes[:, 2:3].pow(2) / 12
    b = boxes[:, 3:4].pow(2) / 12
    c = boxes[:, 4:5]
    cos, sin = c.cos(), c.sin()
    cos2, sin2 = cos.pow(2), sin.pow(2)
    return a * cos2 + b * sin2, a * sin2 + b * cos2, (a - b) * cos * sin


@torch.no_grad()
def batch_probiou(obb1, obb2, eps=1e-7):
    x1, y1 = obb1[..., :2].split(1, dim=-1)
    x2, y2 = (x.squeeze(-1)[None] for x in obb2[..., :2].split(1, dim=-1))
    a1, b1, c1 = _cov(obb1)
    a2, b2, c2 = (x.squeeze(-1)[None] for x in _cov(obb2))
    t1 = (((a1 + a2) * (y1 - y2).pow(2) + (b1 + b2) * (x1 - x2).pow(2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.25
    t2 = (((c1 + c2) * (x2 - x1) * (y1 - y2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.5
    t3 = (((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2))
          / (4 * ((a1 * b1 - c1.pow(2)).clamp_(0) * (a2 * b2 - c2.pow(2)).clamp_(0)).sqrt() + eps) + eps).log() * 0.5
    bd = (t1 + t2 + t3).clamp(eps, 100.0)
    hd = (1.0 - (-bd).exp() + eps).sqrt()
    return 1 - hd


@torch.no_grad()
def nms_rotated(boxes, scores, threshold):
    """Fast-NMS: keep j iff no higher-scored i has probiou >= threshold (suppressed boxes still suppress)."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    ious = batch_probiou(b, b).triu_(diagonal=1)
    pick = torch.nonzero((ious >= threshold).sum(0) <= 0).squeeze_(-1)
    return order[pick]


@torch.no_grad()
def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.7, max_det=300, nc=12, max_wh=7680, max_nms=30000):
    """pred [B, 4+nc+1, A] -> list of [n,7] (x, y, w, h, conf, cls, theta)"""
    out = []
    mi = 4 + nc
    xc = pred[:, 4:mi].amax(1) > conf_thres
    pred = pred.transpose(-1, -2)
    for xi, x in enumerate(pred):
        x = x[xc[xi]]
        if not x.shape[0]:
            out.append(torch.zeros((0, 7)))
            continue
        box, cls, mask = x.split((4, nc, 1), 1)
        conf, j = cls.max(1, keepdim=True)
        x = torch.cat((box, conf, j.float(), mask), 1)[conf.view(-1) > conf_thres]
        n = x.shape[0]
        if not n:
            out.append(torch.zeros((0, 7)))
            continue
        if n > max_nms:
            x = x[x[:, 4].argsort(descending=True, stable=True)[:max_nms]]
        c = x[:, 5:6] * max_wh
        boxes = torch.cat((x[:, :2] + c, x[:, 2:4], x[:, -1:]), dim=-1)
        i = nms_rotated(boxes, x[:, 4], iou_thres)[:max_det]
        out.append(x[i])
    return out


@torch.no_grad()
def regularize_rboxes(rboxes):
    x, y, w, h, t = rboxes.unbind(dim=-1)
    swap = t % math.pi >= math.pi / 2
    w_ = torch.where(swap, h, w)
    h_ = torch.where(swap, w, h)
    t = t % (math.pi / 2)
    return torch.stack([x, y, w_, h_, t], dim=-1)


def scale_boxes_xywh(img1_shape, boxes, img0_shape):
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    boxes = boxes.clone()
    boxes[..., 0] -= pad[0]
    boxes[..., 1] -= pad[1]
    boxes[..., :4] /= gain
    return boxes, gain, pad


@torch.no_grad()
def xywhr2xyxyxyxy(x):
    ctr = x[..., :2]
    w, h, angle = (x[..., i: i + 1] for i in range(2, 5))
    cos_value, sin_value = torch.cos(angle), torch.sin(angle)
    vec1 = torch.cat([w / 2 * cos_value, w / 2 * sin_value], -1)
    vec2 = torch.cat([-h / 2 * sin_value, h / 2 * cos_value], -1)
    pt1 = ctr + vec1 + vec2
    pt2 = ctr + vec1 - vec2
    pt3 = ctr - vec1 - vec2
    pt4 = ctr - vec1 + vec2
    return torch.stack([pt1, pt2, pt3, pt4], -2)


@torch.no_grad()
def construct_result(pred, img_shape, orig_shape):
    """pred [n,7] from NMS -> obb.data [n,7] = (x,y,w,h,theta,conf,cls) in crop pixels, and corners [n,4,2]"""
    rb = regularize_rboxes(torch.cat([pred[:, :4], pred[:, -1:]], dim=-1))
    rb[:, :4], gain, pad = scale_boxes_xywh(img_shape, rb[:, :4], orig_shape)
    obb = torch.cat([rb, pred[:, 4:6]], dim=-1)
    return obb, xywhr2xyxyxyxy(obb[:, :5])


# ---------------------------------------------------------------- LetterBox(auto=True, stride=32, center=True, scaleup=True)

def letterbox_params(h, w, imgsz, stride=32):
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (imgsz - new_w) % stride, (imgsz - new_h) % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return {"r": r, "new_w": new_w, "new_h": new_h, "top": top, "bottom": bottom, "left": left, "right": right,
            "out_h": new_h + top + bottom, "out_w": new_w + left + right, "resize": (w, h) != (new_w, new_h)}


def resize_bilinear_u8(img, new_w, new_h):
    """cv2.resize(..., INTER_LINEAR) for uint8, restated: half-pixel centres, 11-bit fixed-point coefficients
    (cv2 is absent offline -> parity unpinned)."""
    h, w, c = img.shape
    sx, sy = w / new_w, h / new_h

    def coeffs(n_dst, n_src, scale):
        idx = np.zeros(n_dst, np.int64)
        a = np.zeros((n_dst, 2), np.int64)
        for d in range(n_dst):
            f = np.float32((d + 0.5) * scale - 0.5)  # OpenCV: double expression, stored to float
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            if s < 0:
                s, f = 0, 0.0
            if s >= n_src - 1:
                s, f = n_src - 1, 0.0
            idx[d] = s
            a1 = int(np.rint(np.float32(f) * np.float32(2048)))  # cvRound: ties-to-even
            a[d] = (2048 - a1, a1)
        return idx, a

    xi, xa = coeffs(new_w, w, sx)
    yi, ya = coeffs(new_h, h, sy)
    src = img.astype(np.int64)
    x1 = np.minimum(xi + 1, w - 1)
    rows = src[:, xi, :] * xa[:, 0][None, :, None] + src[:, x1, :] * xa[:, 1][None, :, None]  # [h,new_w,c]
    y1 = np.minimum(yi + 1, h - 1)
    out = rows[yi] * ya[:, 0][:, None, None] + rows[y1] * ya[:, 1][:, None, None]
    return ((out + (1 << 21)) >> 22).clip(0, 255).astype(np.uint8)


def letterbox(img, imgsz, stride=32):
    h, w = img.shape[:2]
    p = letterbox_params(h, w, imgsz, stride)
    if p["resize"]:
        img = resize_bilinear_u8(img, p["new_w"], p["new_h"])
    out = np.full((p["out_h"], p["out_w"], img.shape[2]), 114, np.uint8)
    out[p["top"]: p["top"] + p["new_h"], p["left"]: p["left"] + p["new_w"]] = img
    return out, p
