"""TEST INFRASTRUCTURE ONLY -- torch restatement of the ProbIoU rotated-box loss of ultralytics==8.3.196 (utils/loss.py RotatedBboxLoss
+ utils/metrics.py probiou / _get_covariance_matrix), reached from `model.train(...)` at Train_OBB.py:796-841.  ultralytics is not
vendored and not installable offline: restated from the published formula (arXiv:2106.06072) -- **parity
unpinned**.  Gradients come from torch.autograd, which is what the HIP kernel's closed-form backward is checked against."""
import torch


def _cov(boxes):
    gbbs = torch.cat((boxes[:, 2:4].pow(2) / 12, boxes[:, 4:]), dim=-1)
    a, b, c = gbbs.split(1, dim=-1)
    cos, sin = c.cos(), c.sin()
    cos2, sin2 = cos.pow(2), sin.pow(2)
    return a * cos2 + b * sin2, a * sin2 + b * cos2, (a - b) * cos * sin


def probiou(obb1, obb2, eps=1e-7):
    x1, y1 = obb1[..., :2].split(1, dim=-1)
    x2, y2 = obb2[..., :2].split(1, dim=-1)
    a1, b1, c1 = _cov(obb1)
    a2, b2, c2 = _cov(obb2)
    t1 = (((a1 + a2) * (y1 - y2).pow(2) + (b1 + b2) * (x1 - x2).pow(2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.25
    t2 = (((c1 + c2) * (x2 - x1) * (y1 - y2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.5
    t3 = (((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2)) / (4 * ((a1 * b1 - c1.pow(2)).clamp(0) * (a2 * b2 - c2.pow(2)).clamp(0)).sqrt() + eps) + eps).log() * 0.5
    bd = (t1 + t2 + t3).clamp(eps, 100.0)
    hd = (1.0 - (-bd).exp() + eps).sqrt()
    return 1 - hd


def probiou_loss(pred, target, weight, target_scores_sum):
    iou = probiou(pred, target)
    w = weight if weight is not None else torch.ones(pred.shape[0], dtype=pred.dtype)
    return ((1.0 - iou) * w[:, None]).sum() / target_scores_sum


def dfl_loss(pred_dist, target_ltrb, weight, target_scores_sum, reg_max=16):
    """ultralytics DFLoss as used by RotatedBboxLoss: target clamped to [0, reg_max - 1 - 0.01], cross entropy against the two neighbouring
    bins weighted by the distances to them, mean over the 4 sides, times weight, over target_scores_sum.  pred_dist [n, 4*reg_max]."""
    import torch.nn.functional as F
    t = target_ltrb.clamp(0, reg_max - 1 - 0.01)
    tl = t.long()
    tr = tl + 1
    wl = tr - t
    wr = 1 - wl
    pd = pred_dist.reshape(-1, reg_max)
    l = (F.cross_entropy(pd, tl.reshape(-1), reduction="none").reshape(tl.shape) * wl + F.cross_entropy(pd, tr.reshape(-1), reduction="none").reshape(tl.shape) * wr).mean(-1, keepdim=True)
    w = weight if weight is not None else torch.ones(t.shape[0], dtype=pred_dist.dtype)
    return (l * w[:, None]).sum() / target_scores_sum


def bce_loss(logits, targets, target_scores_sum):
    import torch.nn.functional as F
    return F.binary_cross_entropy_with_logits(logits, targets, reduction="none").sum() / target_scores_sum
