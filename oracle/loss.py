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


# ---------------------------------------------------------------------------------------------- RotatedTaskAlignedAssigner
def xywhr2xyxyxyxy(x):
    """ultralytics.utils.ops.xywhr2xyxyxyxy (torch branch): [..., 5] -> [..., 4, 2]"""
    ctr = x[..., :2]
    w, h, angle = (x[..., i:i + 1] for i in range(2, 5))
    cos_value, sin_value = torch.cos(angle), torch.sin(angle)
    vec1 = torch.cat([w / 2 * cos_value, w / 2 * sin_value], -1)
    vec2 = torch.cat([-h / 2 * sin_value, h / 2 * cos_value], -1)
    return torch.stack([ctr + vec1 + vec2, ctr + vec1 - vec2, ctr - vec1 - vec2, ctr - vec1 + vec2], -2)


def rotated_tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=10, alpha=0.5, beta=6.0, eps=1e-9):
    """ultralytics==8.3.196 utils/tal.py TaskAlignedAssigner.forward with the RotatedTaskAlignedAssigner overrides (iou_calculation =
    probiou(...).clamp_(0), select_candidates_in_gts = point-in-rotated-box by projections), restated from the published source
    (**parity unpinned**: ultralytics is not installable offline).  Shapes as in ultralytics: gt_labels [bs, n_max, 1], mask_gt [bs, n_max, 1].
    -> (target_labels, target_bboxes, target_scores, fg_mask, target_gt_idx) + the intermediates (align_metric, overlaps) for the tests."""
    bs, na, nc = pd_scores.shape
    n_max = gt_bboxes.shape[1]
    # select_candidates_in_gts
    corners = xywhr2xyxyxyxy(gt_bboxes)
    a, b, _, d = corners.split(1, dim=-2)
    ab, ad = b - a, d - a
    ap = anc_points - a
    norm_ab, norm_ad = (ab * ab).sum(-1), (ad * ad).sum(-1)
    ap_dot_ab, ap_dot_ad = (ap * ab).sum(-1), (ap * ad).sum(-1)
    mask_in_gts = ((ap_dot_ab >= 0) & (ap_dot_ab <= norm_ab) & (ap_dot_ad >= 0) & (ap_dot_ad <= norm_ad)).to(pd_scores.dtype)
    # get_box_metrics
    mgt = (mask_in_gts * mask_gt).bool()
    overlaps = torch.zeros([bs, n_max, na], dtype=pd_bboxes.dtype)
    bbox_scores = torch.zeros([bs, n_max, na], dtype=pd_scores.dtype)
    ind = torch.zeros([2, bs, n_max], dtype=torch.long)
    ind[0] = torch.arange(end=bs).view(-1, 1).expand(-1, n_max)
    ind[1] = gt_labels.squeeze(-1)
    bbox_scores[mgt] = pd_scores[ind[0], :, ind[1]][mgt]
    pd_boxes = pd_bboxes.unsqueeze(1).expand(-1, n_max, -1, -1)[mgt]
    gt_boxes = gt_bboxes.unsqueeze(2).expand(-1, -1, na, -1)[mgt]
    overlaps[mgt] = probiou(gt_boxes, pd_boxes).squeeze(-1).clamp_(0)
    align_metric = bbox_scores.pow(alpha) * overlaps.pow(beta)
    # select_topk_candidates
    topk_metrics, topk_idxs = torch.topk(align_metric, topk, dim=-1, largest=True)
    topk_mask = mask_gt.expand(-1, -1, topk).bool()
    topk_idxs.masked_fill_(~topk_mask, 0)
    count_tensor = torch.zeros(align_metric.shape, dtype=torch.int8)
    ones = torch.ones_like(topk_idxs[:, :, :1], dtype=torch.int8)
    for k in range(topk):
        count_tensor.scatter_add_(-1, topk_idxs[:, :, k:k + 1], ones)
    count_tensor.masked_fill_(count_tensor > 1, 0)
    mask_pos = count_tensor.to(align_metric.dtype) * mask_in_gts * mask_gt
    # select_highest_overlaps
    fg_mask = mask_pos.sum(-2)
    if fg_mask.max() > 1:
        mask_multi_gts = (fg_mask.unsqueeze(1) > 1).expand(-1, n_max, -1)
        max_overlaps_idx = overlaps.argmax(1)
        is_max_overlaps = torch.zeros(mask_pos.shape, dtype=mask_pos.dtype)
        is_max_overlaps.scatter_(1, max_overlaps_idx.unsqueeze(1), 1)
        mask_pos = torch.where(mask_multi_gts, is_max_overlaps, mask_pos).float()
        fg_mask = mask_pos.sum(-2)
    target_gt_idx = mask_pos.argmax(-2)
    # get_targets
    batch_ind = torch.arange(end=bs, dtype=torch.int64)[..., None]
    tgi = target_gt_idx + batch_ind * n_max
    target_labels = gt_labels.long().flatten()[tgi]
    target_bboxes = gt_bboxes.view(-1, gt_bboxes.shape[-1])[tgi]
    target_labels.clamp_(0)
    target_scores = torch.zeros((bs, na, nc), dtype=torch.int64)
    target_scores.scatter_(2, target_labels.unsqueeze(-1), 1)
    fg_scores_mask = fg_mask[:, :, None].repeat(1, 1, nc)
    target_scores = torch.where(fg_scores_mask > 0, target_scores, 0)
    # normalise
    align_metric = align_metric * mask_pos
    pos_align_metrics = align_metric.amax(dim=-1, keepdim=True)
    pos_overlaps = (overlaps * mask_pos).amax(dim=-1, keepdim=True)
    norm_align_metric = (align_metric * pos_overlaps / (pos_align_metrics + eps)).amax(-2).unsqueeze(-1)
    target_scores = target_scores * norm_align_metric
    return target_labels, target_bboxes, target_scores, fg_mask.bool(), target_gt_idx, align_metric, overlaps
