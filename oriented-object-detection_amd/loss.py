"""First slice of the training step: the ProbIoU rotated-box loss that `model.train(...)`
(Train_OBB.py:796-841) reaches through Ultralytics' v8OBBLoss / RotatedBboxLoss, as an autograd function whose forward AND backward
are one HIP kernel (csrc/loss.hip), plus the two other terms of that loss: DFL (box-side distributions) and BCE-with-logits (class
scores).  The other slices of the step: ops.rotated_tal_assign (assigner), ops.conv_dgrad_bf16 / conv_wgrad_bf16 (conv backward), train.py
(optimiser step, DDP gradient all-reduce)."""
import torch

from . import ops


class _ProbIoULoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, target_scores_sum):
        loss, grad = ops.probiou_loss(pred.contiguous(), target.contiguous(), None if weight is None else weight.contiguous(), target_scores_sum)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def probiou_loss(pred_bboxes, target_bboxes, weight=None, target_scores_sum=1.0):
    """sum((1 - probiou(pred, target)) * weight) / target_scores_sum over matched pairs [n,5] (x, y, w, h, theta); differentiable in pred."""
    return _ProbIoULoss.apply(pred_bboxes, target_bboxes, weight, float(target_scores_sum))


class _DFLLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_dist, target_ltrb, weight, target_scores_sum, reg_max):
        loss, grad = ops.dfl_loss(pred_dist.contiguous(), target_ltrb.contiguous(), None if weight is None else weight.contiguous(), target_scores_sum, reg_max)
        ctx.save_for_backward(grad)
        ctx.shape = pred_dist.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).reshape(ctx.shape), None, None, None, None


def dfl_loss(pred_dist, target_ltrb, weight=None, target_scores_sum=1.0, reg_max=16):
    """sum_i mean_side(CE(pred_i, floor t) (floor t + 1 - t) + CE(pred_i, floor t + 1) (t - floor t)) weight_i / target_scores_sum;
    pred_dist [n, 4*reg_max] logits, target_ltrb [n, 4] in bins; differentiable in pred_dist."""
    return _DFLLoss.apply(pred_dist, target_ltrb, weight, float(target_scores_sum), int(reg_max))


class _BCELoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, target_scores_sum):
        loss, grad = ops.bce_loss(logits.contiguous(), targets.contiguous(), target_scores_sum)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def bce_loss(logits, targets, target_scores_sum=1.0):
    """BCEWithLogitsLoss(reduction="none")(logits, targets).sum() / target_scores_sum; differentiable in logits."""
    return _BCELoss.apply(logits, targets, float(target_scores_sum))
