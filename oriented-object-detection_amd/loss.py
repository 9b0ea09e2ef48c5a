"""First slice of the training step: the ProbIoU rotated-box loss that `model.train(...)`
(Train_OBB.py:796-841) reaches through Ultralytics' v8OBBLoss / RotatedBboxLoss, as an autograd function whose forward AND backward
are one HIP kernel (csrc/loss.hip).  Nothing else of training exists yet (assigner, DFL / BCE terms, conv backward, DDP)."""
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
