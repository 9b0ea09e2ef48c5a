"""Training-step slices behind `model.train(...)` (Train_OBB.py:796-841) that sit BETWEEN the loss / backward
kernels (loss.py, ops.conv_dgrad_bf16 / conv_wgrad_bf16, ops.rotated_tal_assign) and the next forward:

  * `optimizer_config` / `param_group_of`: Ultralytics' `BaseTrainer.build_optimizer` rules (ultralytics==8.3.x, recalled -- the package
    is not installed here, so this restatement is UNPINNED; the numerics below are pinned against torch.optim itself): optimizer "auto"
    = SGD(lr 0.01, momentum 0.9, nesterov) above 10 000 iterations, else AdamW(lr = round(0.002 * 5 / (4 + nc), 6), betas (0.9, 0.999));
    three groups: biases (no decay), norm weights (no decay), other weights (decay = weight_decay * batch * accumulate / nbs);
  * `FlatOptimizer`: a parameter group as ONE flat fp32 device buffer (parameters, gradients, state) updated by one HIP launch
    (csrc/optim.hip) -- the arithmetic of torch.optim.SGD / AdamW, checked against them step by step;
  * `allreduce_gradients`: the DDP gradient average of `device="0,1"` (Train_OBB.py: DEVICE) as bucketed all-reduces over
    torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; gloo in the CPU test): the flat gradient buffers are already
    contiguous, so a bucket is a view, not a copy; buckets of `bucket_mb` keep a ring all-reduce per-link bandwidth-bound rather than
    latency-bound (7 xGMI links x ~153 GB/s per GPU: a 25 MB bucket is ~0.1 ms of wire time per hop)."""
import math

import torch

from . import ops


def optimizer_config(nc, iterations, name="auto", lr0=0.003, momentum=0.937, weight_decay=0.001, batch=16, nbs=64):
    """-> dict(name, lr, momentum, weight_decay, accumulate).  `name="auto"` ignores lr0 / momentum like the trainer does."""
    accumulate = max(round(nbs / batch), 1)
    wd = weight_decay * batch * accumulate / nbs
    if name == "auto":
        lr_fit = round(0.002 * 5 / (4 + nc), 6)
        name, lr0, momentum = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", lr_fit, 0.9)
    if name not in ("SGD", "AdamW"):
        raise NotImplementedError(f"optimizer {name}: only the two that `auto` selects are built")
    return {"name": name, "lr": lr0, "momentum": momentum, "weight_decay": wd, "accumulate": accumulate}


def iterations_of(n_train_tiles, epochs, batch=16, nbs=64):
    """The trainer's iteration estimate that feeds `auto`: ceil(len(dataset) / max(batch, nbs)) * epochs."""
    return math.ceil(n_train_tiles / max(batch, nbs)) * epochs


def param_group_of(param_name, module_is_norm=False):
    """0 = weights with decay, 1 = norm weights (no decay), 2 = biases (no decay) -- the trainer's g[0] / g[1] / g[2]."""
    if "bias" in param_name:
        return 2
    if module_is_norm or "logit_scale" in param_name:
        return 1
    return 0


class FlatOptimizer:
    """One parameter group: `param` (flat fp32 device tensor, updated in place) with its optimiser state.  `views(shapes)` hands out the
    per-layer tensors as views of the flat buffer, so the layers' kernels write gradients straight into `grad`."""

    def __init__(self, n, device, name="SGD", lr=0.01, momentum=0.9, weight_decay=0.0, nesterov=True, betas=None, eps=1e-8):
        if name not in ("SGD", "AdamW"):
            raise NotImplementedError(name)
        self.name, self.lr, self.momentum, self.weight_decay, self.nesterov, self.eps = name, lr, momentum, weight_decay, nesterov, eps
        self.betas = betas or (momentum, 0.999)
        n4 = (n + 3) // 4 * 4
        self.n = n
        self.param = torch.zeros(n4, dtype=torch.float32, device=device)[:n]
        self.grad = torch.zeros(n4, dtype=torch.float32, device=device)[:n]
        self.state = [torch.zeros(n4, dtype=torch.float32, device=device)[:n] for _ in range(1 if name == "SGD" else 2)]
        self.steps = 0

    @staticmethod
    def views(flat, shapes):
        out, o = [], 0
        for s in shapes:
            k = int(math.prod(s))
            out.append(flat[o:o + k].view(*s))
            o += k
        if o != flat.numel():
            raise ValueError("views: the shapes do not cover the buffer")
        return out

    def step(self, lr=None):
        lr = self.lr if lr is None else lr
        self.steps += 1
        if self.name == "SGD":
            ops.sgd_step(self.param, self.grad, self.state[0], lr, self.momentum, self.weight_decay, self.nesterov, first_step=self.steps == 1)
        else:
            ops.adamw_step(self.param, self.grad, self.state[0], self.state[1], self.steps, lr, self.betas, self.eps, self.weight_decay)

    def zero_grad(self):
        self.grad.zero_()


def allreduce_gradients(flat_grads, group=None, bucket_mb=25.0, average=True):
    """Average the flat gradient buffers over the ranks of `group` in place: bucketed `all_reduce`s (views of the flat buffers, launched
    asynchronously, waited for at the end).  Returns the number of collectives issued.  World size 1 / no process group: nothing to do."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    per = max(1, int(bucket_mb * (1 << 20)) // 4)
    works = []
    for g in flat_grads:
        for o in range(0, g.numel(), per):
            works.append(dist.all_reduce(g[o:o + per], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if average:
        for g in flat_grads:
            g.mul_(1.0 / world)
    return len(works)
