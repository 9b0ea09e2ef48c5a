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


class BoxBranchStep:
    """ONE assembled training step of the box branch of a head level (ultralytics Detect.cv2[i]: Conv 3x3 + SiLU, Conv 3x3 + SiLU, Conv2d 1x1
    -> 4 x 16 DFL logits per anchor; BatchNorm folded as in the OBBW blob) under the DFL term of the OBB loss -- what `loss.backward()` +
    `optimizer.step()` do to these three layers inside `model.train(...)` (Train_OBB.py:796-841; bf16 autocast over fp32 master weights):
        pack (device) -> conv -> SiLU -> conv -> SiLU -> conv -> DFL loss + gradient -> [wgrad, bias grad, dgrad, SiLU'] x 3 -> optimiser.
    Every arithmetic step is a kernel of libobbhip; the master weights live in two flat optimiser groups (weights with decay, biases
    without), the gradients are written straight into the groups' flat gradient buffers.  Activations and gradients between layers are bf16
    (one rounding per tensor), sums fp32."""

    def __init__(self, weights, biases, H, W, optimizer="SGD", lr=0.01, momentum=0.9, weight_decay=5e-4):
        dev = weights[0].device
        self.H, self.W = H, W
        self.wshapes = [tuple(w.shape) for w in weights]
        self.bshapes = [tuple(b.shape) for b in biases]
        self.opt_w = FlatOptimizer(sum(int(math.prod(s)) for s in self.wshapes), dev, optimizer, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.opt_b = FlatOptimizer(sum(int(math.prod(s)) for s in self.bshapes), dev, optimizer, lr=lr, momentum=momentum, weight_decay=0.0)
        self.w, self.dw = FlatOptimizer.views(self.opt_w.param, self.wshapes), FlatOptimizer.views(self.opt_w.grad, self.wshapes)
        self.b, self.db = FlatOptimizer.views(self.opt_b.param, self.bshapes), FlatOptimizer.views(self.opt_b.grad, self.bshapes)
        for dst, src in zip(self.w + self.b, list(weights) + list(biases)):
            dst.copy_(src)

    def forward_backward(self, x, target_ltrb, weight=None, target_scores_sum=1.0):
        """x bf16 [B,H,W,cin]; target_ltrb fp32 [B*H*W, 4] (bins), weight fp32 [B*H*W] or None -> (loss fp32[1], dx bf16 like x); the weight
        and bias gradients are left in the optimiser groups' gradient buffers (self.dw, self.db)."""
        H, W = self.H, self.W
        ks = [s[2] for s in self.wshapes]
        fw = [ops.conv_pack_bf16(w, H, W) for w in self.w]                       # this step's bf16 weights, forward order
        z1 = ops.conv_fwd_bf16(x, fw[0], self.b[0], self.wshapes[0][0], ks[0]); a1 = ops.silu_bf16(z1)
        z2 = ops.conv_fwd_bf16(a1, fw[1], self.b[1], self.wshapes[1][0], ks[1]); a2 = ops.silu_bf16(z2)
        out = ops.conv_fwd_bf16(a2, fw[2], self.b[2], self.wshapes[2][0], ks[2])
        loss, g = ops.dfl_loss(out.float().reshape(-1, self.wshapes[2][0]), target_ltrb, weight, target_scores_sum)
        d3 = g.reshape(out.shape).to(torch.bfloat16)
        bw = [ops.conv_pack_bf16(w, H, W, dgrad_form=True) for w in self.w]     # flipped / transposed: the input-gradient convolutions
        self.dw[2].copy_(ops.conv_wgrad_bf16(a2, d3, ks[2])); ops.bias_grad_bf16(d3, self.db[2])
        d2 = ops.silu_bwd_bf16(z2, ops.conv_fwd_bf16(d3, bw[2], None, self.wshapes[2][1], ks[2]))
        self.dw[1].copy_(ops.conv_wgrad_bf16(a1, d2, ks[1])); ops.bias_grad_bf16(d2, self.db[1])
        d1 = ops.silu_bwd_bf16(z1, ops.conv_fwd_bf16(d2, bw[1], None, self.wshapes[1][1], ks[1]))
        self.dw[0].copy_(ops.conv_wgrad_bf16(x, d1, ks[0])); ops.bias_grad_bf16(d1, self.db[0])
        dx = ops.conv_fwd_bf16(d1, bw[0], None, self.wshapes[0][1], ks[0])
        return loss, dx

    def step(self, x, target_ltrb, weight=None, target_scores_sum=1.0, group=None):
        loss, dx = self.forward_backward(x, target_ltrb, weight, target_scores_sum)
        allreduce_gradients([self.opt_w.grad, self.opt_b.grad], group)           # DDP: the gradient average (no-op on one rank)
        self.opt_w.step(); self.opt_b.step()
        return loss, dx
