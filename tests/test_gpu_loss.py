"""f1 (first slice): the ProbIoU rotated-box loss, forward + backward in one HIP kernel, against torch.autograd on the restated
Ultralytics formula (oracle/loss.py) -- fp32 on the device vs an fp64 autograd evaluation of the same inputs."""
import numpy as np
import pytest
import torch

from oracle import loss as ol

pytestmark = pytest.mark.gpu


def _pairs(seed, n):
    rng = np.random.default_rng(seed)
    t = np.stack([rng.uniform(0, 416, n), rng.uniform(0, 416, n), rng.uniform(8, 120, n), rng.uniform(8, 120, n), rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)], 1)
    p = t.copy()
    far = rng.uniform(size=n) < 0.3  # a third of the predictions far from their target, the rest near it (the matched regime of training)
    p[:, :2] += np.where(far[:, None], rng.normal(0, 60, (n, 2)), rng.normal(0, 4, (n, 2)))
    p[:, 2:4] *= rng.uniform(0.6, 1.6, (n, 2))
    p[:, 4] += rng.normal(0, 0.3, n)
    w = rng.uniform(0.05, 1.0, n)
    return p.astype(np.float32), t.astype(np.float32), w.astype(np.float32)


@pytest.mark.parametrize("n", [1, 257, 20000, 300000])
def test_probiou_loss_forward_and_backward(n):
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import loss as L
    p, t, w = _pairs(n, n)
    tss = float(w.sum())
    pd = torch.tensor(p, device="cuda", requires_grad=True)
    out = L.probiou_loss(pd, torch.tensor(t).cuda(), torch.tensor(w).cuda(), tss)
    out.backward()
    p64 = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    ref = ol.probiou_loss(p64, torch.tensor(t, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), tss)
    ref.backward()
    g, gr = pd.grad.cpu().double(), p64.grad
    rel = float((g - gr).abs().max() / gr.abs().max())
    print(n, "loss", float(out), float(ref), "max |d grad| / max |grad|", rel)
    assert abs(float(out) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    assert rel < 2e-4
    # elementwise: every gradient within 2 % of its own row's scale (fp32 evaluation of a formula with exp / log / sqrt chains and
    # cancelling terms; measured worst case over 3e5 pairs: 5.3e-3)
    scale = gr.abs().amax(1, keepdim=True).clamp_min(1e-6 * float(gr.abs().max()))
    assert float(((g - gr).abs() / scale).max()) < 2e-2
    # upstream gradient is honoured and the weightless form works
    pd2 = torch.tensor(p, device="cuda", requires_grad=True)
    (3.0 * L.probiou_loss(pd2, torch.tensor(t).cuda(), None, float(n))).backward()
    p64b = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    (3.0 * ol.probiou_loss(p64b, torch.tensor(t, dtype=torch.float64), None, float(n))).backward()
    assert float((pd2.grad.cpu().double() - p64b.grad).abs().max() / p64b.grad.abs().max()) < 2e-4


@pytest.mark.parametrize("n", [1, 300, 70000])
def test_dfl_loss_forward_and_backward(n):
    """f1, second slice: the DFL term (box-side distributions, 16 bins) against torch's cross_entropy in fp64 (oracle/loss.py)."""
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import loss as L
    rng = np.random.default_rng(n)
    logits = rng.normal(0, 2.5, (n, 64)).astype(np.float32)
    t = rng.uniform(-0.5, 16.5, (n, 4)).astype(np.float32)   # beyond both ends: the clamp of the reference
    t[rng.uniform(size=(n, 4)) < 0.05] = 7.0                  # exact bin centres
    w = rng.uniform(0.05, 1.0, n).astype(np.float32)
    tss = float(w.sum())
    x = torch.tensor(logits, device="cuda", requires_grad=True)
    out = L.dfl_loss(x, torch.tensor(t).cuda(), torch.tensor(w).cuda(), tss)
    (2.0 * out).backward()
    x64 = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    ref = ol.dfl_loss(x64, torch.tensor(t, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), tss)
    (2.0 * ref).backward()
    assert abs(float(out) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(out), float(ref))
    g, gr = x.grad.cpu().double(), x64.grad
    assert float((g - gr).abs().max()) <= 2e-5 * float(gr.abs().max()) + 1e-9
    # weightless form
    x2 = torch.tensor(logits, device="cuda", requires_grad=True)
    L.dfl_loss(x2, torch.tensor(t).cuda(), None, float(n)).backward()
    x64b = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    ol.dfl_loss(x64b, torch.tensor(t, dtype=torch.float64), None, float(n)).backward()
    assert float((x2.grad.cpu().double() - x64b.grad).abs().max()) <= 2e-5 * float(x64b.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("shape", [(1, 1), (3, 3549, 12), (64, 3549, 12)])
def test_bce_loss_forward_and_backward(shape):
    """f1, second slice: the classification term (BCE with logits, summed, over target_scores_sum) against torch in fp64."""
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import loss as L
    rng = np.random.default_rng(sum(shape))
    logits = rng.normal(-4, 4, shape).astype(np.float32)
    logits.flat[:: max(1, logits.size // 50)] = rng.choice([-90.0, 90.0, 0.0], size=len(logits.flat[:: max(1, logits.size // 50)]))  # saturated logits
    tgt = np.where(rng.uniform(size=shape) < 0.02, rng.uniform(0.1, 1.0, shape), 0.0).astype(np.float32)
    tss = max(float(tgt.sum()), 1.0)
    x = torch.tensor(logits, device="cuda", requires_grad=True)
    out = L.bce_loss(x, torch.tensor(tgt).cuda(), tss)
    out.backward()
    x64 = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    ref = ol.bce_loss(x64, torch.tensor(tgt, dtype=torch.float64), tss)
    ref.backward()
    assert abs(float(out) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(out), float(ref))
    assert float((x.grad.cpu().double() - x64.grad).abs().max()) <= 2e-6 * float(x64.grad.abs().max()) + 1e-12
