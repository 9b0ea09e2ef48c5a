import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
blob = open(make_weights.ensure("n", 12, 3, 0), "rb").read()
x = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (2, 416, 416, 3), dtype=np.uint8)).cuda()
res = {}
names = ("model.7", "model.8.m.0.cv3", "model.8.cv2", "model.22.m.0.cv3", "model.22.cv2")
for tail in (False, True):
    ops.model_load(blob, tail=tail)
    h = ops.forward(x).clone()
    res[(tail, "head")] = h
    for name in names:
        res[(tail, name)] = ops.debug_activation(name, 2, 416, 416).clone()
torch.cuda.synchronize()
for name in names + ("head",):
    a, b = res[(False, name)], res[(True, name)]
    d = (a - b).abs()
    print(name, "max", float(d.max()), "mean", float(d.mean()), "frac nonzero", float((d > 0).float().mean()), "ref absmean", float(a.abs().mean()))
