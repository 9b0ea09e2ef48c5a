"""Scratch check: closing 1x1 fused behind the Bottleneck vs separate launches, per block; dumps tensors for offline analysis."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oracle.yolo11_obb import Yolo11OBB
net = Yolo11OBB("n", nc=12, ch=3, seed=0)
B, h, w = 1, 416, 416
x = torch.as_tensor(np.random.default_rng(1).integers(0, 256, (B, h, w, 3), dtype=np.uint8)).cuda()
names = ("model.2.cv2", "model.4.cv2", "model.16.cv2")
os.environ["OBB_BNECK_CV2"] = "0"
ops.model_load(net.to_blob(), precision="f16")
ops.forward(x)
dump = {}
for n in ("model.2.cv1", "model.2.m.0.cv2", "model.2.cv2", "model.4.cv1", "model.4.m.0.cv2", "model.4.cv2"):
    dump["ref_" + n] = ops.debug_activation(n, B, h, w).cpu().numpy().astype(np.float16)
ref = {n: ops.debug_activation(n, B, h, w).clone() for n in names}
del os.environ["OBB_BNECK_CV2"]
ops.model_load(net.to_blob(), precision="f16")
ops.forward(x)
for n in names:
    got = ops.debug_activation(n, B, h, w)
    d = (got - ref[n]).abs()
    print(n, tuple(got.shape), "max", float(d.max()), "frac>0", float((d > 0).float().mean()), "mean", float(d.mean()), "ref mean", float(ref[n].abs().mean()))
for n in ("model.2.cv2", "model.4.cv2"):
    dump["got_" + n] = ops.debug_activation(n, B, h, w).cpu().numpy().astype(np.float16)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bn_dump.npz"), **dump)
