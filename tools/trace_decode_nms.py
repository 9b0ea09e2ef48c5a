"""candidate histogram of the bench heads + 12 calls of obb_decode_nms_gate (run under rocprofv3 --kernel-trace --stats for the per-kernel times)"""
import sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd  # noqa
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416, precision=prec) if prec != "f16" else YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
B, A = 1024, 3549
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
cmax = torch.empty((B, A), dtype=torch.float32, device="cuda")
head = ops.forward(tiles, cmax=cmax)
torch.cuda.synchronize()
cnts = []
for i in range(0, B, 64):
    pred = ops.decode(head[i:i + 64].contiguous(), 416, 416)
    cnts.append((pred[..., 4:16].amax(-1) > 0.25).sum(1).cpu())
c = torch.cat(cnts).numpy()
print("tiles", len(c), "mean", c.mean(), "max", c.max(), "n>256", int((c > 256).sum()), "n>512", int((c > 512).sum()), "n>1024", int((c > 1024).sum()))
print("quantiles 50/90/99:", np.percentile(c, [50, 90, 99]))
print("top 20:", np.sort(c)[-20:])
for _ in range(12):
    ops.decode_nms(head, 416, 416, 0.25, 0.7, 300, cmax=cmax)
torch.cuda.synchronize()
