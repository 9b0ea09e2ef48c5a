"""kernel trace target: a few obb_decode_nms calls on bench-like heads (run under rocprofv3 --kernel-trace --stats)"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
B = 1024
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
head = ops.forward(tiles)
torch.cuda.synchronize()
for _ in range(5):
    det, cnt = ops.decode_nms(head, 416, 416, 0.25, 0.7, 300)
torch.cuda.synchronize()
c = cnt.cpu().numpy()
print("kept/tile mean", c.mean(), "max", c.max())
