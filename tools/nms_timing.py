import sys, time, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
B = 256
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
head = ops.forward(tiles)
pred = ops.decode(head, 416, 416)
conf = pred[..., 4:16].amax(-1)
for c in (0.25, 0.5, 0.9):
    print("conf", c, "cands/tile mean", float((conf > c).sum(1).float().mean()), "max", int((conf > c).sum(1).max()))
for c in (0.25, 0.5, 0.9, 0.001):
    for _ in range(2): ops.decode_nms(head, 416, 416, c, 0.7, 300)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): det, cnt = ops.decode_nms(head, 416, 416, c, 0.7, 300)
    torch.cuda.synchronize(); print("decode_nms conf", c, (time.time() - t) / 5 * 1e3, "ms", "kept/tile", float(cnt.float().mean()))
