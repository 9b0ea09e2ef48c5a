import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
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
for c in (0.25, 0.9, 0.999):
    k = (conf > c).sum(1)
    print("conf", c, "cands/tile mean", float(k.float().mean()), "max", int(k.max()), "top5", sorted(k.tolist())[-5:])
def t_event(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("decode only us", t_event(lambda: ops.decode(head, 416, 416)))
for c in (0.25, 0.9, 0.999, 0.001):
    us = t_event(lambda: ops.decode_nms(head, 416, 416, c, 0.7, 300))
    det, cnt = ops.decode_nms(head, 416, 416, c, 0.7, 300)
    print("decode_nms conf", c, "%.1f us" % us, "kept/tile", float(cnt.float().mean()), "max", int(cnt.max()))
