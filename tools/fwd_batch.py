import sys, time, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
import os
prec = os.environ.get("OBB_PREC", "f16")
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416, precision=prec)
print("precision", prec)
for B in [int(v) for v in os.environ.get('OBB_BATCHES', '8,16,32,64,128,256,512').split(',')]:
    tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
    for _ in range(3): ops.forward(tiles)
    torch.cuda.synchronize()
    reps = max(2, 1024 // B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.forward(tiles)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"B={B:4d} forward {ms:8.3f} ms  {ms/B*1e3:7.2f} us/tile  {B/ms*1e3:9.0f} tiles/s  {B*2.785e9/ms/1e9:7.1f} TFLOP/s", flush=True)
