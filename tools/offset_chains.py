"""Experiment: two forward chains of 512 tiles each on two streams, in lockstep (what one obb_forward of 1024 tiles issues internally) vs
offset by half a forward (stream Y first runs a 256-tile forward, so that afterwards its HBM-bound front layers meet the other stream's
MFMA-bound head layers).  Prints ms per 1024 tiles for both."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd  # noqa: F401
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO

m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416, precision="f32")
ops._call("obb_set_option", ops.ctx(), b"fwd_split", 1)
rng = np.random.default_rng(0)
tx = torch.as_tensor(rng.integers(0, 256, (512, 416, 416, 3), dtype=np.uint8)).cuda()
ty = torch.as_tensor(rng.integers(0, 256, (512, 416, 416, 3), dtype=np.uint8)).cuda()
tq = ty[:256].contiguous()
sx, sy = torch.cuda.Stream(), torch.cuda.Stream()
hx = hy = hq = None
for st, t in ((sx, tx), (sy, ty), (sy, tq)):
    with torch.cuda.stream(st):
        for _ in range(3):
            h = ops.forward(t)
torch.cuda.synchronize()
with torch.cuda.stream(sx): hx = ops.forward(tx)
with torch.cuda.stream(sy): hy = ops.forward(ty); hq = ops.forward(tq)
torch.cuda.synchronize()


def run(offset, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset:
        with torch.cuda.stream(sy):
            ops.forward(tq, out=hq)
    for _ in range(n):
        with torch.cuda.stream(sx):
            ops.forward(tx, out=hx)
        with torch.cuda.stream(sy):
            ops.forward(ty, out=hy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for rep in range(3):
    a = run(False)
    b = run(True)
    # the offset run carries one extra 256-tile forward: 12.25 x 1024 tiles
    print(f"lockstep {a / 12:.3f} ms per 1024 tiles   offset {b / 12.25:.3f} ms per 1024 tiles", flush=True)
