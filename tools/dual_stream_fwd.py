"""Experiment: two half-batches on two streams (separate model slots -> separate activation slabs) vs one full batch."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
w = make_weights.ensure("n", 12, 3, 0)
ms = [YOLO(w, imgsz=416) for _ in range(4)]
B = 256
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
def run(nsplit, reps=10):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    bounds = [(i * B // nsplit, (i + 1) * B // nsplit) for i in range(nsplit)]
    outs = [torch.zeros((hi - lo, 3549, 80), dtype=torch.float32, device="cuda") for lo, hi in bounds]
    def once():
        for i, ((lo, hi), st) in enumerate(zip(bounds, streams)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                ms[i]._ensure_active()
                ops.forward(tiles[lo:hi], out=outs[i])
        for st in streams: torch.cuda.current_stream().wait_stream(st)
    for _ in range(3): once()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    print(f"nsplit={nsplit}: {(time.perf_counter() - t) / reps * 1e3:.3f} ms per {B} tiles", flush=True)
for n in (1, 2, 4, 1):
    run(n)
