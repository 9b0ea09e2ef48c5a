import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
for B in (64, 256, 1024):
    x = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
    for _ in range(2): ops.build_multich(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.build_multich(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("build_multich B=%d 416x416: %.2f ms (%.0f tiles/s)" % (B, dt * 1e3, B / dt))
