"""obb_build_multich on 1024 crops of 416 px; run under rocprofv3 --kernel-trace --stats"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
B = 1024
t = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
out = torch.empty((B, 416, 416, 4), dtype=torch.uint8, device="cuda")
for _ in range(3): ops.build_multich(t, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): ops.build_multich(t, out)
e1.record(); torch.cuda.synchronize()
print("build_multich 1024 crops: %.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
