"""Durations of the k_fused_chain launches of one forward: `rocprofv3 --kernel-trace ... -- python3 tools/fused_timing.py run`, then
`python3 tools/fused_timing.py report <dir>`."""
import csv, glob, sys
import numpy as np
if sys.argv[1] == "report":
    f = glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ker = [r for r in rows if "k_fused_chain" in r["Kernel_Name"]]
    n = len(ker) // 3
    print(sys.argv[2], " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in ker[-n:]))
else:
    import torch
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    import make_weights
    import oriented_object_detection_amd
    from oriented_object_detection_amd import ops
    from oriented_object_detection_amd.model import YOLO
    B = 256
    m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
    tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
    for _ in range(3):
        ops.forward(tiles)
    torch.cuda.synchronize()
