import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle.yolo11_obb import Yolo11OBB
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
m = Yolo11OBB("n", nc=12, ch=3, seed=0)
ops.model_load(m.to_blob())
x = np.random.default_rng(1).integers(0, 256, (2, 416, 416, 3), dtype=np.uint8)
taps = {}
m.forward_raw(x, "f16", taps)
ops.forward(torch.as_tensor(x).cuda()); torch.cuda.synchronize()
for name in ["model.0", "model.1", "model.2.cv1", "model.2.m.0.cv1", "model.2.cv2", "model.3", "model.9.cv1"]:
    got = ops.debug_activation(name, 2, 416, 416).cpu(); exp = taps[name].permute(0, 2, 3, 1)
    d = (got - exp).abs()
    print(name, "max", float(d.max()), "mean", float(d.mean()), "frac bad", float((d > 0.05).float().mean()))
    if name == "model.0":
        bad = (d > 0.05).nonzero()
        print(" first bad idx", bad[:5].tolist(), "got", got[tuple(bad[0])].item() if len(bad) else None, "exp", exp[tuple(bad[0])].item() if len(bad) else None)
        print(" bad by channel", (d > 0.05).float().mean((0, 1, 2)).tolist())
        print(" bad by y%13", [(float((d[:, y::13] > 0.05).float().mean())) for y in range(13)][:13])
