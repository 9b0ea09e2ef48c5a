import sys, time, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
for n in (4096, 16384, 65536):
    b, c, s, _ = synth.make_dets(1, n, extent=4096.0)
    B, Cc, S = torch.tensor(b).cuda(), torch.tensor(c).cuda(), torch.tensor(s).cuda()
    for _ in range(2): ops.merge_detections(B, Cc, S, 0.4)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): o, k, nk = ops.merge_detections(B, Cc, S, 0.4)
    torch.cuda.synchronize(); print("merge", n, (time.time() - t) / 5 * 1e3, "ms kept", int(nk.item()), flush=True)
m = 1 << 22
b, _, _, _ = synth.make_dets(2, 1 << 16, extent=1000.0)
A = torch.tensor(b).cuda().repeat(m // len(b), 1); Bt = A.roll(1, 0).contiguous()
ops.poly_iou_pairs(A, Bt); torch.cuda.synchronize(); t = time.time()
for _ in range(5): out = ops.poly_iou_pairs(A, Bt)
torch.cuda.synchronize(); dt = (time.time() - t) / 5
print("pairs", m, dt * 1e3, "ms", m * 136 / dt / 1e9, "GB/s", "frac>0", float((out > 0).double().mean()))
