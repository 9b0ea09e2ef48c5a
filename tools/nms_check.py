import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
from oracle import postproc as pp
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
B = 256
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
head = ops.forward(tiles)
det, cnt = ops.decode_nms(head, 416, 416, 0.25, 0.7, 300)
np.save("gpurun_out/nms_cnt_%s.npy" % os.environ.get("OBB_NMS_EXACT", "0"), cnt.cpu().numpy())
print("mode exact=%s kept/tile %.6f" % (os.environ.get("OBB_NMS_EXACT", "0"), float(cnt.float().mean())))
if os.path.exists("gpurun_out/nms_cnt_0.npy") and os.path.exists("gpurun_out/nms_cnt_1.npy"):
    a, b = np.load("gpurun_out/nms_cnt_0.npy"), np.load("gpurun_out/nms_cnt_1.npy")
    diff = np.nonzero(a != b)[0]
    print("tiles differing:", diff.tolist(), a[diff].tolist(), b[diff].tolist())
    pred = ops.decode(head, 416, 416).cpu()
    for t in diff[:4]:
        exp = pp.non_max_suppression(pred[t:t + 1].transpose(1, 2).contiguous(), 0.25, 0.7, 300, 12)[0]
        print(" tile", t, "oracle", exp.shape[0], "this mode", int(cnt[t]))
