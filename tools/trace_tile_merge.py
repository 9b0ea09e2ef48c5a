"""decode + NMS + obb_tile_survivors on the bench heads (per-tile merge: k_merge_segments[_wave]); run under rocprofv3 --kernel-trace --stats"""
import sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "."); 
import make_weights
import bench
import oriented_object_detection_amd  # noqa
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO
from oriented_object_detection_amd.detect import Config as _C
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416, precision="f32")
B, A = 1024, 3549
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
cmax = torch.empty((B, A), dtype=torch.float32, device="cuda")
head = ops.forward(tiles, cmax=cmax)
cfg = _C()
rects = torch.as_tensor(bench.synthetic_rects(B)).cuda()
tile_ids = torch.arange(B, dtype=torch.int32, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(reps):
    det, cnt = ops.decode_nms(head, 416, 416, cfg.conf_predict, cfg.iou_nms, cfg.max_det, zero=False, cmax=cmax)
    margin = cfg.margin_for(416) if cfg.APPLY_BORDER_FILTER else 0
    rec, _, n = ops.tile_survivors(det, cnt, None, tile_ids, rects, margin, cfg.iou_threshold, cfg.strike_cls)
torch.cuda.synchronize()
c = cnt.cpu().numpy()
print("rows/tile: mean", c.mean(), "n>64:", int((c > 64).sum()), "n>150:", int((c > 150).sum()), "n==max:", int((c == cfg.max_det).sum()), "records", int(n))
