"""Stage-by-stage timing of one bench step (HIP events, averaged)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import make_weights
import oriented_object_detection_amd
from oriented_object_detection_amd import ops, detect as D
from oriented_object_detection_amd.model import YOLO
sys.path.insert(0, ".")
import bench
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
cfg = D.Config(tile_sizes=(416,), overlaps=(100,))
B = 256
dev = torch.device("cuda", 0)
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).to(dev)
rects_dev = torch.as_tensor(bench.synthetic_rects(B)).to(dev)
tile_ids = torch.arange(0, B, dtype=torch.int32, device=dev)
acc = {}
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0) + (time.perf_counter() - t); return r
for it in range(12):
    if it == 2: acc.clear()
    head = T("forward", lambda: ops.forward(tiles))
    det, cnt = T("decode_nms", lambda: ops.decode_nms(head, 416, 416, 0.25, 0.7, 300))
    md = 300
    def glue1():
        valid = (torch.arange(md, device=dev)[None, :] < cnt[:, None]).reshape(-1)
        rows = torch.nonzero(valid).squeeze(1)
        d = det.reshape(-1, 7)[rows].contiguous(); slot = (rows // md).long()
        return d, slot
    d, slot = T("glue_compact", glue1)
    _, pts = T("results", lambda: ops.results(d, None))
    rec = T("tile_records", lambda: D._tile_records(pts, d[:, 5].int().contiguous(), d[:, 4].contiguous(), tile_ids[slot].contiguous(), rects_dev, cfg, 416))
    ds = T("records_to_detset", lambda: D.records_to_detset(rec, rects_dev, cfg, 416))
    merged = T("final_merge", lambda: D.merge_detections_device(ds, 0.4))
for k, v in acc.items(): print(f"{k:20s} {v/10*1e3:8.3f} ms")
print("sum", sum(acc.values()) / 10 * 1e3, "ms; records", len(rec), "final", len(merged[0]))
