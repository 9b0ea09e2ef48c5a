"""Host-side mirror of Detect_OBB.py's own functions -- same names, argument meaning and error behaviour -- with the
arithmetic routed through libobbhip.so:

    compute_polygon_iou            Detect_OBB.py:144-154
    merge_detections               Detect_OBB.py:176-200   (sorts the caller's list in place, returns the kept sub-list)
    cross_scale_consensus_filter   Detect_OBB.py:347-423
    detect_symbols                 Detect_OBB.py:202-266
    process_image                  Detect_OBB.py:268-345   (control flow; drawing / xlsx are out of scope -> CSV rows)

A detection is the reference's 11-tuple (x1,y1,x2,y2,x3,y3,x4,y4, cls_id, conf, angle).  The *_device variants keep
everything in HBM as SoA tensors (DetSet) and are what bench.py and the multi-GPU path use.
"""
import csv
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops
from .model import YOLO


# ---------------------------------------------------------------- configuration (Detect_OBB.py:20-72)
CLASS_NAMES = {0: "Landslide 1", 1: "Strike", 2: "Spring 1", 3: "Minepit 1", 4: "Hillside", 5: "Feuchte", 6: "Torf",
               7: "Bergsturz", 8: "Landslide 2", 9: "Spring 2", 10: "Spring 3", 11: "Minepit 2"}


@dataclass
class Config:
    calculate_metrics: bool = False            # :23
    tile_sizes: tuple = (128, 416)             # :24
    overlaps: tuple = (30, 100)                # :25
    channels: int = 3                          # :28
    MAP_MIN_SCORE: float = 0.001               # :34
    iou_thr: float = 0.25                      # :35 (metrics)
    iou_threshold: float = 0.4                 # :36 (merge)
    APPLY_BORDER_FILTER: bool = True           # :38
    MARGIN_128: int = 10                       # :39
    MARGIN_416: int = 20                       # :40
    CLASS_NAMES: dict = field(default_factory=lambda: dict(CLASS_NAMES))
    conf_predict: float = 0.25                 # :83
    conf_metrics: float = 0.001                # :81
    iou_nms: float = 0.7                       # Ultralytics predict default
    max_det: int = 300                         # Ultralytics predict default

    def margin_for(self, tile_size):           # :156-157
        return self.MARGIN_128 if tile_size <= 128 else self.MARGIN_416

    @property
    def strike_cls(self):
        for k, v in self.CLASS_NAMES.items():
            if v == "Strike":
                return k
        return -1


DEFAULT = Config()


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class DetSet:
    """SoA detections resident on the device: boxes f64 [n,8], cls i32 [n], conf f64 [n], angle f64 [n].
    `count` (optional device int32[1]): only the first count rows are valid -- the tensors then have capacity rows and the number is read
    from the device the first time somebody asks for it (len(), to_tuples(), ...), not when the set is produced."""

    def __init__(self, boxes, cls, conf, angle, count=None):
        self._b, self._c, self._f, self._a, self.count, self._n = boxes, cls, conf, angle, count, None

    def __len__(self):
        if self._n is None:
            self._n = int(self.count.item()) if self.count is not None else int(self._c.shape[0])
        return self._n

    def _trimmed(self, t):
        return t if self.count is None or t.shape[0] == len(self) else t[:len(self)]

    boxes = property(lambda self: self._trimmed(self._b))
    cls = property(lambda self: self._trimmed(self._c))
    conf = property(lambda self: self._trimmed(self._f))
    angle = property(lambda self: self._trimmed(self._a))

    @staticmethod
    def empty(device=None):
        d = device or _dev()
        return DetSet(torch.zeros((0, 8), dtype=torch.float64, device=d), torch.zeros(0, dtype=torch.int32, device=d),
                      torch.zeros(0, dtype=torch.float64, device=d), torch.zeros(0, dtype=torch.float64, device=d))

    @staticmethod
    def from_tuples(dets, device=None):
        d = device or _dev()
        if not dets:
            return DetSet.empty(d)
        arr = np.array([t[:8] + (t[9], t[10] if len(t) > 10 else 0.0) for t in dets], np.float64)
        cls = np.array([int(t[8]) for t in dets], np.int32)
        return DetSet(torch.as_tensor(arr[:, :8]).contiguous().to(d), torch.as_tensor(cls).to(d),
                      torch.as_tensor(arr[:, 8]).contiguous().to(d), torch.as_tensor(arr[:, 9]).contiguous().to(d))

    def select(self, idx):
        idx = idx.long()
        return DetSet(self.boxes[idx].contiguous(), self.cls[idx].contiguous(), self.conf[idx].contiguous(), self.angle[idx].contiguous())

    @staticmethod
    def cat(sets):
        sets = [s for s in sets]
        if not sets:
            return DetSet.empty()
        return DetSet(torch.cat([s.boxes for s in sets]), torch.cat([s.cls for s in sets]), torch.cat([s.conf for s in sets]),
                      torch.cat([s.angle for s in sets]))

    def to_tuples(self):
        b, c, s, a = self.boxes.cpu().numpy(), self.cls.cpu().numpy(), self.conf.cpu().numpy(), self.angle.cpu().numpy()
        return [tuple(float(v) for v in b[i]) + (int(c[i]), float(s[i]), float(a[i])) for i in range(len(c))]


# ---------------------------------------------------------------- S2
def compute_polygon_iou(box1, box2):
    """Compute IoU between two rotated bounding boxes (8 floats each).  Invalid polygons give 0.0."""
    d = _dev()
    a = torch.tensor([[float(v) for v in box1[:8]]], dtype=torch.float64, device=d)
    b = torch.tensor([[float(v) for v in box2[:8]]], dtype=torch.float64, device=d)
    return float(ops.poly_iou_pairs(a, b).item())


# ---------------------------------------------------------------- S3
def build_multich(bgr, out_channels=4):
    """Detect_OBB.py:87-133: HxWx3 BGR uint8 (numpy array or device tensor) -> HxWx3 unchanged, or HxWx4 = RGB + distance-transform
    edge channel (computed on the device, returned in the container type that came in)."""
    assert out_channels in (3, 4), f"Unsupported out_channels={out_channels}"
    if out_channels == 3:
        return np.ascontiguousarray(bgr) if isinstance(bgr, np.ndarray) else bgr.contiguous()
    t = torch.as_tensor(np.ascontiguousarray(bgr)).cuda() if isinstance(bgr, np.ndarray) else bgr
    out = ops.build_multich(t.contiguous()[None])[0]
    return out.cpu().numpy() if isinstance(bgr, np.ndarray) else out


def merge_detections_device(ds, iou_threshold=0.5):
    """-> (kept DetSet in confidence order, order tensor: sorted position -> input row)"""
    if len(ds) == 0:
        return ds, torch.zeros(0, dtype=torch.int32, device=ds.cls.device)
    order, keep, _ = ops.merge_detections(ds.boxes, ds.cls, ds.conf, iou_threshold)
    b, c, f, a, cnt = ops.select_kept(order, keep, ds.boxes, ds.cls, ds.conf, ds.angle)  # kept rows in merge order; their number stays on the device
    return DetSet(b, c, f, a, count=cnt), order


def merge_detections(detections, iou_threshold=0.5):
    """Merge overlapping detections while considering confidence and class types.
    Like the reference (Detect_OBB.py:183) this sorts `detections` in place (stable, confidence descending)."""
    if not detections:
        return []
    ds = DetSet.from_tuples(detections)
    order, keep, _ = ops.merge_detections(ds.boxes, ds.cls, ds.conf, iou_threshold)
    order, keep = order.cpu().numpy(), keep.cpu().numpy().astype(bool)
    srt = [detections[i] for i in order]
    detections[:] = srt
    return [d for d, k in zip(srt, keep) if k]


# ---------------------------------------------------------------- S4
CONS_IOU_PARTNER, CONS_LOW, CONS_HIGH = 0.40, 0.25, 0.70  # Detect_OBB.py:349-351


def cross_scale_consensus_filter_device(sets_by_scale):
    scales = sorted(sets_by_scale.keys())
    allset = DetSet.cat([sets_by_scale[s] for s in scales])
    off = [0]
    for s in scales:
        off.append(off[-1] + len(sets_by_scale[s]))
    if len(allset) == 0:
        return allset
    idx, n = ops.consensus(allset.boxes, allset.cls, allset.conf, off, CONS_IOU_PARTNER, CONS_LOW, CONS_HIGH)
    return allset.select(idx[: int(n.item())])


def cross_scale_consensus_filter(dets_by_scale):
    scales = sorted(dets_by_scale.keys())
    flat, off = [], [0]
    for s in scales:
        flat.extend(dets_by_scale[s])
        off.append(len(flat))
    if not flat:
        return []
    ds = DetSet.from_tuples(flat)
    idx, n = ops.consensus(ds.boxes, ds.cls, ds.conf, off, CONS_IOU_PARTNER, CONS_LOW, CONS_HIGH)
    return [flat[i] for i in idx[: int(n.item())].cpu().numpy()]


# ---------------------------------------------------------------- S5
USE_TILE_SURVIVORS = True  # False: the stand-alone kernels glued by host-side compactions (kept as the A/B reference of the fused path)


class TileRecords:
    """Per-tile survivors in the exchange format of the multi-GPU path (48 B/record):
    tile index, class, float32 confidence and the 8 float32 LOCAL corners.  Global float64 coordinates and the strike
    angle are re-derived exactly at the consumer (records_to_detset), so nothing is lost by shipping float32.
    Two carriers: separate arrays (tile, cls, conf, pts), or the packed int32 [capacity, 12] rows of obb_tile_survivors with their
    count still on the device (`packed`, `count`): the count is read once, when the length or a column is first needed."""

    def __init__(self, tile=None, cls=None, conf=None, pts=None, packed=None, count=None):
        self._cols = None if tile is None else (tile, cls, conf, pts)  # i32[n], i32[n], f32[n], f32[n,8]
        self.packed, self.count, self._n = packed, count, None

    @staticmethod
    def from_packed(packed, count):
        return TileRecords(packed=packed, count=count)

    def __len__(self):
        if self._n is None:
            self._n = int(self._cols[0].shape[0]) if self._cols is not None else int(self.count.item())
        return self._n

    def _col(self, i):
        if self._cols is None:
            self._cols = TileRecords.unpack(self.packed[:len(self)])._cols
        return self._cols[i]

    tile = property(lambda self: self._col(0))
    cls = property(lambda self: self._col(1))
    conf = property(lambda self: self._col(2))
    pts = property(lambda self: self._col(3))

    @property
    def device(self):
        return self.packed.device if self.packed is not None else self._cols[0].device

    @staticmethod
    def empty(device):
        return TileRecords(torch.zeros(0, dtype=torch.int32, device=device), torch.zeros(0, dtype=torch.int32, device=device),
                           torch.zeros(0, dtype=torch.float32, device=device), torch.zeros((0, 8), dtype=torch.float32, device=device))

    def pack(self):
        """-> int32 [n,12] (bit-exact container for the all-gather)"""
        if self.packed is not None:
            return self.packed[:len(self)]
        n = len(self)
        buf = torch.zeros((n, 12), dtype=torch.int32, device=self.tile.device)
        buf[:, 0], buf[:, 1] = self.tile, self.cls
        buf[:, 2] = self.conf.contiguous().view(torch.int32)
        buf[:, 4:] = self.pts.contiguous().view(torch.int32)
        return buf

    @staticmethod
    def unpack(buf):
        return TileRecords(buf[:, 0].contiguous(), buf[:, 1].contiguous(), buf[:, 2].contiguous().view(torch.float32),
                           buf[:, 4:].contiguous().view(torch.float32))


def _tile_records(local_pts, cls, conf32, det_tile, rects_dev, cfg, tile_size):
    """Per-detection body of the tile loop (:229-262) + per-tile merge (:264) on device; inputs flat over tiles in tile
    order.  -> TileRecords of the survivors, in the reference's output order."""
    ntiles = rects_dev.shape[0]
    dev = local_pts.device
    margin = cfg.margin_for(tile_size) if cfg.APPLY_BORDER_FILTER else 0
    gb, ang, ins = ops.tile_postprocess(local_pts, cls, det_tile, rects_dev, margin, cfg.strike_cls)
    sel = torch.nonzero(ins).squeeze(1)
    if sel.numel() == 0:
        return TileRecords.empty(dev)
    gb, cls, conf32, det_tile, local_pts = gb[sel].contiguous(), cls[sel].contiguous(), conf32[sel].contiguous(), det_tile[sel].contiguous(), local_pts[sel]
    seg = torch.zeros(ntiles + 1, dtype=torch.int32, device=dev)
    seg[1:] = torch.cumsum(torch.bincount(det_tile.long(), minlength=ntiles), 0).int()
    order, keep = ops.merge_segments(gb, cls, conf32.double(), seg, cfg.iou_threshold)
    kept = order[keep.bool()].long()
    return TileRecords(det_tile[kept].contiguous(), cls[kept].contiguous(), conf32[kept].contiguous(), local_pts[kept].contiguous())


def records_to_detset(rec, rects_dev, cfg, tile_size):
    """Rebuild global float64 boxes + strike angles from exchange records (exact: float32 local + integer offset)."""
    if len(rec) == 0:
        return DetSet.empty(rec.device)
    if rec.packed is not None:  # one kernel over the packed rows (no column copies)
        return DetSet(*ops.records_to_dets(rec.packed, len(rec), rects_dev, cfg.strike_cls))
    gb, ang, _ = ops.tile_postprocess(rec.pts, rec.cls, rec.tile, rects_dev, 0, cfg.strike_cls)
    return DetSet(gb, rec.cls, rec.conf.double(), ang)


def predict_tile_records(model, tiles, rects_dev, tile_ids, lb, cfg, tile_size, conf):
    """forward -> decode -> Fast-NMS -> result construction -> border filter -> per-tile merge for one batch of
    letterboxed tiles.  tile_ids int32[B] (index into rects_dev), lb float[B,3] or None.  -> TileRecords."""
    dev = tiles.device
    md = cfg.max_det
    fused = USE_TILE_SURVIVORS and md <= 512
    det, cnt = model.predict_tiles(tiles, conf, cfg.iou_nms, md, zero=not fused)
    if fused:  # everything up to the exchange records on the device (obb_tile_survivors); the count is read when it is first needed
        margin = cfg.margin_for(tile_size) if cfg.APPLY_BORDER_FILTER else 0
        rec, _, n = ops.tile_survivors(det, cnt, lb, tile_ids, rects_dev, margin, cfg.iou_threshold, cfg.strike_cls)
        return TileRecords.from_packed(rec, n)
    valid = (torch.arange(md, device=dev)[None, :] < cnt[:, None]).reshape(-1)
    rows = torch.nonzero(valid).squeeze(1)  # tile-major, score order inside a tile
    if rows.numel() == 0:
        return TileRecords.empty(dev)
    d = det.reshape(-1, 7)[rows].contiguous()
    slot = (rows // md).long()
    xywhr, pts = ops.results(d, lb[slot].contiguous() if lb is not None else None)
    return _tile_records(pts, d[:, 5].int().contiguous(), d[:, 4].contiguous(), tile_ids[slot].contiguous(), rects_dev, cfg, tile_size)


def detect_symbols_records(image, model, tile_size, overlap, cfg=DEFAULT, conf=None, batch=256, tile_subset=None):
    """Batched device-resident tile loop for a libobbhip YOLO.  image: uint8 [H,W,C] tensor on the device.
    tile_subset: optional list of tile indices (the multi-GPU shard).  -> (TileRecords, rects_dev)"""
    H, W, C = image.shape
    conf = cfg.conf_metrics if (conf is None and cfg.calculate_metrics) else (cfg.conf_predict if conf is None else conf)
    rects = ops.tile_grid(H, W, tile_size, overlap)
    dev = image.device
    rects_dev = torch.as_tensor(rects).to(dev) if len(rects) else torch.zeros((0, 4), dtype=torch.int32, device=dev)
    todo = range(len(rects)) if tile_subset is None else tile_subset
    groups = {}
    for t in todo:
        x, y, x2, y2 = rects[t]
        groups.setdefault((int(y2 - y), int(x2 - x)), []).append(int(t))
    parts = []
    for (ch, cw), idxs in groups.items():  # one network shape per distinct crop shape
        p = ops.letterbox_shape(ch, cw, model.imgsz)
        for i0 in range(0, len(idxs), batch):
            part = idxs[i0:i0 + batch]
            pidx = torch.as_tensor(part, dtype=torch.int32, device=dev)
            multich = C == 3 and getattr(model, "ch", C) == 4  # 4-channel checkpoints: RGB + DT-edge channel built per crop (:76-77, :87-133)
            if ch == cw == model.imgsz:
                tiles, lb = ops.gather_tiles(image, rects_dev[pidx.long()].contiguous(), model.imgsz), None
                if multich:
                    tiles = ops.build_multich(tiles)
            else:
                if multich:  # the channel is computed on the raw crop, the letterbox comes after it (as in the reference)
                    crops = ops.build_multich(torch.stack([image[rects[t][1]:rects[t][3], rects[t][0]:rects[t][2]] for t in part]).contiguous())
                    tiles = torch.stack([ops.letterbox(crops[k].contiguous(), 0, 0, cw, ch, model.imgsz)[0] for k in range(len(part))])
                else:
                    tiles = torch.stack([ops.letterbox(image, *[int(v) for v in rects[t]], model.imgsz)[0] for t in part])
                lb = torch.tensor([[p["gain"], p["pad_x"], p["pad_y"]]], dtype=torch.float32, device=dev).repeat(len(part), 1)
            parts.append(predict_tile_records(model, tiles, rects_dev, pidx, lb, cfg, tile_size, conf))
    if not parts:
        return TileRecords.empty(dev), rects_dev
    if len(parts) == 1:
        return parts[0], rects_dev  # (stays in packed form, count on the device)
    rec = TileRecords(torch.cat([r.tile for r in parts]), torch.cat([r.cls for r in parts]), torch.cat([r.conf for r in parts]),
                      torch.cat([r.pts for r in parts]))
    if len(groups) > 1 and len(rec):  # restore the reference's tile visiting order (stable: keeps score order inside a tile)
        o = torch.sort(rec.tile, stable=True).indices
        rec = TileRecords(rec.tile[o].contiguous(), rec.cls[o].contiguous(), rec.conf[o].contiguous(), rec.pts[o].contiguous())
    return rec, rects_dev


def detect_symbols_device(image, model, tile_size, overlap, cfg=DEFAULT, conf=None, batch=256):
    """-> DetSet in the reference's output order (tile order, confidence order inside a tile)."""
    rec, rects_dev = detect_symbols_records(image, model, tile_size, overlap, cfg, conf, batch)
    return records_to_detset(rec, rects_dev, cfg, tile_size)


def detect_symbols(image, model, tile_size: int, overlap: int, cfg=DEFAULT):
    """Tiled detection.  Output format for each det: (x1,y1,x2,y2,x3,y3,x4,y4, cls_id, conf, angle).

    `model` may be a libobbhip YOLO (batched device path) or any callable with the Ultralytics result interface the
    reference consumes (`model(crop, conf=...)[0].obb` items with .xyxyxyxy/.cls/.conf): then tiles are visited one by
    one exactly like Detect_OBB.py:216-231 and only the per-detection geometry + per-tile merge run on the device."""
    if isinstance(model, YOLO):
        img = image if torch.is_tensor(image) else torch.as_tensor(np.ascontiguousarray(image)).to(model.device)
        return detect_symbols_device(img, model, tile_size, overlap, cfg).to_tuples()
    H, W = image.shape[:2]
    rects = ops.tile_grid(H, W, tile_size, overlap)
    conf = cfg.conf_metrics if cfg.calculate_metrics else cfg.conf_predict
    pts, cls, cf, tid = [], [], [], []
    for t, (x, y, x2, y2) in enumerate(rects):
        results = model(np.ascontiguousarray(image[y:y2, x:x2]), conf=conf)
        for det in results[0].obb:
            pts.append([float(v) for v in np.asarray(_host(det.xyxyxyxy[0])).flatten().tolist()])
            cls.append(int(_host(det.cls)[0]))
            cf.append(np.float32(_host(det.conf)[0]))
            tid.append(t)
    dev = _dev()
    if not pts:
        return []
    rects_dev = torch.as_tensor(rects).to(dev)
    rec = _tile_records(torch.tensor(pts, dtype=torch.float32, device=dev), torch.tensor(cls, dtype=torch.int32, device=dev),
                        torch.tensor(np.array(cf, np.float32), device=dev), torch.tensor(tid, dtype=torch.int32, device=dev),
                        rects_dev, cfg, tile_size)
    return records_to_detset(rec, rects_dev, cfg, tile_size).to_tuples()


def _host(v):
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


# ---------------------------------------------------------------- process_image (Detect_OBB.py:268-345)
def imread_bgr(path):
    """cv2.imread replacement (PNG/JPG via Pillow): HxWx3 uint8 in BGR order, or None when unreadable (:271-273)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except Exception:
        return None


def process_image_device(image, models, cfg=DEFAULT):
    """image: uint8 [H,W,C] device tensor.  -> dict(merged_for_pr=DetSet, merged_for_map=DetSet|None, dets_by_scale=...)"""
    sets = {}
    for tile_size, overlap, model in zip(cfg.tile_sizes, cfg.overlaps, models):
        sets[tile_size] = detect_symbols_device(image, model, tile_size, overlap, cfg)
    merged_for_map = None
    if cfg.calculate_metrics:  # :281-288
        merged_for_map, _ = merge_detections_device(DetSet.cat([sets[s] for s in sets]), cfg.iou_threshold)
    consensus = cross_scale_consensus_filter_device(sets)
    merged_for_pr, _ = merge_detections_device(consensus, cfg.iou_threshold)
    return {"merged_for_pr": merged_for_pr, "merged_for_map": merged_for_map, "dets_by_scale": sets}


def process_image(image_path, output_dir, models, cfg=DEFAULT):
    """Per-image orchestration.  Returns the final detection tuples (the reference stores them in all_dets_per_image);
    writes <name>.csv with the reference's xlsx columns when output_dir is given (drawing/xlsx are out of scope)."""
    image = imread_bgr(image_path) if isinstance(image_path, str) else image_path
    if image is None:
        print(f"[Warn] Could not read image: {image_path}")
        return None
    dimg = torch.as_tensor(np.ascontiguousarray(image)).to(models[0].device)
    out = process_image_device(dimg, models, cfg)
    rows = out["merged_for_pr"].to_tuples()
    if output_dir and isinstance(image_path, str):
        os.makedirs(output_dir, exist_ok=True)
        name = os.path.splitext(os.path.basename(image_path))[0] + ".csv"
        with open(os.path.join(output_dir, name), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Class", "X1", "Y1", "X2", "Y2", "X3", "Y3", "X4", "Y4", "Confidence", "Angle"])
            for r in rows:
                wr.writerow([cfg.CLASS_NAMES.get(r[8], f"Class{r[8]}")] + [repr(v) for v in r[:8]] + [repr(r[9]), repr(r[10])])
    return rows
