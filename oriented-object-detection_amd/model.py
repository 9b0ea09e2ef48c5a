"""`YOLO(...)` / `model(img, conf=...)` / `results[0].obb` -- the slice of the Ultralytics API that Detect_OBB.py uses
(:16, :26, :81-83, :228-231), backed by libobbhip.so.  Same names, argument meaning and result accessors; the
network, decode, NMS and result construction all run as HIP kernels.
"""
import os

import numpy as np
import torch

from . import ops


class OBB:
    """Mirror of ultralytics.engine.results.OBB for the accessors the reference touches: iteration / indexing yield
    1-row views exposing .xyxyxyxy (1,4,2), .cls (1,), .conf (1,); plus .xywhr, .data, len()."""

    def __init__(self, data, pts):
        self.data = data  # [n,7] (x, y, w, h, theta, conf, cls)  float32, device
        self._pts = pts   # [n,8]

    def __len__(self):
        return int(self.data.shape[0])

    def __getitem__(self, i):
        if isinstance(i, int):
            i = slice(i, i + 1) if i >= 0 else slice(len(self) + i, len(self) + i + 1)
        return OBB(self.data[i], self._pts[i])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def xywhr(self):
        return self.data[:, :5]

    @property
    def conf(self):
        return self.data[:, 5]

    @property
    def cls(self):
        return self.data[:, 6]

    @property
    def xyxyxyxy(self):
        return self._pts.view(-1, 4, 2)

    def cpu(self):
        return OBB(self.data.cpu(), self._pts.cpu())


class Results:
    def __init__(self, obb, orig_shape, names=None):
        self.obb = obb
        self.orig_shape = orig_shape
        self.names = names or {}
        self.boxes = None

    def __len__(self):
        return len(self.obb)


class YOLO:
    """model = YOLO(weights); results = model(crop_bgr_uint8, conf=0.25)

    weights: path to an "OBBW" blob, raw bytes, or any object with .to_blob() (the trained .pt files of the reference are
    Google-Drive links and need ultralytics to un-pickle: not available offline).
    imgsz: the size the checkpoint was trained at (Ultralytics reads it from the checkpoint: 416 / 128 here).
    precision: the arithmetic of the forward.
      "f32" (default)  fp32 weights, activations and accumulation end to end (exact-f32 MFMA): what `model(net_input, conf=...)` computes in
                       Detect_OBB.py:79-83 (Ultralytics' half=False).  Detections agree with the fp32 reference pipeline one to one (same count,
                       classes and order up to confidence near-ties; confidences within 2e-4, corners within 0.1 px: tests/test_gpu_fp32.py).
      "f16"            opt-in fast mode (4-5x the tiles/s): fp16 storage, fp32 accumulation -- what Ultralytics' half=True does.  Head logits
                       move by ~1e-2, so candidates within that of a hard threshold (conf 0.25 / 0.70, NMS 0.7, merge 0.4) flip: >= 95 % of the
                       detections of a dense scene match the fp32 pipeline (IoU >= 0.5, |d conf| <= 0.12), not all of them.
      "bf16"           like "f16" with bf16 storage: 8 significand bits, ~78 % agreement; only for experiments."""

    def __init__(self, weights, imgsz=416, device=None, precision="f32", names=None, engine_options=None):
        if not torch.cuda.is_available():
            raise RuntimeError("YOLO: no HIP device visible; this implementation has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        if hasattr(weights, "to_blob"):
            blob = weights.to_blob()
        elif isinstance(weights, (bytes, bytearray)):
            blob = bytes(weights)
        else:
            if not os.path.exists(weights):
                raise FileNotFoundError(f"{weights}: weight blob not found (no download is ever attempted)")
            with open(weights, "rb") as f:
                blob = f.read()
        self._blob = blob
        self.precision = precision
        self.engine_options = dict(engine_options or {})  # ops.MODEL_OPTIONS switches (A/B measurements and parity tests; all default on)
        self.imgsz = int(imgsz)
        self.names = names or {}
        self._load()

    _next_slot = {}   # per device: every YOLO object owns one model slot of the device's libobbhip context ...
    _free_slots = {}  # ... and hands it back in close() / __del__ (the library keeps 64 slots per context)

    def _load(self):
        idx = self.device.index
        free = YOLO._free_slots.setdefault(idx, [])
        if free:
            self._slot = free.pop()
        else:
            self._slot = YOLO._next_slot.get(idx, 0)
            YOLO._next_slot[idx] = self._slot + 1
        with torch.cuda.device(self.device):
            ops.select_model(self._slot, self.device)
            ops.model_load(self._blob, self.device, self.precision, **self.engine_options)
            info = ops.model_info(self.imgsz, self.imgsz, self.device)
        self.nc, self.ch = info["nc"], info["ch"]

    def _ensure_active(self):
        if self._slot is None:
            raise RuntimeError("YOLO: this model has been closed")
        ops.select_model(self._slot, self.device)

    def close(self):
        """Releases the weights, activation slabs and captured graphs of this model on the device; the object is unusable afterwards."""
        slot, self._slot = getattr(self, "_slot", None), None
        if slot is None:
            return
        try:
            with torch.cuda.device(self.device):
                ops.model_unload(slot, self.device)
            YOLO._free_slots.setdefault(self.device.index, []).append(slot)
        except Exception:  # interpreter shutdown: the context may already be gone
            pass

    def __del__(self):
        self.close()

    # ------------------------------------------------------------------ batched device API
    def predict_tiles(self, tiles, conf=0.25, iou=0.7, max_det=300, zero=True):
        """tiles uint8 [B,h,w,ch] on device, already letterboxed -> (det [B,max_det,7] (x,y,w,h,conf,cls,theta), count [B])"""
        self._ensure_active()
        B, h, w, ch = tiles.shape
        # cmax: the largest class logit per anchor -- the dense candidate gate of the NMS kernel.  Only the fp32 plan writes it inside its
        # fused class tails; the 16-bit plans would pay an extra pass over the head rows for it (measured: +0.15 ms per 1024 tiles)
        cmax = None
        if self.precision in ("f32", "fp32"):
            cmax = torch.empty((B, ops.model_info(h, w, tiles.device)["anchors"]), dtype=torch.float32, device=tiles.device)
        head = ops.forward(tiles, cmax=cmax)
        return ops.decode_nms(head, h, w, conf, iou, max_det, zero=zero, cmax=cmax)

    # ------------------------------------------------------------------ Ultralytics-shaped API
    def __call__(self, source, conf=0.25, iou=0.7, max_det=300, **kwargs):
        return self.predict(source, conf=conf, iou=iou, max_det=max_det, **kwargs)

    def predict(self, source, conf=0.25, iou=0.7, max_det=300, **kwargs):
        self._ensure_active()
        imgs = source if isinstance(source, (list, tuple)) else [source]
        out = []
        with torch.cuda.device(self.device):
            for im in imgs:
                if isinstance(im, np.ndarray):
                    if im.dtype != np.uint8 or im.ndim != 3:
                        raise ValueError("predict: expected an HxWxC uint8 array (as Detect_OBB.py passes crops)")
                    im = torch.as_tensor(np.ascontiguousarray(im)).to(self.device)
                H, W, C = im.shape
                if C == 3 and self.ch == 4:  # run_inference_on_crop (Detect_OBB.py:76-77): net_input = build_multich(crop_bgr, 4)
                    im = ops.build_multich(im.contiguous()[None])[0]
                    C = 4
                if C != self.ch:
                    raise ValueError(f"predict: model expects {self.ch} channels, got {C}")
                lbimg, p = ops.letterbox(im.contiguous(), 0, 0, W, H, self.imgsz)
                det, count = self.predict_tiles(lbimg[None], conf, iou, max_det)
                n = int(count[0].item())
                rows = det[0, :n].contiguous()
                lb = torch.tensor([[p["gain"], p["pad_x"], p["pad_y"]]], dtype=torch.float32, device=self.device).repeat(max(n, 1), 1)[:n]
                xywhr, pts = ops.results(rows, lb.contiguous() if n else None)
                data = torch.cat([xywhr, rows[:, 4:6]], 1)  # (x,y,w,h,theta,conf,cls)
                out.append(Results(OBB(data, pts), (H, W), self.names))
        return out
