"""Seeded synthetic inputs shared by the golden generator and the parity tests.

Rotated boxes: centres U[0,extent)^2, w,h ~ U(10,100) px, theta ~ U[0,pi/2), class drawn from the GeoMap val
histogram (57 % "Strike"), conf ~ U(0.001,1) as float32.  Corners follow Ultralytics' xywhr2xyxyxyxy
 evaluated in float32 and widened to float64, then an integer tile offset is added in
float64 -- exactly how Detect_OBB.py:229-240 builds global coordinates.
"""
import numpy as np

VAL_CLASS_P = np.array([0.06, 0.57, 0.08, 0.03, 0.06, 0.05, 0.04, 0.03, 0.03, 0.02, 0.02, 0.01])
VAL_CLASS_P = VAL_CLASS_P / VAL_CLASS_P.sum()


def xywhr_to_corners_f32(xywhr):
    x = np.asarray(xywhr, np.float32)
    cx, cy, w, h, r = (x[:, i] for i in range(5))
    c, s = np.cos(r).astype(np.float32), np.sin(r).astype(np.float32)
    v1x, v1y = (w / np.float32(2)) * c, (w / np.float32(2)) * s
    v2x, v2y = -(h / np.float32(2)) * s, (h / np.float32(2)) * c
    p = np.stack([cx + v1x + v2x, cy + v1y + v2y, cx + v1x - v2x, cy + v1y - v2y,
                  cx - v1x - v2x, cy - v1y - v2y, cx - v1x + v2x, cy - v1y + v2y], 1)
    return p.astype(np.float32)


def make_dets(seed, n, extent=4096.0, dup_frac=0.35, jitter=6.0, tile=416):
    """-> boxes f64 [n,8], cls i32 [n], conf f64 [n] (float32-representable), xywhr f32 [n,5].
    A fraction of boxes are jittered near-duplicates of earlier ones so that NMS has real work."""
    rng = np.random.default_rng(seed)
    base = max(1, int(round(n * (1.0 - dup_frac))))
    cx = rng.uniform(0, extent, base)
    cy = rng.uniform(0, extent, base)
    w = rng.uniform(10, 100, base)
    h = rng.uniform(10, 100, base)
    r = rng.uniform(0, np.pi / 2, base)
    cls = rng.choice(len(VAL_CLASS_P), size=base, p=VAL_CLASS_P)
    nd = n - base
    if nd > 0:
        src = rng.integers(0, base, nd)
        cx = np.concatenate([cx, cx[src] + rng.normal(0, jitter, nd)])
        cy = np.concatenate([cy, cy[src] + rng.normal(0, jitter, nd)])
        w = np.concatenate([w, w[src] * rng.uniform(0.85, 1.15, nd)])
        h = np.concatenate([h, h[src] * rng.uniform(0.85, 1.15, nd)])
        r = np.concatenate([r, np.clip(r[src] + rng.normal(0, 0.05, nd), 0, np.pi / 2 - 1e-3)])
        same = rng.uniform(size=nd) < 0.85
        cls = np.concatenate([cls, np.where(same, cls[src], rng.choice(len(VAL_CLASS_P), size=nd, p=VAL_CLASS_P))])
    perm = rng.permutation(n)
    cx, cy, w, h, r, cls = cx[perm], cy[perm], w[perm], h[perm], r[perm], cls[perm]
    conf = rng.uniform(0.001, 1.0, n).astype(np.float32)
    # a few exact conf ties to exercise sort stability
    if n >= 8:
        t = rng.integers(0, n, max(2, n // 16))
        conf[t] = conf[t[0]]
    # local coords inside a tile + integer offset (exact in f64)
    ox = np.floor(cx / (tile - 100)) * (tile - 100)
    oy = np.floor(cy / (tile - 100)) * (tile - 100)
    xywhr = np.stack([cx - ox, cy - oy, w, h, r], 1).astype(np.float32)
    loc = xywhr_to_corners_f32(xywhr).astype(np.float64)
    boxes = loc.copy()
    boxes[:, 0::2] += ox[:, None]
    boxes[:, 1::2] += oy[:, None]
    return boxes, cls.astype(np.int32), conf.astype(np.float64), xywhr


def dets_to_tuples(boxes, cls, conf, tag0=0):
    """11-tuples as the reference builds them (Detect_OBB.py:256-262); the angle slot carries a unique tag."""
    out = []
    for i in range(boxes.shape[0]):
        out.append(tuple(float(v) for v in boxes[i]) + (int(cls[i]), float(conf[i]), float(tag0 + i)))
    return out


# ---------------------------------------------------------------- detect_symbols stub-model support

def coord_image(H, W):
    """BGR uint8 image whose pixel (y,x) encodes its own coordinates, so a stub model can recover a crop's origin."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = np.zeros((H, W, 3), np.uint8)
    img[..., 0] = xx & 255
    img[..., 1] = (xx >> 8) | ((yy >> 8) << 4)
    img[..., 2] = yy & 255
    return img


def crop_origin(crop):
    p = crop[0, 0]
    x0 = int(p[0]) | ((int(p[1]) & 15) << 8)
    y0 = int(p[2]) | ((int(p[1]) >> 4) << 8)
    return x0, y0


def stub_local_dets(x0, y0, h, w, seed, n_max=24):
    """Deterministic per-tile local detections: xywhr f32 [n,5] -> (corners f32 [n,8], cls i32, conf f32).
    Boxes deliberately straddle the border-filter margins and include near duplicates."""
    rng = np.random.default_rng([seed, x0, y0, h, w])
    n = int(rng.integers(0, n_max + 1))
    if n == 0:
        return np.zeros((0, 8), np.float32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    base = max(1, (2 * n) // 3)
    cx = rng.uniform(-5, w + 5, base)
    cy = rng.uniform(-5, h + 5, base)
    # snap some centres exactly onto the inclusive margins (10 / 20 px) to exercise <= vs <
    snap = rng.uniform(size=base) < 0.15
    cx = np.where(snap, rng.choice([10.0, 20.0, max(w - 10.0, 0), max(w - 20.0, 0)], size=base), cx)
    ww = rng.uniform(10, 60, base)
    hh = rng.uniform(10, 60, base)
    r = rng.uniform(0, np.pi / 2, base)
    cls = rng.choice(len(VAL_CLASS_P), size=base, p=VAL_CLASS_P)
    nd = n - base
    if nd > 0:
        src = rng.integers(0, base, nd)
        cx = np.concatenate([cx, cx[src] + rng.normal(0, 2.0, nd)])
        cy = np.concatenate([cy, cy[src] + rng.normal(0, 2.0, nd)])
        ww = np.concatenate([ww, ww[src]])
        hh = np.concatenate([hh, hh[src]])
        r = np.concatenate([r, r[src]])
        cls = np.concatenate([cls, cls[src]])
    xywhr = np.stack([cx, cy, ww, hh, r], 1).astype(np.float32)
    conf = rng.uniform(0.25, 1.0, n).astype(np.float32)
    return xywhr_to_corners_f32(xywhr), cls.astype(np.int32), conf


class _StubDet:
    def __init__(self, pts, c, s):
        self.xyxyxyxy = np.asarray(pts, np.float32).reshape(1, 4, 2)
        self.cls = np.array([c], np.float32)
        self.conf = np.array([s], np.float32)


class _StubResults:
    def __init__(self, dets):
        self.obb = dets


class StubModel:
    """Duck-types the slice of the Ultralytics API that Detect_OBB.py:81-83,228-231 touches."""

    def __init__(self, seed):
        self.seed = seed
        self.calls = []

    def __call__(self, crop, conf=0.25):
        x0, y0 = crop_origin(crop)
        h, w = crop.shape[:2]
        self.calls.append((x0, y0, h, w))
        pts, cls, sc = stub_local_dets(x0, y0, h, w, self.seed)
        return [_StubResults([_StubDet(pts[i], cls[i], sc[i]) for i in range(len(cls))])]
