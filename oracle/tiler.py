"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the training-set tiler's per-tile label logic (Train_OBB.py:44-146) and of the
empty-tile selection (:167-190).  The reference function reads and writes image files through cv2 (absent offline) around this logic,
so it cannot be executed here: **parity unpinned**, restated line by line (pandas column expressions -> numpy on the same float64
values, same operation order) and checked by known answers."""
import numpy as np


def cov_frac(row8, x, y, ts):  # _cov_frac :60-70
    xs, ys = row8[0::2], row8[1::2]
    bx1, by1, bx2, by2 = min(xs), min(ys), max(xs), max(ys)
    ax = max(0, min(bx2, x + ts) - max(bx1, x))
    ay = max(0, min(by2, y + ts) - max(by1, y))
    inter = ax * ay
    area = max(1e-6, (bx2 - bx1) * (by2 - by1))
    return inter / area


def tile_image(H, W, labels, tile_size=128, overlap=50, thr=0.1):
    """labels [n, 9] (class + 8 pixel coords) -> (positives [(tile_id, x, y, rows [k, 9])], empties [(tile_id, x, y)])"""
    stride = tile_size - overlap
    assert stride > 0
    lab = np.asarray(labels, dtype=np.float64).reshape(-1, 9)
    pos, emp = [], []
    tile_id = 0
    for y in range(0, H, stride):            # :80
        for x in range(0, W, stride):        # :81
            if y + tile_size > H or x + tile_size > W:  # :82-83
                continue
            cxm = (lab[:, 1] + lab[:, 7]) / 2   # (x1 + x4) / 2   :86-89
            cym = (lab[:, 2] + lab[:, 8]) / 2   # (y1 + y4) / 2
            cand = lab[(cxm >= x) & (cxm < x + tile_size) & (cym >= y) & (cym < y + tile_size)].copy()
            if len(cand) > 0:                   # :91-93
                cov = np.array([cov_frac(r[1:], x, y, tile_size) for r in cand])
                cand = cand[cov >= thr].copy()
            if len(cand) > 0:                   # :95-101
                cand[:, 1::2] -= x
                cand[:, 2::2] -= y
                cand[:, 1:] = np.clip(cand[:, 1:], 0, tile_size)
                cand[:, 1:] /= tile_size
                pos.append((tile_id, x, y, cand))
            else:
                emp.append((tile_id, x, y))
            tile_id += 1                        # :121
    return pos, emp


def select_empty(n_empty, keep_fraction, rng_seed=42):  # :178-184
    k = int(round(keep_fraction * n_empty))
    rng = np.random.RandomState(rng_seed)
    idx = np.arange(n_empty)
    rng.shuffle(idx)
    return idx[:k]
