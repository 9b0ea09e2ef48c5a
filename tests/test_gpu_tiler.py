"""f4 (second part): the training-set tiler (Train_OBB.py:44-146, :167-190) -- crops and label rows from the GPU kernels against the
numpy restatement (oracle/tiler.py; parity unpinned: the reference function is wrapped in cv2 file I/O) and known answers."""
import numpy as np
import pytest
import torch

from oracle import tiler as ot

pytestmark = pytest.mark.gpu


def _labels(seed, n, H, W):
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
    w, h, a = rng.uniform(6, 90, n), rng.uniform(6, 90, n), rng.uniform(0, np.pi, n)
    dx, dy = np.stack([-w, w, w, -w], 1) / 2, np.stack([-h, -h, h, h], 1) / 2
    px = cx[:, None] + dx * np.cos(a)[:, None] - dy * np.sin(a)[:, None]
    py = cy[:, None] + dx * np.sin(a)[:, None] + dy * np.cos(a)[:, None]
    lab = np.zeros((n, 9))
    lab[:, 0] = rng.integers(0, 12, n)
    lab[:, 1::2], lab[:, 2::2] = px, py
    return lab


@pytest.mark.parametrize("H,W,ts,ov,n", [(700, 900, 128, 50, 400), (416, 416, 416, 100, 30), (1000, 640, 416, 100, 250), (300, 300, 128, 50, 0), (100, 100, 128, 50, 5)])
def test_tiler_matches_restatement(H, W, ts, ov, n):
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import tiler
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lab = _labels(n + 1, n, H, W)
    pos, emp = tiler.tile_image(img, lab, ts, ov)
    epos, eemp = ot.tile_image(H, W, lab, ts, ov, 0.1)
    assert [(p["tile_id"], p["x"], p["y"]) for p in pos] == [(t, x, y) for t, x, y, _ in epos]
    assert [(e["tile_id"], e["x"], e["y"]) for e in emp] == list(eemp)
    for p, (_, x, y, rows) in zip(pos, epos):
        assert np.array_equal(p["labels"], rows)                       # same float64 operations in the same order: bit-identical
        assert np.array_equal(p["crop"].cpu().numpy(), img[y:y + ts, x:x + ts])
    assert tiler.enumerate_full_tiles(H, W, ts, ov) == [(i, x, y) for i, (x, y) in enumerate((x, y) for y in range(0, H, ts - ov) for x in range(0, W, ts - ov)
                                                                                             if y + ts <= H and x + ts <= W)]
    keep = tiler.select_empty_tiles(emp, 0.37, 7)
    assert [emp[i] for i in ot.select_empty(len(emp), 0.37, 7)] == keep


def test_tiler_known_answers():
    """one tile at the origin, tile size 100: a box fully inside; a box whose corner midpoint is inside but 95 % of which lies outside
    (dropped); a box half outside (kept, clipped to the tile); a box whose midpoint sits exactly on the right border (half-open: not ours)"""
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops
    lab = np.array([[10, 10, 30, 10, 30, 30, 10, 30],
                    [95, 40, 195, 40, 195, 60, 97, 60],        # midpoint of corners 1 and 4 = (96, 50): inside; coverage 5/100 < 0.1
                    [60, 20, 140, 20, 140, 40, 60, 40],        # midpoint (60, 30): inside; coverage 0.5; clipped to x <= 100
                    [100, 70, 120, 70, 120, 90, 100, 90]], dtype=np.float64)   # midpoint (100, 80): on the border -> next tile
    rects = torch.tensor([[0, 0, 100, 100]], dtype=torch.int32, device="cuda")
    mask, out = ops.tile_labels(torch.tensor(lab).cuda(), rects, 0.1)
    assert mask.cpu().numpy().tolist() == [[1, 0, 1, 0]]
    o = out.cpu().numpy()[0]
    assert np.array_equal(o[0], lab[0] / 100)
    assert np.array_equal(o[2], np.array([60, 20, 100, 20, 100, 40, 60, 40]) / 100)
    assert not o[1].any() and not o[3].any()
