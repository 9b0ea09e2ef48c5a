"""f4 (second part): the training-set tiler of Train_OBB.py -- `enumerate_and_save_nonempty_tiles` :44-146 and
`save_selected_empty_tiles` :167-226 -- with the per-tile work on the GPU: every full tile of an image is cropped in one launch
(obb_gather_tiles) and every (tile, label) pair is decided and transformed in one launch (obb_tile_labels).  The file I/O around it
(jpg / txt writing) stays with the caller; what this module returns is exactly what the reference writes: per positive tile its id,
origin, crop and label rows (class + 8 normalised coordinates), and the metadata of the empty tiles."""
import numpy as np
import torch

from . import ops

OBJECT_BOUNDARY_THRESHOLD = 0.1  # Train_OBB.py:33


def enumerate_full_tiles(H, W, tile_size=128, overlap=50):
    """(tile_id, x, y) of every tile the reference visits (:80-85): row-major over range(0, H, stride) x range(0, W, stride), tiles that
    would cross the image border are skipped and do NOT consume an id."""
    stride = tile_size - overlap
    assert stride > 0, "overlap must be < tile_size"
    out = []
    tid = 0
    for y in range(0, H, stride):
        for x in range(0, W, stride):
            if y + tile_size > H or x + tile_size > W:
                continue
            out.append((tid, x, y))
            tid += 1
    return out


def tile_image(image, labels, tile_size=128, overlap=50, boundary_threshold=OBJECT_BOUNDARY_THRESHOLD, device=None):
    """image: uint8 [H, W, C] (numpy or tensor); labels: array [n, 9] = class + 8 pixel coordinates (read_labels_or_empty's frame).
    -> (positives, empties): positives = list of dicts {tile_id, x, y, crop (uint8 tensor [ts, ts, C] on the device), labels (float64
    numpy [k, 9], class + normalised corners, in label-file order)}; empties = list of {tile_id, x, y, tile_size} (:113-119)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    img = torch.as_tensor(image).to(dev).contiguous()
    H, W, C = img.shape
    lab = np.asarray(labels, dtype=np.float64).reshape(-1, 9)
    tiles = enumerate_full_tiles(H, W, tile_size, overlap)
    if not tiles:
        return [], []
    rects = torch.tensor([[x, y, x + tile_size, y + tile_size] for _, x, y in tiles], dtype=torch.int32, device=dev)
    crops = ops.gather_tiles(img, rects, tile_size)
    mask, out = ops.tile_labels(torch.tensor(lab[:, 1:], dtype=torch.float64, device=dev), rects, boundary_threshold)
    mask_h, out_h = mask.cpu().numpy().astype(bool), out.cpu().numpy()
    positives, empties = [], []
    for k, (tid, x, y) in enumerate(tiles):
        sel = np.nonzero(mask_h[k])[0]
        if len(sel):
            rows = np.concatenate([lab[sel, :1], out_h[k, sel]], axis=1)
            positives.append({"tile_id": tid, "x": x, "y": y, "crop": crops[k], "labels": rows})
        else:
            empties.append({"tile_id": int(tid), "x": int(x), "y": int(y), "tile_size": int(tile_size)})
    return positives, empties


def select_empty_tiles(empties, keep_fraction, rng_seed=42):
    """save_selected_empty_tiles :167-190: round(keep_fraction * len) of the enumerated empty tiles, chosen by a seeded shuffle."""
    assert 0.0 <= keep_fraction <= 1.0
    if len(empties) == 0:
        return []
    k = int(round(keep_fraction * len(empties)))
    rng = np.random.RandomState(rng_seed)
    idx = np.arange(len(empties))
    rng.shuffle(idx)
    return [empties[i] for i in idx[:k]]
