"""Row f3: the 4-channel input builder (RGB + distance-transform edge channel, Detect_OBB.py:87-133) on the device vs the numpy
restatement in oracle/dtedge.py: byte for byte.  Integer stages are exact by construction; the float tail follows the float32 promotion of
the reference's pinned numpy 1.26.4, with exp() as the same sequence of IEEE double operations on both sides (see oracle/dtedge.py)."""
import numpy as np
import pytest
import torch

from oracle import dtedge as od

pytestmark = pytest.mark.gpu


def _images():
    rng = np.random.default_rng(7)
    out = []
    noise = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    out.append(noise)
    shapes = np.full((175, 263, 3), 230, np.uint8)                      # map-like: light background, dark strokes and blobs
    shapes[40:44, 20:240] = 20
    shapes[60:150, 100:104] = (10, 40, 200)
    yy, xx = np.mgrid[0:175, 0:263]
    shapes[(yy - 110) ** 2 + (xx - 200) ** 2 < 400] = (0, 120, 60)
    shapes = np.clip(shapes.astype(np.int32) + rng.integers(-6, 7, shapes.shape), 0, 255).astype(np.uint8)
    out.append(shapes)
    grad = np.stack([np.tile(np.linspace(0, 255, 416), (416, 1))] * 3, -1).astype(np.uint8)
    grad[200:216, :] = 0
    out.append(grad)                                                     # full 416 x 416 tile
    out.append(np.full((64, 96, 3), 127, np.uint8))                      # constant: no edges at all
    out.append(rng.integers(0, 256, (5, 9, 3), dtype=np.uint8))          # ragged border crop smaller than the blur radius: repeated reflection
    yy, xx = np.mgrid[0:128, 0:128]
    wave = (127 + 90 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.uint8)
    out.append(np.stack([wave, wave.T, 255 - wave], -1))                 # the 128-px scale of the dual-scale run (2 pixels per lane)
    wide = rng.integers(0, 256, (40, 600, 3), dtype=np.uint8)
    wide[:, 300:] = (wide[:, 300:] // 32) * 32                            # many ties in the order statistics; 16 pixels per lane
    out.append(wide)
    return out


@pytest.mark.parametrize("idx", range(7))
def test_build_multich_matches_numpy_restatement(idx):
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops
    img = _images()[idx]
    exp = od.build_multich(img, 4)
    batch = np.stack([img, img[::-1].copy(), img[:, ::-1].copy()])       # three crops of the same shape in one call
    got = ops.build_multich(torch.as_tensor(batch).cuda()).cpu().numpy()
    assert got.shape == (3,) + img.shape[:2] + (4,) and got.dtype == np.uint8
    assert np.array_equal(got[0, ..., :3], img[..., ::-1])               # RGB
    d = np.abs(got[0, ..., 3].astype(np.int32) - exp[..., 3].astype(np.int32))
    print(img.shape, "dt channel: max diff", d.max(), "pixels differing", int((d > 0).sum()), "of", d.size, "range", exp[..., 3].min(), exp[..., 3].max())
    assert np.array_equal(got[0], exp)
    for k, im in ((1, img[::-1].copy()), (2, img[:, ::-1].copy())):
        assert np.array_equal(got[k], od.build_multich(im, 4)), k


def test_build_multich_ragged_shapes():
    """Crops of arbitrary size (ragged border tiles): every pixel-tile / lane-chunk remainder of the six kernels, one batch per shape."""
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops
    rng = np.random.default_rng(11)
    shapes = [(2, 2), (3, 17), (17, 3), (33, 65), (64, 129), (97, 257), (31, 513), (130, 1024), (200, 70)]
    for h, w in shapes:
        base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        smooth = np.clip(np.cumsum(rng.integers(-3, 4, (h, w, 3)), axis=1) + 128, 0, 255).astype(np.uint8)  # few distinct gradients: ties
        batch = np.stack([base, smooth])
        got = ops.build_multich(torch.as_tensor(batch).cuda()).cpu().numpy()
        for k in range(2):
            exp = od.build_multich(batch[k], 4)
            d = np.abs(got[k, ..., 3].astype(np.int32) - exp[..., 3].astype(np.int32))
            assert np.array_equal(got[k], exp), ((h, w), k, int(d.max()), int((d > 0).sum()))


def test_build_multich_rejects_unsupported_sizes():
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops
    with pytest.raises(RuntimeError):
        ops.build_multich(torch.zeros((1, 8, 1025, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.build_multich(torch.zeros((1, 1, 8, 3), dtype=torch.uint8, device="cuda"))
