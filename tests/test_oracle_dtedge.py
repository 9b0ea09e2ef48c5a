"""Known answers for the numpy restatement of build_multich's OpenCV steps (oracle/dtedge.py; cv2 is absent: parity unpinned)."""
import numpy as np

from oracle import dtedge as od


def test_grey_and_kernels():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)  # BGR
    assert od.gray_u8(px).tolist() == [[255, 0, 29, 150, 76]]           # 0.114 B + 0.587 G + 0.299 R
    for s, n in ((0.6, 5), (1.2, 9), (2.4, 15)):
        q = od.gauss_kernel_q8(s)
        assert len(q) == n and q.sum() == 256 and np.array_equal(q, q[::-1]) and q.min() >= 0
    flat = np.full((20, 30), 77, np.uint8)
    assert np.array_equal(od.gaussian_blur_u8(flat, 2.4), flat)         # a normalised kernel keeps a constant image
    assert float(od.scharr_mag(flat).max()) == 0.0


def test_reflect101_folds_repeatedly():
    # cv2 borderInterpolate(BORDER_REFLECT_101): gfedcb|abcdefgh|gfedcba, repeated for offsets beyond one image length
    assert od._reflect101(np.arange(-7, 10), 3).tolist() == [1, 2, 1, 0, 1, 2, 1, 0, 1, 2, 1, 0, 1, 2, 1, 0, 1]
    assert od._reflect101(np.arange(-2, 7), 5).tolist() == [2, 1, 0, 1, 2, 3, 4, 3, 2]
    assert od._reflect101(np.arange(-3, 4), 1).tolist() == [0] * 7
    tiny = np.arange(15, dtype=np.uint8).reshape(3, 5) * 10
    ref = np.pad(tiny.astype(np.int64), 7, mode="reflect")                # numpy folds repeatedly as well
    q = od.gauss_kernel_q8(2.4)
    row = sum(int(q[i]) * ref[7:10, i:i + 5] for i in range(15))
    assert np.array_equal(od.gaussian_blur_u8(tiny, 2.4)[1], ((sum(int(q[i]) * sum(int(q[j]) * ref[1 + i, j:j + 5] for j in range(15)) for i in range(15)) + (1 << 15)) >> 16).astype(np.uint8))
    assert row.shape == (3, 5)


def test_scharr_open_distance():
    step = np.zeros((9, 9), np.uint8)
    step[:, 5:] = 10
    m = od.scharr_mag(step)
    assert m[4, 4] == 160.0 and m[4, 5] == 160.0 and m[4, 2] == 0.0     # (3 + 10 + 3) * 10 across the step
    e = np.zeros((7, 7), np.uint8)
    e[3, 3] = 255                                                       # an isolated pixel does not survive the opening
    assert od.morph_open_cross(e).max() == 0
    e[2:5, 2:5] = 255
    o = od.morph_open_cross(e)
    assert o[3, 3] == 255 and o[2, 3] == 255 and o[2, 2] == 0           # the cross-shaped core of a 3x3 block survives
    ne = np.full((5, 9), 255, np.uint8)
    ne[2, 0] = 0
    d = od.distance_transform_3x3(ne)
    assert d[2, 0] == 0 and abs(d[2, 1] - 0.955) < 1e-4 and abs(d[3, 1] - 1.3693) < 1e-4 and abs(d[2, 8] - 8 * 0.955) < 1e-3
    assert abs(d[4, 3] - (2 * 1.3693 + 0.955)) < 1e-3


def test_build_multich_shape_and_ranges():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    out = od.build_multich(img, 4)
    assert out.shape == (40, 56, 4) and out.dtype == np.uint8 and np.array_equal(out[..., :3], img[..., ::-1])
    assert np.array_equal(od.build_multich(img, 3), img)
    const = od.build_multich(np.full((16, 16, 3), 9, np.uint8), 4)
    assert const[..., 3].min() == const[..., 3].max()                   # no structure -> a flat channel
