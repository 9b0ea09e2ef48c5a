"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's 4-channel input builder `build_multich` (Detect_OBB.py:87-133,
identical to Train_OBB.py:615-653): RGB + a distance-transform edge channel, per crop.

The reference calls OpenCV (opencv-python==4.11.0.86, absent offline) for every image operation, so each cv2 call is restated here
from its documented algorithm; **parity with cv2 itself is unpinned** (no fixture of the reference covers it):

    cv2.cvtColor(BGR2GRAY)            8-bit fixed point: (B*1868 + G*9617 + R*4899 + 2^13) >> 14
    cv2.GaussianBlur(gray,(0,0),s)    ksize = round(6 s + 1) | 1 (8-bit rule) -> 5 / 9 / 15; separable, BORDER_REFLECT_101; 8-bit fixed-point
                                      path: kernel in 1/256 units (sum forced to 256 at the centre tap), horizontal pass kept in 8.8 fixed
                                      point, vertical pass rounded to uint8
    cv2.Scharr(.., CV_32F, 1,0 / 0,1) [-3 0 3; -10 0 10; -3 0 3] and its transpose, BORDER_REFLECT_101, exact in float32
    cv2.magnitude                     sqrt(dx*dx + dy*dy) in float32
    cv2.getStructuringElement(MORPH_ELLIPSE,(3,3)) + morphologyEx(OPEN)   3x3 cross; erode then dilate, pixels outside the image ignored
    cv2.distanceTransform(DIST_L2, 3) two-pass 3x3 chamfer in 16.16 fixed point (a = 0.955, b = 1.3693), result * 2^-16 as float32
    cv2.normalize(NORM_MINMAX, 0..1)  float32: x * scale + shift with scale = 1 / (max - min), shift = -min * scale

numpy steps: np.percentile (float64 linear interpolation) is executed by numpy itself.  The arithmetic BEHIND the percentiles follows the
promotion rules of the reference's pinned numpy==1.26.4 (requirements.txt:1), not those of the numpy that runs this file: before NEP 50
(numpy < 2) a float32 ARRAY combined with a numpy float64 SCALAR stays float32 (value-based casting), so `acc >= hi`, `dist - lo`,
`/ max(1e-6, hi - lo)`, `np.exp(-dist / tau)`, the 0.7 / 0.3 blend and `soft * 255` all run in float32 there; numpy >= 2 would promote every
one of them to float64.  The casts are therefore explicit below.  np.exp on float32 is numpy's SIMD routine (a few ulp, build-dependent):
restated as float32(exp64(x)) with exp64 a fixed double-precision polynomial of IEEE basic operations -- within 0.5000001 ulp of the true
value, and bit-reproducible on the device (csrc/dtedge.hip evaluates the same operations); parity with numpy's own float32 exp is
unpinned (it may differ by one ulp, i.e. one grey level on ~1e-5 of the pixels)."""
import numpy as np

MS_SIGMAS = (0, 0.6, 1.2, 2.4)   # Detect_OBB.py:29
DT_P_HI, DT_P_LO = 90, 65        # :31
DT_MORPH_OPEN = 1                # :32
HV_DIST, DIAG_DIST, INIT_DIST = 62587, 89738, (2 ** 31 - 1) >> 2   # round(0.955 * 2^16), round(1.3693 * 2^16), INT_MAX >> 2


def gray_u8(bgr):
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def gauss_kernel_q8(sigma):
    n = int(round(sigma * 6 + 1)) | 1
    c = (n - 1) / 2.0
    g = np.exp(-((np.arange(n) - c) ** 2) / (2.0 * sigma * sigma))
    g /= g.sum()
    q = np.rint(g * 256.0).astype(np.int64)
    q[n // 2] += 256 - q.sum()
    return q


def _reflect101(idx, n):
    """cv2 borderInterpolate(BORDER_REFLECT_101): folds until the index is inside, i.e. an even extension of period 2 (n - 1)"""
    if n == 1:
        return np.zeros_like(idx)
    idx = np.mod(idx, 2 * (n - 1))
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def gaussian_blur_u8(gray, sigma):
    q = gauss_kernel_q8(sigma)
    r = len(q) // 2
    h, w = gray.shape
    xs = _reflect101(np.arange(-r, w + r), w)
    row = np.zeros((h, w), np.int64)
    g = gray.astype(np.int64)
    for i, qi in enumerate(q):
        row += qi * g[:, xs[i:i + w]]            # 8.8 fixed point (<= 255 * 256)
    ys = _reflect101(np.arange(-r, h + r), h)
    col = np.zeros((h, w), np.int64)
    for i, qi in enumerate(q):
        col += qi * row[ys[i:i + h], :]
    return ((col + (1 << 15)) >> 16).astype(np.uint8)


def scharr_mag(img_u8):
    h, w = img_u8.shape
    p = np.pad(img_u8.astype(np.float32), 1, mode="reflect")  # numpy "reflect" == BORDER_REFLECT_101
    dx = (3 * (p[0:h, 2:] - p[0:h, 0:w]) + 10 * (p[1:h + 1, 2:] - p[1:h + 1, 0:w]) + 3 * (p[2:, 2:] - p[2:, 0:w])).astype(np.float32)
    dy = (3 * (p[2:, 0:w] - p[0:h, 0:w]) + 10 * (p[2:, 1:w + 1] - p[0:h, 1:w + 1]) + 3 * (p[2:, 2:] - p[0:h, 2:])).astype(np.float32)
    return np.sqrt(dx * dx + dy * dy, dtype=np.float32)


def morph_open_cross(edges):
    def reduce(img, fn, fill):
        p = np.pad(img, 1, mode="constant", constant_values=fill)
        h, w = img.shape
        return fn(fn(fn(p[1:h + 1, 1:w + 1], p[0:h, 1:w + 1]), fn(p[2:, 1:w + 1], p[1:h + 1, 0:w])), p[1:h + 1, 2:])
    return reduce(reduce(edges, np.minimum, 255), np.maximum, 0)


def distance_transform_3x3(non_edge):
    """distance (float32) of every pixel to the nearest zero pixel of `non_edge`"""
    h, w = non_edge.shape
    t = np.full((h + 2, w + 2), INIT_DIST, np.int64)
    for i in range(1, h + 1):           # forward
        up = t[i - 1]
        c = np.minimum(np.minimum(up[0:w] + DIAG_DIST, up[1:w + 1] + HV_DIST), up[2:w + 2] + DIAG_DIST)
        c = np.where(non_edge[i - 1] == 0, 0, c)
        # d[x] = min(c[x], d[x-1] + HV): prefix minimum of c[k] - k*HV
        x = np.arange(w, dtype=np.int64)
        left0 = INIT_DIST + HV_DIST    # border column
        d = np.minimum(np.minimum.accumulate(c - x * HV_DIST) + x * HV_DIST, left0 + x * HV_DIST)
        t[i, 1:w + 1] = d
    for i in range(h, 0, -1):           # backward
        dn = t[i + 1]
        cur = t[i, 1:w + 1]
        c = np.minimum(cur, np.minimum(np.minimum(dn[2:w + 2] + DIAG_DIST, dn[1:w + 1] + HV_DIST), dn[0:w] + DIAG_DIST))
        x = np.arange(w, dtype=np.int64)[::-1]   # distance from the right end
        cr = c[::-1]
        d = np.minimum(np.minimum.accumulate(cr - np.arange(w) * HV_DIST) + np.arange(w) * HV_DIST, INIT_DIST + HV_DIST + np.arange(w) * HV_DIST)
        t[i, 1:w + 1] = d[::-1]
    return (t[1:h + 1, 1:w + 1].astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)


def normalize_minmax01(acc):
    mn, mx = float(acc.min()), float(acc.max())
    scale = (1.0 / (mx - mn)) if mx > mn else 0.0
    shift = -mn * scale
    return (acc * np.float32(scale) + np.float32(shift)).astype(np.float32)


_EXP_C = [1.0 / float(__import__("math").factorial(k)) for k in range(15)]  # 1 / k!: correctly rounded quotients of exact integers


def exp32(x):
    """float32(exp(x)) for float32 x in [-0.5, 0]: Horner evaluation of the degree-14 Taylor polynomial in float64 (remainder < 1e-18),
    one multiplication and one addition per step, each rounded -- the device runs the identical sequence (no fused multiply-add)"""
    x = np.asarray(x, np.float32).astype(np.float64)
    assert x.size == 0 or (float(x.min()) >= -0.5 and float(x.max()) <= 0.0)
    p = np.full_like(x, _EXP_C[14])
    for k in range(13, -1, -1):
        p = p * x
        p = p + _EXP_C[k]
    return p.astype(np.float32)


def build_multich(bgr, out_channels=4):
    """Detect_OBB.py:87-133"""
    assert out_channels in (3, 4)
    if out_channels == 3:
        return np.ascontiguousarray(bgr)
    rgb_raw = bgr[..., ::-1]
    gray = gray_u8(bgr)
    acc = None
    for s in MS_SIGMAS:
        blur = gaussian_blur_u8(gray, s) if s > 0 else gray
        mag = scharr_mag(blur)
        acc = mag if acc is None else np.maximum(acc, mag)
    lo, hi = np.percentile(acc, [DT_P_LO, DT_P_HI])
    edges = (acc >= np.float32(hi)).astype(np.uint8) * 255  # numpy 1.26: the float64 scalar is cast to the array's float32 before the compare
    if DT_MORPH_OPEN > 0:
        for _ in range(DT_MORPH_OPEN):
            edges = morph_open_cross(edges)
    non_edge = np.where(edges > 0, 0, 255).astype(np.uint8)
    dist = distance_transform_3x3(non_edge).astype(np.float32)
    lo, hi = np.percentile(dist, [1, 99])
    den = max(1e-6, (hi - lo))                                          # float64 scalars so far
    dist = np.clip((dist - np.float32(lo)) / np.float32(den), np.float32(0), np.float32(1))  # float32 from here on (numpy 1.26 value-based casting)
    tau = np.float32(3.0)
    soft = exp32(-dist / tau)
    acc8_nrm = normalize_minmax01(acc)
    soft = np.float32(0.7) * soft + np.float32(0.3) * acc8_nrm
    soft = np.clip(soft, np.float32(0), np.float32(1))
    assert soft.dtype == np.float32
    dt_edge = (soft * np.float32(255)).astype(np.uint8)
    return np.ascontiguousarray(np.dstack([rgb_raw, dt_edge]).astype(np.uint8))
