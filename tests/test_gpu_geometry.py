"""GPU parity: HIP geometry kernels (through the C-ABI) vs the CPU oracle and the reference-generated goldens.
Bit-exact for keep masks / orders / indices and for fp64 IoU values (same op order, -ffp-contract=off both sides)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import CLASS_IDS, GOLDEN, load_xlsx_csv
from oracle import geom as og
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops as o
    return o


def dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_iou_pairs_bit_exact(ops):
    boxes, cls, conf, _ = synth.make_dets(123, 20000, extent=900.0)
    rng = np.random.default_rng(5)
    a = boxes
    b = boxes[rng.permutation(len(boxes))]
    b[:5000] = a[:5000] + rng.normal(0, 3.0, (5000, 1))  # guaranteed heavy overlaps
    b[5000:5010] = a[5000:5010]  # identical
    a[6000] = [0, 0, 1, 1, 1, 0, 0, 1]  # bow-tie -> 0
    a[6001] = np.nan
    exp = og.poly_iou_pairs(a, b)
    got = ops.poly_iou_pairs(dev(a, torch.float64), dev(b, torch.float64)).cpu().numpy()
    assert (exp > 0.3).sum() > 1000
    assert np.array_equal(got, exp)
    assert got[6000] == 0.0 and got[6001] == 0.0 and np.all(got[5000:5010] == 1.0)


def test_iou_pairs_empty_and_ragged(ops):
    e = torch.zeros((0, 8), dtype=torch.float64, device="cuda")
    assert ops.poly_iou_pairs(e, e).numel() == 0
    for m in (1, 63, 64, 65, 255, 257):
        boxes, _, _, _ = synth.make_dets(m, 2 * m, extent=120.0)
        a, b = boxes[:m], boxes[m:]
        assert np.array_equal(ops.poly_iou_pairs(dev(a, torch.float64), dev(b, torch.float64)).cpu().numpy(), og.poly_iou_pairs(a, b))


def test_iou_matrix(ops):
    b1, c1, _, _ = synth.make_dets(1, 150, extent=300.0)
    b2, c2, _, _ = synth.make_dets(2, 77, extent=300.0)
    got = ops.poly_iou_matrix(dev(b1, torch.float64), dev(b2, torch.float64), dev(c1, torch.int32), dev(c2, torch.int32)).cpu().numpy()
    exp = np.zeros((150, 77))
    for i in range(150):
        for j in range(77):
            if c1[i] == c2[j]:
                exp[i, j] = og.compute_polygon_iou(b1[i], b2[j])
    assert np.array_equal(got, exp)
    got2 = ops.poly_iou_matrix(dev(b1, torch.float64), dev(b2, torch.float64)).cpu().numpy()
    assert got2[3, 5] == og.compute_polygon_iou(b1[3], b2[5])


def test_sort_desc_stable(ops):
    rng = np.random.default_rng(0)
    for n in (1, 2, 100, 1024, 1025, 5000, 8192, 40000, 150000):  # >= 8192: the bucketed O(n) form must give the same order
        k = rng.uniform(0, 1, n).astype(np.float32).astype(np.float64)
        k[rng.integers(0, n, n // 4 + 1)] = k[0]  # ties
        got = ops.sort_desc_stable(dev(k, torch.float64)).cpu().numpy()
        assert np.array_equal(got, og.sort_desc_stable(k))
        assert np.array_equal(got, np.argsort(-k, kind="stable"))
    # confidences as the path produces them ([0.25, 1], float32 values), all-equal keys, negative / signed-zero / infinite keys
    for k in (rng.uniform(0.25, 1, 30000).astype(np.float32).astype(np.float64), np.full(9000, 0.5), np.concatenate([rng.normal(0, 1, 9000), [0.0, -0.0, np.inf, -np.inf, 0.0]])):
        got = ops.sort_desc_stable(dev(k, torch.float64)).cpu().numpy()
        assert np.array_equal(got, og.sort_desc_stable(k))


def test_merge_matches_reference_goldens(ops, ref_vectors):
    rv = ref_vectors
    for ci in rv["merge_cases"]:
        boxes, cls, conf = rv[f"merge{ci}_boxes"].reshape(-1, 8), rv[f"merge{ci}_cls"], rv[f"merge{ci}_conf"]
        if len(cls) == 0:
            continue
        order, keep, nk = ops.merge_detections(dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64), float(rv[f"merge{ci}_thr"]))
        order, keep = order.cpu().numpy(), keep.cpu().numpy().astype(bool)
        assert np.array_equal(order, rv[f"merge{ci}_sorted"]), f"case {ci}"
        assert np.array_equal(order[keep], rv[f"merge{ci}_kept"]), f"case {ci}"
        assert int(nk.item()) == len(rv[f"merge{ci}_kept"])


@pytest.mark.parametrize("n,extent", [(513, 400.0), (4096, 1500.0), (16384, 4096.0)])
def test_merge_dense_path_vs_oracle(ops, n, extent):
    boxes, cls, conf, _ = synth.make_dets(1000 + n, n, extent=extent)
    eo, ek = og.merge_arrays(boxes, cls, conf, 0.4)
    order, keep, nk = ops.merge_detections(dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64), 0.4)
    assert np.array_equal(order.cpu().numpy(), eo)
    assert np.array_equal(keep.cpu().numpy(), ek)
    assert int(nk.item()) == int(ek.sum())
    # low-level API: mask + reduce on pre-sorted input gives the same keep flags
    sb, sc = boxes[eo], cls[eo]
    mask = ops.nms_mask(dev(sb, torch.float64), dev(sc, torch.int32), 0.4)
    k2, nk2 = ops.nms_reduce(mask, n)
    assert np.array_equal(k2.cpu().numpy(), ek)


@pytest.mark.parametrize("case", ["pile", "thr0", "thr0_grid", "pile_grid"])
def test_merge_matrix_free_dense_scan(ops, case):
    """the device-predicated fall-back of the sparse pair list (k_nms_lazy): (a) a pair list that overflows its 64 n + 4096 entries -- piles of
    near-identical same-class boxes, (b) thr <= 0 where every same-class pair suppresses (Detect_OBB.py:193: IoU >= thr always holds) -- must
    give the oracle's keep flags, below and above the grid pair search's threshold of 8192 rows; no host read decides between the paths."""
    rng = np.random.default_rng(21)
    if case in ("pile", "pile_grid"):
        n = 3000 if case == "pile" else 9000
        base, _, _, _ = synth.make_dets(5, 6, extent=300.0)
        boxes = base[rng.integers(0, 6, n)] + rng.normal(0, 0.05, (n, 8))  # six piles of ~n/6 near-identical boxes each: ~n^2/12 pairs
        cls = np.zeros(n, np.int32)
        thr = 0.4
    else:
        n = 2500 if case == "thr0" else 8500
        boxes, cls, _, _ = synth.make_dets(6, n, extent=3000.0)
        thr = 0.0
    conf = rng.uniform(0.25, 1, n).astype(np.float32).astype(np.float64)
    eo, ek = og.merge_arrays(boxes, cls, conf, thr)
    order, keep, nk = ops.merge_detections(dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64), thr)
    assert np.array_equal(order.cpu().numpy(), eo)
    assert np.array_equal(keep.cpu().numpy(), ek)
    assert int(nk.item()) == int(ek.sum())
    if thr == 0.0:
        assert int(ek.sum()) == len(np.unique(cls))  # one survivor per class


def test_merge_idempotent_at_scale(ops):
    """size-independent property at a size the oracle is too slow for: merging the survivors again changes nothing."""
    n = 65536
    boxes, cls, conf, _ = synth.make_dets(7, n, extent=8192.0)
    B, Cc, S = dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64)
    order, keep, nk = ops.merge_detections(B, Cc, S, 0.4)
    sel = order[keep.bool()].long()
    assert len(sel) == int(nk.item()) and 0 < len(sel) < n
    o2, k2, nk2 = ops.merge_detections(B[sel].contiguous(), Cc[sel].contiguous(), S[sel].contiguous(), 0.4)
    assert bool(k2.all()) and int(nk2.item()) == len(sel)
    assert np.array_equal(o2.cpu().numpy(), np.arange(len(sel)))  # already conf-sorted
    srt = S[order.long()].cpu().numpy()
    assert np.all(np.diff(srt) <= 0)
    # the grid-binned pair search (n >= 8192) against the all-pairs bit matrix on the same sorted boxes: identical keep flags; also
    # with one box as large as the whole map, and with 5 % of the rows 20x enlarged (rows above the cell size stay out of the grid and
    # are tested against every row)
    for huge in (0, 1, 2):
        bb = boxes.copy()
        if huge == 1:
            bb[17] = [0, 0, 8192, 0, 8192, 8192, 0, 8192]
        if huge == 2:
            pick = np.random.default_rng(9).permutation(n)[: n // 20]
            ctr = bb[pick].reshape(-1, 4, 2).mean(1, keepdims=True)
            bb[pick] = ((bb[pick].reshape(-1, 4, 2) - ctr) * 20.0 + ctr).reshape(-1, 8)
        Bh = dev(bb, torch.float64)
        order, keep, nk = ops.merge_detections(Bh, Cc, S, 0.4)
        sb, sc = Bh[order.long()].contiguous(), Cc[order.long()].contiguous()
        k2, _ = ops.nms_reduce(ops.nms_mask(sb, sc, 0.4), n)
        assert torch.equal(k2, keep), huge


def test_sort_with_heavy_ties(ops):
    """confidences that went through 16-bit arithmetic: a few thousand distinct values, so single key buckets hold thousands of equal keys
    (ranked by a workgroup per bucket); order must still be the stable descending order"""
    rng = np.random.default_rng(11)
    for n, kinds in ((20000, 40), (70000, 3), (9000, 1)):
        s = rng.choice(rng.uniform(0.25, 1.0, kinds).astype(np.float16).astype(np.float64), n)
        mix = rng.integers(0, n, n // 3)
        s[mix] = rng.uniform(0.25, 1.0, len(mix)).astype(np.float32)  # a third of the rows with (nearly) unique keys in between
        order = ops.sort_desc_stable(dev(s, torch.float64)).cpu().numpy()
        exp = np.argsort(-s, kind="stable")
        assert np.array_equal(order, exp), (n, kinds)


def test_merge_segments_vs_oracle(ops):
    rng = np.random.default_rng(3)
    sizes = [0, 1, 5, 300, 0, 64, 65, 512, 17]
    bs, cs, ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        b, c, s, _ = synth.make_dets(50 + k, n, extent=250.0) if n else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        bs.append(b); cs.append(c); ss.append(s); off.append(off[-1] + n)
    B, Cc, S = np.concatenate(bs), np.concatenate(cs), np.concatenate(ss)
    order, keep = ops.merge_segments(dev(B, torch.float64), dev(Cc, torch.int32), dev(S, torch.float64), dev(np.array(off), torch.int32), 0.4)
    order, keep = order.cpu().numpy(), keep.cpu().numpy()
    for k, n in enumerate(sizes):
        if n == 0:
            continue
        eo, ek = og.merge_arrays(bs[k], cs[k], ss[k], 0.4)
        assert np.array_equal(order[off[k]:off[k + 1]], eo + off[k]), k
        assert np.array_equal(keep[off[k]:off[k + 1]], ek), k


def test_merge_segments_many_long_segments_walk_the_device_list(ops):
    """More long segments (65 .. 512 rows) than the chip holds workgroups of the 512-row kernel: the resident workgroups walk the device-side
    list that the wave kernel builds, several segments each (LDS reused between them); mixed with short and empty segments.  One segment is
    a single-class pile of near-duplicates (every pair a candidate: the pair list overflows into the in-place clips, and the IoU upper bound
    sits at the threshold for many pairs).  Against the C oracle, bit for bit."""
    rng = np.random.default_rng(11)
    sizes = [int(v) for v in rng.choice([0, 3, 40, 64, 65, 90, 130, 200, 300, 512], size=640, p=[.05, .05, .1, .05, .15, .2, .2, .1, .05, .05])]
    sizes[7] = 180
    bs, cs, ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        b, c, s, _ = synth.make_dets(2000 + k, n, extent=120.0 + 3.0 * n) if n else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        if k == 7:  # the pile: one class, boxes jittered by a pixel or two
            b = np.repeat(b[:1], n, 0) + rng.normal(0, 1.5, (n, 8))
            c = np.zeros(n, np.int32)
        bs.append(b); cs.append(c); ss.append(s); off.append(off[-1] + n)
    assert sum(n > 64 for n in sizes) > 300
    B, Cc, S = np.concatenate(bs), np.concatenate(cs), np.concatenate(ss)
    order, keep = ops.merge_segments(dev(B, torch.float64), dev(Cc, torch.int32), dev(S, torch.float64), dev(np.array(off), torch.int32), 0.4)
    order, keep = order.cpu().numpy(), keep.cpu().numpy()
    for k, n in enumerate(sizes):
        if n == 0:
            continue
        eo, ek = og.merge_arrays(bs[k], cs[k], ss[k], 0.4)
        assert np.array_equal(order[off[k]:off[k + 1]], eo + off[k]), k
        assert np.array_equal(keep[off[k]:off[k + 1]], ek), (k, n)


def test_merge_segments_longer_than_the_lds_kernel(ops):
    """a tile with more than 512 detections (max_det > 512, or a foreign model without a cap): such segments used to be dropped silently;
    they now go through the dense path, the short ones around them through the LDS kernel, all in one call"""
    sizes = [40, 513, 0, 1500, 512, 7]
    bs, cs, ss, off = [], [], [], [0]
    for k, n in enumerate(sizes):
        b, c, s, _ = synth.make_dets(90 + k, n, extent=600.0) if n else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        bs.append(b); cs.append(c); ss.append(s); off.append(off[-1] + n)
    B, Cc, S = np.concatenate(bs), np.concatenate(cs), np.concatenate(ss)
    order, keep = ops.merge_segments(dev(B, torch.float64), dev(Cc, torch.int32), dev(S, torch.float64), dev(np.array(off), torch.int32), 0.4)
    order, keep = order.cpu().numpy(), keep.cpu().numpy()
    for k, n in enumerate(sizes):
        if n == 0:
            continue
        eo, ek = og.merge_arrays(bs[k], cs[k], ss[k], 0.4)
        assert np.array_equal(order[off[k]:off[k + 1]], eo + off[k]), k
        assert np.array_equal(keep[off[k]:off[k + 1]], ek), k
        assert 0 < ek.sum() < n or n < 3


def test_consensus_matches_reference_goldens(ops, ref_vectors):
    rv = ref_vectors
    for ci in rv["cons_cases"]:
        b = np.concatenate([rv[f"cons{ci}_b1"].reshape(-1, 8), rv[f"cons{ci}_b2"].reshape(-1, 8)])
        c = np.concatenate([rv[f"cons{ci}_c1"], rv[f"cons{ci}_c2"]]).astype(np.int32)
        s = np.concatenate([rv[f"cons{ci}_s1"], rv[f"cons{ci}_s2"]])
        n1 = len(rv[f"cons{ci}_c1"])
        if len(c) == 0:
            continue
        idx, nout = ops.consensus(dev(b, torch.float64), dev(c, torch.int32), dev(s, torch.float64), [0, n1, len(c)])
        idx = idx.cpu().numpy()[:int(nout.item())]
        tags = np.where(idx < n1, idx, idx - n1 + 100000)
        assert np.array_equal(tags, rv[f"cons{ci}_kept"]), f"case {ci}"
    ci = rv["cons_cases"][-1]
    n2 = len(rv[f"cons{ci}_c2"])
    idx, nout = ops.consensus(dev(rv[f"cons{ci}_b2"], torch.float64), dev(rv[f"cons{ci}_c2"], torch.int32), dev(rv[f"cons{ci}_s2"], torch.float64), [0, n2])
    assert np.array_equal(idx.cpu().numpy()[:int(nout.item())] + 100000, rv["cons_single_kept"])


def test_detect_symbols_body_matches_reference(ops):
    """tile grid + per-detection body + per-tile merge == the reference's detect_symbols on the same stub model."""
    cases = json.load(open(os.path.join(GOLDEN, "detect_symbols_cases.json")))
    for c in cases:
        rects = ops.tile_grid(c["H"], c["W"], c["tile"], c["overlap"])
        assert [list(map(int, (x, y, y2 - y, x2 - x))) for (x, y, x2, y2) in rects] == c["tiles"]
        pts, cls, conf, tid = [], [], [], []
        for t, (x, y, x2, y2) in enumerate(rects):
            p, cl, sc = synth.stub_local_dets(int(x), int(y), int(y2 - y), int(x2 - x), c["seed"])
            pts.append(p); cls.append(cl); conf.append(sc); tid.append(np.full(len(cl), t, np.int32))
        pts, cls, conf, tid = np.concatenate(pts), np.concatenate(cls), np.concatenate(conf), np.concatenate(tid)
        margin = 10 if c["tile"] <= 128 else 20
        gb, ang, ins = ops.tile_postprocess(dev(pts, torch.float32), dev(cls, torch.int32), dev(tid, torch.int32), dev(rects, torch.int32), margin)
        ins = ins.bool()
        gb, ang, cl_d, cf_d, tid_d = gb[ins], ang[ins], dev(cls, torch.int32)[ins], dev(conf, torch.float32)[ins].double(), dev(tid, torch.int32)[ins]
        counts = torch.bincount(tid_d.long(), minlength=len(rects))
        seg = torch.zeros(len(rects) + 1, dtype=torch.int32, device="cuda")
        seg[1:] = torch.cumsum(counts, 0).int()
        order, keep = ops.merge_segments(gb.contiguous(), cl_d.contiguous(), cf_d.contiguous(), seg, 0.4)
        sel = order[keep.bool()].long()
        got = torch.cat([gb[sel], cl_d[sel].double()[:, None], cf_d[sel][:, None], ang[sel][:, None]], 1).cpu().numpy()
        exp = np.array(c["dets"], np.float64).reshape(-1, 11)
        assert got.shape == exp.shape, (c["tile"], got.shape, exp.shape)
        assert np.array_equal(got[:, :10], exp[:, :10])
        assert np.max(np.abs(got[:, 10] - exp[:, 10]), initial=0.0) <= 1e-9  # atan2 ulp differences only


def _cross_scale_sets(seed, sizes, extent, jitter=5.0):
    """detections of several scales over one map: every scale sees (a random subset of) the same objects, slightly displaced"""
    rng = np.random.default_rng(seed)
    base_b, base_c, _, base_x = synth.make_dets(seed, max(sizes), extent=extent, dup_frac=0.2)
    bs, cs, ss = [], [], []
    for k, n in enumerate(sizes):
        pick = rng.permutation(len(base_c))[:n]
        x = base_x[pick].copy()
        x[:, 0] += rng.normal(0, jitter, n); x[:, 1] += rng.normal(0, jitter, n)
        x[:, 2:4] *= rng.uniform(0.9, 1.1, (n, 2))
        loc = synth.xywhr_to_corners_f32(x.astype(np.float32)).astype(np.float64)
        off = base_b[pick][:, :2] - synth.xywhr_to_corners_f32(base_x[pick]).astype(np.float64)[:, :2]   # the tile offset of the base box
        loc[:, 0::2] += np.round(off[:, 0:1]); loc[:, 1::2] += np.round(off[:, 1:2])
        bs.append(loc); cs.append(base_c[pick].astype(np.int32))
        conf = rng.uniform(0.1, 1.0, n).astype(np.float32)
        conf[rng.integers(0, n, max(2, n // 10))] = conf[0]                    # exact confidence ties
        ss.append(conf.astype(np.float64))
    return np.concatenate(bs), np.concatenate(cs), np.concatenate(ss), [0] + list(np.cumsum(sizes))


@pytest.mark.parametrize("sizes,extent", [((3000, 2200), 3000.0), ((900, 700, 800), 1500.0), ((20000, 6000), 9000.0), ((600, 40), 60.0)])
def test_consensus_parallel_form_matches_the_walk(ops, sizes, extent):
    """above 512 detections the candidate edges are built by the whole chip and the greedy order is resolved in dependency rounds;
    kept indices and their order must equal the sequential walk of Detect_OBB.py:373-421 (C oracle), for 2 and 3 scales, with confidence
    ties, and -- last case: 600 + 40 boxes piled onto one spot -- when an adjacency list overflows and the walk kernel takes over"""
    b, c, s, off = _cross_scale_sets(17 + len(sizes), sizes, extent)
    exp = og.consensus_arrays(b, c, s, off)
    idx, nout = ops.consensus(dev(b, torch.float64), dev(c, torch.int32), dev(s, torch.float64), off)
    got = idx.cpu().numpy()[:int(nout.item())]
    assert len(exp) > 10
    assert np.array_equal(got, exp), (len(got), len(exp))


@pytest.mark.parametrize("name", ["Test1", "Test2"])
def test_xlsx_goldens_are_fixed_points(ops, name):
    names, boxes, conf, angle = load_xlsx_csv(name)
    cls = np.array([CLASS_IDS[n] for n in names], np.int32)
    order, keep, nk = ops.merge_detections(dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64), 0.4)
    assert np.array_equal(order.cpu().numpy(), np.arange(len(cls))) and bool(keep.all())
    idx, nout = ops.consensus(dev(boxes, torch.float64), dev(cls, torch.int32), dev(conf, torch.float64), [0, len(cls)])
    assert np.array_equal(idx.cpu().numpy()[:int(nout.item())], np.arange(len(cls)))
