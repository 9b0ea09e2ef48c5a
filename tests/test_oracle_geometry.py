"""Pins the CPU oracle (oracle/obb_oracle.c) -- runs without a GPU.

 * polygon IoU: analytic known answers + properties (Shapely/GEOS is absent: SURVEY.md section 8(c));
 * control flow: bit-exact against vectors produced by the reference's own functions (tests/golden/make_golden.py);
 * the reference's end-to-end goldens Output/Test{1,2}.xlsx: invariants that merge_detections implies.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import CLASS_IDS, GOLDEN, load_xlsx_csv
from oracle import geom as og
import synth


def rect(cx, cy, w, h, r=0.0):
    c, s = math.cos(r), math.sin(r)
    v1 = (w / 2 * c, w / 2 * s)
    v2 = (-h / 2 * s, h / 2 * c)
    return [cx + v1[0] + v2[0], cy + v1[1] + v2[1], cx + v1[0] - v2[0], cy + v1[1] - v2[1],
            cx - v1[0] - v2[0], cy - v1[1] - v2[1], cx - v1[0] + v2[0], cy - v1[1] + v2[1]]


def test_iou_known_answers():
    sq = [0, 0, 1, 0, 1, 1, 0, 1]
    assert og.compute_polygon_iou(sq, sq) == 1.0
    assert og.compute_polygon_iou(sq, sq[::-1][1:] + sq[::-1][:1]) >= 0.0  # arbitrary re-ordering stays finite
    assert og.compute_polygon_iou(sq, [2, 2, 3, 2, 3, 3, 2, 3]) == 0.0  # disjoint
    assert og.compute_polygon_iou(sq, [1, 0, 2, 0, 2, 1, 1, 1]) == 0.0  # touching edge
    # axis-aligned overlap closed form: 0.5 x 1 overlap -> 0.5 / 1.5
    assert og.compute_polygon_iou(sq, [0.5, 0, 1.5, 0, 1.5, 1, 0.5, 1]) == pytest.approx(1 / 3, abs=1e-15)
    # unit square vs itself rotated 45 deg about its centre: inter 2(sqrt2-1), IoU 1/sqrt2
    a, b = rect(0, 0, 1, 1, 0), rect(0, 0, 1, 1, math.pi / 4)
    assert og.compute_polygon_iou(a, b) == pytest.approx(2 * (math.sqrt(2) - 1) / (2 - 2 * (math.sqrt(2) - 1)), abs=1e-14)
    assert og.compute_polygon_iou(a, b) == pytest.approx(1 / math.sqrt(2), abs=1e-14)
    # containment -> area ratio
    assert og.compute_polygon_iou(rect(5, 5, 10, 10), rect(5, 5, 2, 4, 0.3)) == pytest.approx(8 / 100, abs=1e-14)
    # opposite windings give the same answer
    cw = [0, 0, 0, 1, 1, 1, 1, 0]
    assert og.compute_polygon_iou(cw, [0.5, 0, 1.5, 0, 1.5, 1, 0.5, 1]) == pytest.approx(1 / 3, abs=1e-15)


def test_iou_invalid_polygons_give_zero():
    sq = [0, 0, 1, 0, 1, 1, 0, 1]
    bowtie = [0, 0, 1, 1, 1, 0, 0, 1]
    line = [0, 0, 1, 1, 2, 2, 3, 3]
    point = [1, 1, 1, 1, 1, 1, 1, 1]
    spike = [0, 0, 2, 0, 1, 0, 1, 1]
    nan = [0, 0, 1, 0, float("nan"), 1, 0, 1]
    for bad in (bowtie, line, point, spike, nan):
        assert og.compute_polygon_iou(sq, bad) == 0.0
        assert og.compute_polygon_iou(bad, sq) == 0.0
        assert og.compute_polygon_iou(bad, bad) == 0.0


def test_iou_concave_quads():
    # arrow-head (concave, valid) vs a square; areas by hand: arrow = 2 triangles
    arrow = [0, 0, 2, 1, 0, 2, 1, 1]  # reflex vertex at (1,1); area = 1
    sq = [0, 0, 2, 0, 2, 2, 0, 2]
    assert og.compute_polygon_iou(arrow, sq) == pytest.approx(1 / 4, abs=1e-14)
    assert og.compute_polygon_iou(sq, arrow) == pytest.approx(1 / 4, abs=1e-14)
    arrow2 = [v + 0.5 for v in arrow]  # both concave -> triangle decomposition path
    i = og.compute_polygon_iou(arrow, arrow2)
    assert 0.0 < i < 1.0 and i == pytest.approx(og.compute_polygon_iou(arrow2, arrow), abs=1e-14)
    assert og.compute_polygon_iou(arrow, arrow) == pytest.approx(1.0, abs=1e-14)


def _py_iou(b1, b2, n=400):
    """independent check: area by dense sampling is too slow -- use exact half-plane clipping in pure Python
    with a different formulation (clip b2's polygon against b1's edges, fractions via cross ratios)."""
    def area(p):
        return 0.5 * sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p)))
    P = [(b1[i], b1[i + 1]) for i in range(0, 8, 2)]
    Q = [(b2[i], b2[i + 1]) for i in range(0, 8, 2)]
    if area(P) < 0:
        P = P[::-1]
    if area(Q) < 0:
        Q = Q[::-1]
    out = Q
    for i in range(4):
        a, b = P[i], P[(i + 1) % 4]
        nx, ny = -(b[1] - a[1]), (b[0] - a[0])  # inward normal for ccw
        inp, out = out, []
        for j in range(len(inp)):
            s, e = inp[j - 1], inp[j]
            ds = (s[0] - a[0]) * nx + (s[1] - a[1]) * ny
            de = (e[0] - a[0]) * nx + (e[1] - a[1]) * ny
            if de >= 0:
                if ds < 0:
                    t = ds / (ds - de)
                    out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
                out.append(e)
            elif ds >= 0:
                t = ds / (ds - de)
                out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
        if not out:
            return 0.0
    inter = abs(area(out)) if len(out) >= 3 else 0.0
    u = abs(area(P)) + abs(area(Q)) - inter
    return inter / u if u > 0 else 0.0


def test_iou_properties_random():
    rng = np.random.default_rng(0)
    for _ in range(400):
        a = rect(rng.uniform(0, 50), rng.uniform(0, 50), rng.uniform(5, 40), rng.uniform(5, 40), rng.uniform(0, math.pi))
        b = rect(rng.uniform(0, 50), rng.uniform(0, 50), rng.uniform(5, 40), rng.uniform(5, 40), rng.uniform(0, math.pi))
        i1, i2 = og.compute_polygon_iou(a, b), og.compute_polygon_iou(b, a)
        assert 0.0 <= i1 <= 1.0 + 1e-12
        assert i1 == pytest.approx(i2, abs=1e-12)  # symmetry
        assert i1 == pytest.approx(_py_iou(a, b), abs=1e-12)  # independent formulation
        t = rng.uniform(-100, 100, 2)
        at = [v + t[k % 2] for k, v in enumerate(a)]
        bt = [v + t[k % 2] for k, v in enumerate(b)]
        assert og.compute_polygon_iou(at, bt) == pytest.approx(i1, abs=1e-11)  # translation invariance
        # cyclic relabelling of vertices is irrelevant
        assert og.compute_polygon_iou(a[2:] + a[:2], b) == pytest.approx(i1, abs=1e-12)


def test_merge_matches_reference_vectors(ref_vectors):
    rv = ref_vectors
    for ci in rv["merge_cases"]:
        boxes, cls, conf = rv[f"merge{ci}_boxes"], rv[f"merge{ci}_cls"], rv[f"merge{ci}_conf"]
        order, keep = og.merge_arrays(boxes, cls, conf, float(rv[f"merge{ci}_thr"]))
        assert np.array_equal(order, rv[f"merge{ci}_sorted"]), f"stable sort mismatch case {ci}"
        assert np.array_equal(order[keep.astype(bool)], rv[f"merge{ci}_kept"]), f"keep mismatch case {ci}"


def test_merge_list_api_sorts_in_place(ref_vectors):
    rv = ref_vectors
    dets = synth.dets_to_tuples(rv["merge3_boxes"], rv["merge3_cls"], rv["merge3_conf"])
    kept = og.merge_detections(dets, float(rv["merge3_thr"]))
    assert [int(d[10]) for d in dets] == list(rv["merge3_sorted"])  # Detect_OBB.py:183 mutates the caller's list
    assert [int(d[10]) for d in kept] == list(rv["merge3_kept"])
    assert og.merge_detections([], 0.4) == []


def test_consensus_matches_reference_vectors(ref_vectors):
    rv = ref_vectors
    for ci in rv["cons_cases"]:
        b = np.concatenate([rv[f"cons{ci}_b1"].reshape(-1, 8), rv[f"cons{ci}_b2"].reshape(-1, 8)])
        c = np.concatenate([rv[f"cons{ci}_c1"], rv[f"cons{ci}_c2"]])
        s = np.concatenate([rv[f"cons{ci}_s1"], rv[f"cons{ci}_s2"]])
        n1 = len(rv[f"cons{ci}_c1"])
        idx = og.consensus_arrays(b, c, s, [0, n1, len(c)])
        tags = np.where(idx < n1, idx, idx - n1 + 100000)
        assert np.array_equal(tags, rv[f"cons{ci}_kept"]), f"consensus mismatch case {ci}"
    ci = rv["cons_cases"][-1]
    n2 = len(rv[f"cons{ci}_c2"])
    idx = og.consensus_arrays(rv[f"cons{ci}_b2"], rv[f"cons{ci}_c2"], rv[f"cons{ci}_s2"], [0, n2])
    assert np.array_equal(idx + 100000, rv["cons_single_kept"])  # single scale = passthrough, no conf filter


def test_border_filter_and_angle_match_reference(ref_vectors):
    rv = ref_vectors
    pts = rv["scal_pts"]
    ang = np.array([og.compute_angle_from_bbox(p) for p in pts])
    # numpy.arctan2 and libm atan2 differ by 1 ulp on some inputs; the angle is a reported value, not a decision
    assert np.max(np.abs(ang - rv["scal_angle"])) <= 1e-12
    cfgs = ((416, 416, 20), (128, 128, 10), (263, 175, 20), (13, 128, 10))
    ins = np.array([[og.center_inside_safe_region(p, 7, 3, w, h, m) for (w, h, m) in cfgs] for p in pts])
    assert np.array_equal(ins.astype(np.uint8), rv["scal_inside"])


def test_tile_grid_matches_reference():
    cases = json.load(open(os.path.join(GOLDEN, "detect_symbols_cases.json")))
    for c in cases:
        rects = og.tile_grid(c["H"], c["W"], c["tile"], c["overlap"])
        exp = np.array([[x, y, x + w, y + h] for (x, y, h, w) in c["tiles"]], np.int32)
        assert np.array_equal(rects, exp)
    # SURVEY Appendix C
    assert len(og.tile_grid(807, 895, 416, 100)) == 9
    assert len(og.tile_grid(807, 895, 128, 30)) == 90
    assert len(og.tile_grid(1028, 1056, 416, 100)) == 16
    assert len(og.tile_grid(1028, 1056, 128, 30)) == 121
    assert len(og.tile_grid(5, 5, 128, 200)) == 25  # step clamps to 1 (Detect_OBB.py:211)


def _val_gts():
    gts = {}
    for i in range(9):
        rows = [l.split() for l in open(os.path.join(GOLDEN, "val_labels", f"val_{i}.txt")) if len(l.split()) == 9]
        gts[i] = [(int(r[0]), [float(v) * 4096.0 for v in r[1:]]) for r in rows]
    return gts


def test_ap_matches_reference(ref_vectors):
    rv = ref_vectors
    gts = _val_gts()
    assert sum(len(v) for v in gts.values()) == 1061
    rows = json.load(open(os.path.join(GOLDEN, "ap_cases.json")))
    for r in rows:
        cid = r["cls"]
        gb = np.array([b for i in range(9) for (c, b) in gts[i] if c == cid], np.float64).reshape(-1, 8)
        gi = np.array([i for i in range(9) for (c, b) in gts[i] if c == cid], np.int32)
        ap, tp, tot = og.ap_for_class(rv[f"ap_det_boxes_{cid}"], rv[f"ap_det_score_{cid}"], rv[f"ap_det_img_{cid}"],
                                      gb, gi, r["thr"])
        assert (int(tot[0]), int(tot[1]), int(tot[2])) == (r["TP"], r["FP"], r["FN"]), r
        assert ap == pytest.approx(r["ap"], abs=1e-12), r


@pytest.mark.parametrize("name,nrows,max_iou", [("Test1", 34, 0.0721), ("Test2", 10, 0.3356)])
def test_xlsx_golden_invariants(name, nrows, max_iou):
    names, boxes, conf, angle = load_xlsx_csv(name)
    assert len(names) == nrows
    cls = np.array([CLASS_IDS[n] for n in names], np.int32)
    assert np.all(np.diff(conf) <= 0)  # final merge_detections output order
    # idempotence: feeding the rows back through merge(0.4) / single-scale consensus returns them unchanged
    order, keep = og.merge_arrays(boxes, cls, conf, 0.4)
    assert np.array_equal(order, np.arange(nrows)) and keep.all()
    assert np.array_equal(og.consensus_arrays(boxes, cls, conf, [0, nrows]), np.arange(nrows))
    # max same-class pairwise IoU
    m = 0.0
    for i in range(nrows):
        for j in range(i + 1, nrows):
            if cls[i] == cls[j]:
                m = max(m, og.compute_polygon_iou(boxes[i], boxes[j]))
    assert m < 0.4 and m == pytest.approx(max_iou, abs=5e-4)
    # strike angle column == compute_angle_from_bbox(corners) (translation invariant), 0 for other classes
    for i in range(nrows):
        if cls[i] == 1:
            assert og.compute_angle_from_bbox(boxes[i]) == pytest.approx(angle[i], abs=1e-9)
        else:
            assert angle[i] == 0.0
    # all rows are rectangles with theta in [0, pi/2) under Ultralytics' corner convention
    d1 = np.hypot(boxes[:, 0] - boxes[:, 4], boxes[:, 1] - boxes[:, 5])
    d2 = np.hypot(boxes[:, 2] - boxes[:, 6], boxes[:, 3] - boxes[:, 7])
    assert np.max(np.abs(d1 - d2)) < 1e-3


def test_probiou_basics():
    a = [100, 100, 40, 20, 0.3]
    assert og.probiou(a, a) == pytest.approx(1.0 - math.sqrt(1e-7 + 1e-7), abs=1e-3)
    assert og.probiou(a, [500, 500, 40, 20, 0.3]) < 1e-3
    assert og.probiou(a, [104, 102, 40, 20, 0.35]) == pytest.approx(og.probiou([104, 102, 40, 20, 0.35], a), abs=1e-6)
