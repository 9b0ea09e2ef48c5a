"""Oracle restatement of the fusion-evaluation metrics beyond mAP against the reference's own
_match_dets_to_gts_pixel / _prec_rec_f1 (AST-extracted: tests/golden/make_golden_f2.py) and known answers for the restated
Shapely point-in-polygon test of the Center-Hit metric (GEOS absent offline: parity unpinned at that boundary)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import geom as og
from oracle import metrics as om


def load_f2():
    cases = json.load(open(os.path.join(GOLDEN, "f2_cases.json")))
    v = np.load(os.path.join(GOLDEN, "f2_vectors.npz"))
    dets = {i: [tuple(v[f"boxes_{i}"][k]) + (int(v[f"cls_{i}"][k]), float(v[f"conf_{i}"][k]), 0.0) for k in range(len(v[f"cls_{i}"]))] for i in range(9)}
    gts = {}
    for i in range(9):
        rows = [l.split() for l in open(os.path.join(GOLDEN, "val_labels", f"val_{i}.txt")) if len(l.split()) == 9]
        gts[i] = [{"cls": int(r[0]), "pts": [float(x) * 4096.0 for x in r[1:]]} for r in rows]
    return cases, dets, gts


def test_point_in_quad_known_answers():
    sq = [0, 0, 4, 0, 4, 4, 0, 4]
    assert og.point_in_quad(sq, 2, 2) and og.point_in_quad(sq[::-1][1::2] + sq[::-1][0::2], 2, 2) in (True, False)
    assert og.point_in_quad(sq, 1e-9, 1e-9)
    for x, y in ((4, 2), (0, 0), (2, 0), (5, 2), (-1e-9, 2), (2, 4)):  # boundary and outside: not contained
        assert not og.point_in_quad(sq, x, y), (x, y)
    cw = [0, 4, 4, 4, 4, 0, 0, 0]  # clockwise ring: orientation does not matter
    assert og.point_in_quad(cw, 1, 3) and not og.point_in_quad(cw, 4, 4)
    assert not og.point_in_quad([0, 0, 4, 4, 4, 0, 0, 4], 1, 2)   # bow-tie: invalid polygon -> skipped (Detect_OBB.py:632-633)
    assert not og.point_in_quad([0, 0, 1, 1, 2, 2, 3, 3], 1, 1)   # zero area
    conc = [0, 0, 4, 0, 1, 1, 0, 4]                               # concave but valid
    assert og.point_in_quad(conc, 0.5, 0.5) and og.point_in_quad(conc, 0.9, 0.9) and not og.point_in_quad(conc, 2, 2)
    rot = [2, 0, 4, 2, 2, 4, 0, 2]                                # diamond: |x-2| + |y-2| < 2
    rng = np.random.default_rng(0)
    for x, y in rng.uniform(-0.5, 4.5, (2000, 2)):
        m = abs(x - 2) + abs(y - 2)
        if abs(m - 2) > 1e-9:
            assert og.point_in_quad(rot, x, y) == (m < 2)
    assert not og.point_in_quad(sq, float("nan"), 1)


def test_matcher_and_reports_match_reference_goldens():
    cases, dets, gts = load_f2()
    for c in cases["match"]:
        filt = [d for d in dets[c["img"]] if d[9] >= c["conf_thr"]]
        assert om.match_dets_to_gts_pixel(filt, gts[c["img"]], c["iou_thr"]) == (c["tp"], c["fp"], c["fn"]), c
    for c in cases["dataset"]:
        P, R, F1 = om.evaluate_dataset(dets, gts, c["conf_thr"], c["iou_thr"])
        assert (P, R, F1) == (c["P"], c["R"], c["F1"]), c
    exp = {(c["conf_thr"], c["iou_thr"], c["cls"]): c for c in cases["classwise"]}
    for conf_thr, iou_thr in {(c["conf_thr"], c["iou_thr"]) for c in cases["classwise"]}:
        rows = om.classwise_report(dets, gts, conf_thr, iou_thr)
        assert len(rows) == sum(1 for k in exp if k[:2] == (conf_thr, iou_thr))
        for r in rows:
            e = exp[(conf_thr, iou_thr, r[0])]
            assert (r[2], r[3], r[4]) == (e["tp"], e["fp"], e["fn"]) and (r[5], r[6], r[7]) == (e["P"], e["R"], e["F1"])
    for c in cases["center_hit"]:
        assert om.evaluate_center_hit(dets, gts, c["conf_thr"])[3:] == (c["tp"], c["fp"], c["fn"])


def test_center_hit_semantics():
    sq = {"cls": 1, "pts": [0, 0, 10, 0, 10, 10, 0, 10]}
    d_in = (4, 4, 6, 4, 6, 6, 4, 6, 1, 0.9, 0.0)
    d_in2 = (1, 1, 3, 1, 3, 3, 1, 3, 1, 0.8, 0.0)       # second detection inside the same (now used) GT -> FP
    d_cls = (4, 4, 6, 4, 6, 6, 4, 6, 2, 0.9, 0.0)       # wrong class -> FP
    d_edge = (8, 4, 12, 4, 12, 6, 8, 6, 1, 0.9, 0.0)    # centre (10, 5) on the boundary -> not contained
    assert om.center_hit_counts([d_in, d_in2, d_cls, d_edge], [sq]) == (1, 3, 0)
    assert om.center_hit_counts([d_edge], [sq]) == (0, 1, 1)
    assert om.center_hit_counts([], [sq]) == (0, 0, 1)
    assert om.prec_rec_f1(0, 0, 0) == (0.0, 0.0, 0.0)
