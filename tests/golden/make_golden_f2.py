#!/usr/bin/env python3
"""Golden fixtures for the fusion-evaluation metrics beyond mAP.  RUNS ONLY IN THE BUILD CONTAINER.

AST-extracts the reference's `_match_dets_to_gts_pixel` and `_prec_rec_f1` (Detect_OBB.py:456-486; pure Python, the oracle's polygon
IoU injected for `compute_polygon_iou`) and records their outputs on seeded synthetic detections against the 9 GeoMap val label
files (image size fixed at 4096 x 4096, as in the AP fixtures).  The Center-Hit expectations come from the oracle's restatement of
the Shapely point-in-polygon test (flagged "origin": "oracle": GEOS is absent offline).  Fixtures are data only."""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle import geom as ogeom  # noqa: E402
from oracle import metrics as ometrics  # noqa: E402
import synth  # noqa: E402


def load_ref():
    tree = ast.parse(open(os.path.join(REF, "Detect_OBB.py")).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in {"_match_dets_to_gts_pixel", "_prec_rec_f1"}]
    ns = {"np": np, "compute_polygon_iou": ogeom.compute_polygon_iou}
    exec(compile(ast.Module(body=body, type_ignores=[]), "Detect_OBB.py<ast>", "exec"), ns)
    return ns


def main():
    ref = load_ref()
    gts = {}
    for i in range(9):
        rows = [l.split() for l in open(os.path.join(HERE, "val_labels", f"val_{i}.txt")) if len(l.split()) == 9]
        gts[i] = [{"cls": int(r[0]), "pts": [(float(r[1 + 2 * k]) * 4096.0, float(r[2 + 2 * k]) * 4096.0) for k in range(4)]} for r in rows]
    rng = np.random.default_rng(4242)
    dets_source, out = {}, {}
    for i in range(9):
        dets = []
        for g in gts[i]:
            u = rng.uniform()
            flat = np.array([c for pt in g["pts"] for c in pt])
            if u < 0.75:    # jittered true positive
                dets.append(tuple(flat + rng.normal(0, 1.5, 8)) + (g["cls"], float(np.float32(rng.uniform(0.2, 1.0))), 0.0))
            elif u < 0.85:  # right place, wrong class
                dets.append(tuple(flat + rng.normal(0, 1.0, 8)) + ((g["cls"] + 1) % 12, float(np.float32(rng.uniform(0.2, 1.0))), 0.0))
            if u < 0.1:     # duplicate of the same object (second one must become a false positive)
                dets.append(tuple(flat + rng.normal(0, 2.5, 8)) + (g["cls"], float(np.float32(rng.uniform(0.2, 1.0))), 0.0))
        for _ in range(int(rng.integers(3, 12))):  # background false positives
            fb, fc, fs, _ = synth.make_dets(int(rng.integers(1 << 30)), 1, 4000)
            dets.append(tuple(fb[0]) + (int(fc[0]), float(np.float32(rng.uniform(0.05, 0.7))), 0.0))
        order = rng.permutation(len(dets))  # list order matters (no score sort in _match_dets_to_gts_pixel)
        dets = [dets[k] for k in order]
        dets_source[i] = dets
        out[f"boxes_{i}"] = np.array([d[:8] for d in dets], np.float64).reshape(-1, 8)
        out[f"cls_{i}"] = np.array([d[8] for d in dets], np.int32)
        out[f"conf_{i}"] = np.array([d[9] for d in dets], np.float64)
    cases = {"match": [], "dataset": [], "classwise": [], "center_hit": []}
    for conf_thr in (0.25, 0.5):
        for iou_thr in (0.5, 0.3):
            tot = [0, 0, 0]
            for i in range(9):
                filt = [d for d in dets_source[i] if d[9] >= conf_thr]
                tp, fp, fn = ref["_match_dets_to_gts_pixel"](filt, gts[i], iou_thr=iou_thr)
                cases["match"].append({"img": i, "conf_thr": conf_thr, "iou_thr": iou_thr, "tp": tp, "fp": fp, "fn": fn})
                tot = [tot[0] + tp, tot[1] + fp, tot[2] + fn]
            P, R, F1 = ref["_prec_rec_f1"](*tot)
            cases["dataset"].append({"conf_thr": conf_thr, "iou_thr": iou_thr, "P": P, "R": R, "F1": F1, "tp": tot[0], "fp": tot[1], "fn": tot[2]})
            cids = sorted({int(d[8]) for ds in dets_source.values() for d in ds})
            for cid in cids:  # the body of _classwise_report (:660-686) on the extracted matcher
                tp = fp = fn = 0
                for i in range(9):
                    dc = [d for d in dets_source[i] if int(d[8]) == cid and d[9] >= conf_thr]
                    a, b, c = ref["_match_dets_to_gts_pixel"](dc, [g for g in gts[i] if g["cls"] == cid], iou_thr=iou_thr)
                    tp += a; fp += b; fn += c
                P, R, F1 = ref["_prec_rec_f1"](tp, fp, fn)
                cases["classwise"].append({"conf_thr": conf_thr, "iou_thr": iou_thr, "cls": cid, "tp": tp, "fp": fp, "fn": fn, "P": P, "R": R, "F1": F1})
        gsrc = {i: [{"cls": g["cls"], "pts": [c for pt in g["pts"] for c in pt]} for g in gts[i]] for i in range(9)}
        P, R, F1, tp, fp, fn = ometrics.evaluate_center_hit(dets_source, gsrc, conf_thr)
        cases["center_hit"].append({"origin": "oracle", "conf_thr": conf_thr, "P": P, "R": R, "F1": F1, "tp": tp, "fp": fp, "fn": fn})
    # the oracle's restatement of the matcher must agree with the reference's own function on every case
    for c in cases["match"]:
        filt = [d for d in dets_source[c["img"]] if d[9] >= c["conf_thr"]]
        g = [{"cls": x["cls"], "pts": [v for pt in x["pts"] for v in pt]} for x in gts[c["img"]]]
        assert ometrics.match_dets_to_gts_pixel(filt, g, c["iou_thr"]) == (c["tp"], c["fp"], c["fn"]), c
    json.dump(cases, open(os.path.join(HERE, "f2_cases.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(HERE, "f2_vectors.npz"), **out)
    print({k: len(v) for k, v in cases.items()}, "dets", sum(len(v) for v in dets_source.values()))
    print(cases["dataset"][0], cases["center_hit"][0])


if __name__ == "__main__":
    main()
