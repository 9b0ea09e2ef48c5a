#!/usr/bin/env python3
"""Generates the committed golden fixtures.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

What it does:
  1. reads the reference's own end-to-end goldens Output/Test{1,2}.xlsx (zip+XML, no openpyxl) -> CSV;
  2. AST-extracts the reference's pure control-flow functions from Detect_OBB.py (the module itself cannot be
     imported: it loads weights and runs the pipeline at import time, and cv2/ultralytics/shapely are absent),
     injects the oracle's polygon IoU for `compute_polygon_iou` and a stub for `run_inference_on_crop`, and records
     their outputs on seeded synthetic inputs;
  3. copies the 9 GeoMap val label files (data) used by the AP fixtures.
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import ast
import json
import os
import re
import shutil
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle import geom as ogeom  # noqa: E402
import synth  # noqa: E402

WANT = {"compute_angle_from_bbox", "margin_for", "box_center_from_xyxyxyxy", "center_inside_safe_region",
        "merge_detections", "detect_symbols", "cross_scale_consensus_filter", "_match_dets_to_gts_pixel",
        "_prec_rec_f1", "compute_ap_from_pr", "compute_pr_for_class"}
CONST = {"iou_threshold", "APPLY_BORDER_FILTER", "MARGIN_128", "MARGIN_416", "CLASS_NAMES", "MAP_MIN_SCORE"}


def load_reference_functions(stub_infer):
    tree = ast.parse(open(os.path.join(REF, "Detect_OBB.py")).read())
    body = [n for n in tree.body
            if (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in CONST for t in n.targets))
            or (isinstance(n, ast.FunctionDef) and n.name in WANT)]
    ns = {"np": np, "compute_polygon_iou": ogeom.compute_polygon_iou, "run_inference_on_crop": stub_infer}
    exec(compile(ast.Module(body=body, type_ignores=[]), "Detect_OBB.py<ast>", "exec"), ns)
    return ns


def xlsx_rows(path):
    z = zipfile.ZipFile(path)
    s = z.read("xl/worksheets/sheet1.xml").decode()
    rows = []
    for r in re.findall(r"<row [^>]*>(.*?)</row>", s, re.S):
        cells = re.findall(r"<c [^>]*?(?:t=\"(\w+)\")?[^>]*>(.*?)</c>", r, re.S)
        vals = []
        for t, inner in cells:
            m = re.search(r"<t[^>]*>(.*?)</t>", inner, re.S)
            if m:
                vals.append(m.group(1))
            else:
                vals.append(re.search(r"<v>(.*?)</v>", inner).group(1))
        rows.append(vals)
    return rows


def main():
    # ---- 1. xlsx goldens
    for n in ("Test1", "Test2"):
        rows = xlsx_rows(os.path.join(REF, "Output", n + ".xlsx"))
        with open(os.path.join(HERE, f"xlsx_{n}.csv"), "w") as f:
            for r in rows:
                f.write(",".join(r) + "\n")
        print(n, len(rows) - 1, "rows")

    # ---- 3. val labels (data)
    dst = os.path.join(HERE, "val_labels")
    os.makedirs(dst, exist_ok=True)
    src = os.path.join(REF, "datasets/GeoMap/labels/val")
    for i, fn in enumerate(sorted(os.listdir(src))):
        shutil.copyfile(os.path.join(src, fn), os.path.join(dst, f"val_{i}.txt"))

    # ---- 2. control-flow vectors
    stub_box = {}

    def stub_infer(crop, model):
        return model(crop)

    ref = load_reference_functions(stub_infer)
    out = {}

    # merge_detections: several sizes / thresholds
    merge_cases = []
    for ci, (seed, n, ext, thr) in enumerate([(1, 0, 512, 0.4), (2, 1, 512, 0.4), (3, 37, 300, 0.4), (4, 300, 700, 0.4),
                                              (5, 1000, 1500, 0.4), (6, 1000, 1500, 0.1), (7, 2500, 2200, 0.4),
                                              (8, 257, 500, 0.7), (9, 64, 200, 0.0)]):
        boxes, cls, conf, _ = synth.make_dets(seed, n, ext) if n else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        dets = synth.dets_to_tuples(boxes, cls, conf)
        inp = list(dets)
        kept = ref["merge_detections"](inp, thr)
        out[f"merge{ci}_boxes"] = boxes
        out[f"merge{ci}_cls"] = cls
        out[f"merge{ci}_conf"] = conf
        out[f"merge{ci}_thr"] = np.float64(thr)
        out[f"merge{ci}_kept"] = np.array([int(d[10]) for d in kept], np.int32)
        out[f"merge{ci}_sorted"] = np.array([int(d[10]) for d in inp], np.int32)  # in-place sort side effect
        merge_cases.append(ci)
        print("merge", ci, n, "->", len(kept))
    out["merge_cases"] = np.array(merge_cases, np.int32)

    # cross_scale_consensus_filter
    cons_cases = []
    for ci, (seed, n1, n2, ext) in enumerate([(11, 0, 0, 400), (12, 40, 0, 400), (13, 0, 40, 400), (14, 60, 50, 400),
                                              (15, 400, 300, 1200), (16, 900, 700, 1800)]):
        b1, c1, s1, _ = synth.make_dets(seed, n1, ext) if n1 else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        # second scale: jittered copies of part of scale 1 plus fresh boxes -> genuine partners
        rng = np.random.default_rng(seed + 1000)
        b2, c2, s2, _ = synth.make_dets(seed + 500, n2, ext) if n2 else (np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0), None)
        if n1 and n2:
            m = min(n1, n2) // 2
            pick = rng.choice(n1, m, replace=False)
            b2[:m] = b1[pick] + rng.normal(0, 1.5, (m, 1)).repeat(8, 1)
            c2[:m] = c1[pick]
            tie = rng.uniform(size=m) < 0.2
            s2[:m] = np.where(tie, s1[pick], s2[:m])
        d1 = synth.dets_to_tuples(b1, c1, s1, 0)
        d2 = synth.dets_to_tuples(b2, c2, s2, 100000)
        kept = ref["cross_scale_consensus_filter"]({128: list(d1), 416: list(d2)})
        for k, v in (("b1", b1), ("c1", c1), ("s1", s1), ("b2", b2), ("c2", c2), ("s2", s2)):
            out[f"cons{ci}_{k}"] = v
        out[f"cons{ci}_kept"] = np.array([int(d[10]) for d in kept], np.int32)
        cons_cases.append(ci)
        print("consensus", ci, n1, n2, "->", len(kept))
    # single-scale passthrough
    kept = ref["cross_scale_consensus_filter"]({416: list(d2)})
    out["cons_single_kept"] = np.array([int(d[10]) for d in kept], np.int32)
    out["cons_cases"] = np.array(cons_cases, np.int32)

    # border filter / angle scalars on random points
    rng = np.random.default_rng(77)
    pts = rng.uniform(-20, 436, (400, 8)).astype(np.float32).astype(np.float64)
    pts[:40, 0::2] = np.array([10.0, 10.0, 10.0, 10.0])  # centres exactly on the inclusive margin
    ang = np.array([ref["compute_angle_from_bbox"](list(p)) for p in pts])
    ins = np.array([[ref["center_inside_safe_region"](list(p), 7, 3, w, h, ref["margin_for"](ts))
                     for (w, h, ts) in ((416, 416, 416), (128, 128, 128), (263, 175, 416), (13, 128, 128))] for p in pts])
    out["scal_pts"] = pts
    out["scal_angle"] = ang
    out["scal_inside"] = ins.astype(np.uint8)

    # AP instrument against the val labels at a fixed 4096x4096 image size
    gts = {}
    for i in range(9):
        rows = [l.split() for l in open(os.path.join(dst, f"val_{i}.txt")) if len(l.split()) == 9]
        gts[i] = [(int(r[0]), [float(v) * 4096.0 for v in r[1:]]) for r in rows]
    ap_rows = []
    rng = np.random.default_rng(99)
    for cid in range(12):
        g = {f"img{i}": [b for (c, b) in gts[i] if c == cid] for i in range(9)}
        dets = []
        for i in range(9):
            for b in g[f"img{i}"]:
                if rng.uniform() < 0.8:  # jittered true positive
                    jb = list(np.array(b) + rng.normal(0, 1.2, 8))
                    dets.append({"image_id": f"img{i}", "score": float(np.float32(rng.uniform(0.3, 1.0))), "bbox": tuple(jb)})
            for _ in range(int(rng.integers(0, 6))):  # false positives
                fb, _, _, _ = synth.make_dets(int(rng.integers(1 << 30)), 1, 4000)
                dets.append({"image_id": f"img{i}", "score": float(np.float32(rng.uniform(0.001, 0.6))), "bbox": tuple(fb[0])})
        for thr in (0.5, 0.75, 0.3):
            p, r, ap, TP, FP, FN = ref["compute_pr_for_class"](dets, g, iou_thr=thr)
            ap_rows.append({"cls": cid, "thr": thr, "ap": float(ap), "TP": int(TP), "FP": int(FP), "FN": int(FN)})
        out[f"ap_det_boxes_{cid}"] = np.array([d["bbox"] for d in dets], np.float64).reshape(-1, 8)
        out[f"ap_det_score_{cid}"] = np.array([d["score"] for d in dets], np.float64)
        out[f"ap_det_img_{cid}"] = np.array([int(d["image_id"][3:]) for d in dets], np.int32)
    json.dump(ap_rows, open(os.path.join(HERE, "ap_cases.json"), "w"), indent=0)
    print("ap rows", len(ap_rows))

    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)

    # detect_symbols with the stub model on coordinate-coded images
    ds = []
    for (H, W, ts, ov, seed) in [(807, 895, 416, 100, 5), (807, 895, 128, 30, 6), (1028, 1056, 416, 100, 7),
                                 (300, 500, 128, 30, 8), (416, 416, 416, 100, 9), (100, 90, 128, 30, 10)]:
        img = synth.coord_image(H, W)
        model = synth.StubModel(seed)
        dets = ref["detect_symbols"](img, model, ts, ov)
        ds.append({"H": H, "W": W, "tile": ts, "overlap": ov, "seed": seed,
                   "tiles": [list(c) for c in model.calls],
                   "dets": [[float(v) for v in d[:8]] + [int(d[8]), float(d[9]), float(d[10])] for d in dets]})
        print("detect_symbols", H, W, ts, "tiles", len(model.calls), "dets", len(dets))
    json.dump(ds, open(os.path.join(HERE, "detect_symbols_cases.json"), "w"))


if __name__ == "__main__":
    main()
