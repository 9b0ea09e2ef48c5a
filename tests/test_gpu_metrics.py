"""mAP instrument (row a15) on the GPU IoU kernel vs values produced by the reference's own compute_pr_for_class."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _gts():
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import metrics
    return {i: metrics.load_gt_as_pixels(os.path.join(GOLDEN, "val_labels", f"val_{i}.txt"), 4096, 4096) for i in range(9)}, metrics


def test_ap_matches_reference_goldens(ref_vectors):
    assert torch.cuda.is_available()
    gts, metrics = _gts()
    assert sum(len(v) for v in gts.values()) == 1061
    rows = json.load(open(os.path.join(GOLDEN, "ap_cases.json")))
    aps = {}
    for r in rows:
        cid = r["cls"]
        b, s, im = ref_vectors[f"ap_det_boxes_{cid}"].reshape(-1, 8), ref_vectors[f"ap_det_score_{cid}"], ref_vectors[f"ap_det_img_{cid}"]
        dets = [{"image_id": int(im[k]), "score": float(s[k]), "bbox": tuple(b[k])} for k in range(len(s))]
        g = {i: [x["pts"] for x in gts[i] if x["cls"] == cid] for i in range(9)}
        p, rc, ap, TP, FP, FN = metrics.compute_pr_for_class(dets, g, iou_thr=r["thr"])
        assert (TP, FP, FN) == (r["TP"], r["FP"], r["FN"]), r
        assert ap == pytest.approx(r["ap"], abs=1e-12), r
        aps[(cid, r["thr"])] = ap
    # evaluate_map: mean over GT-present classes; |delta mAP@0.5| <= 0.002 is the stated bar (here: identical matches -> 0)
    dets_source = {}
    for cid in range(12):
        b, s, im = ref_vectors[f"ap_det_boxes_{cid}"].reshape(-1, 8), ref_vectors[f"ap_det_score_{cid}"], ref_vectors[f"ap_det_img_{cid}"]
        for k in range(len(s)):
            dets_source.setdefault(int(im[k]), []).append(tuple(b[k]) + (cid, float(s[k]), 0.0))
    res = metrics.evaluate_map(dets_source, gts, iou_list=[0.5, 0.75])
    present = sorted({g["cls"] for gl in gts.values() for g in gl})
    exp50 = float(np.mean([aps[(c, 0.5)] for c in present]))
    assert abs(res["mAP@0.5"] - exp50) <= 1e-12 and abs(res["per_iou"][0.75] - float(np.mean([aps[(c, 0.75)] for c in present]))) <= 1e-12
