"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 4): CPU restatement of the fusion-evaluation metrics of
Detect_OBB.py beyond mAP, loop for loop:

    _match_dets_to_gts_pixel :456-480   greedy matching in LIST order (no score sort), strict `iou > best_iou`, then `best >= thr`
    _prec_rec_f1             :482-486
    evaluate_center_hit      :609-648   detection centre strictly inside a same-class, valid, unused GT polygon (first in GT order)
    _evaluate_dataset        :650-658
    _classwise_report        :660-686   (rows only; the xlsx writer is I/O)

dets: 11-tuples (x1..y4, cls, conf, angle) in pixels; gts: {"cls": int, "pts": [8 floats]}.  Pinned by the reference's own
_match_dets_to_gts_pixel / _prec_rec_f1 (AST-extracted, tests/golden/make_golden_f2.py); the Shapely point-in-polygon test is
restated (oracle/obb_oracle.c ora_point_in_quad), parity unpinned there."""
from . import geom as og


def _pts8(g):
    p = g["pts"]
    return [c for pt in p for c in pt] if len(p) == 4 else list(p)


def match_dets_to_gts_pixel(dets, gts, iou_thr=0.5):
    used = [False] * len(gts)
    tp = 0
    for det in dets:
        box1, cls1 = det[:8], int(det[8])
        best_iou, best_j = 0.0, -1
        for j, g in enumerate(gts):
            if used[j] or cls1 != g["cls"]:
                continue
            iou = og.compute_polygon_iou(box1, _pts8(g))
            if iou > best_iou:
                best_iou, best_j = iou, j
        if best_iou >= iou_thr and best_j >= 0:
            used[best_j] = True
            tp += 1
    return tp, len(dets) - tp, used.count(False)


def prec_rec_f1(tp, fp, fn):
    P = tp / (tp + fp + 1e-9)
    R = tp / (tp + fn + 1e-9)
    return P, R, 2 * P * R / (P + R + 1e-9)


def center_hit_counts(dets, gts):
    used = [False] * len(gts)
    tp = fp = 0
    for d in dets:
        cls = int(d[8])
        cx = (d[0] + d[2] + d[4] + d[6]) / 4.0  # box_center_from_xyxyxyxy :159-165
        cy = (d[1] + d[3] + d[5] + d[7]) / 4.0
        matched = False
        for j, g in enumerate(gts):
            if used[j] or g["cls"] != cls:
                continue
            if og.point_in_quad(_pts8(g), cx, cy):
                tp += 1
                used[j] = True
                matched = True
                break
        if not matched:
            fp += 1
    return tp, fp, sum(1 for u in used if not u)


def evaluate_center_hit(dets_source, gt_source, conf_thr=0.5):
    tp = fp = fn = 0
    for img, gts in gt_source.items():
        dets = [d for d in dets_source.get(img, []) if d[9] >= conf_thr]
        a, b, c = center_hit_counts(dets, gts)
        tp += a; fp += b; fn += c
    return prec_rec_f1(tp, fp, fn) + (tp, fp, fn)


def evaluate_dataset(dets_source, gt_source, conf_thr, iou_thr):
    tp = fp = fn = 0
    for img, gts in gt_source.items():
        filtered = [d for d in dets_source.get(img, []) if d[9] >= conf_thr]
        a, b, c = match_dets_to_gts_pixel(filtered, gts, iou_thr)
        tp += a; fp += b; fn += c
    return prec_rec_f1(tp, fp, fn)


def classwise_report(dets_source, gt_source, conf_thr, iou_thr, class_names=None):
    cids = sorted({int(d[8]) for dets in dets_source.values() for d in dets})
    rows = []
    for cid in cids:
        tp = fp = fn = 0
        for img, gts in gt_source.items():
            dets_c = [d for d in dets_source.get(img, []) if int(d[8]) == cid and d[9] >= conf_thr]
            a, b, c = match_dets_to_gts_pixel(dets_c, [g for g in gts if g["cls"] == cid], iou_thr)
            tp += a; fp += b; fn += c
        rows.append([cid, (class_names or {}).get(cid, str(cid)), tp, fp, fn, *prec_rec_f1(tp, fp, fn)])
    return rows
