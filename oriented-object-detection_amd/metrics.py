"""The mAP instrument of Detect_OBB.py (row a15 of SURVEY.md section 8) on the GPU polygon-IoU kernel:

    _load_gt_as_pixels        :436-454   (labels: `cls x1 y1 ... x4 y4` normalised -> pixels; image size passed in, no cv2)
    gather_detections_and_gts :501-510
    compute_pr_for_class      :512-565   (score-ordered greedy matching, strict `iou > best_iou`, then `best_iou >= thr`)
    compute_ap_from_pr        :489-499   (all-point interpolation)
    evaluate_map              :574-607   (mean over GT-present classes, classes without detections count as AP 0)

All detection x ground-truth IoUs of one (class, image) pair come from one obb_poly_iou_matrix launch; the greedy matching is the
reference's own sequential loop over that matrix (control flow, no arithmetic)."""
import numpy as np
import torch

from . import ops

MAP_MIN_SCORE = 0.001  # Detect_OBB.py:34


def load_gt_as_pixels(label_path, w, h):
    gts = []
    with open(label_path, "r") as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) != 9:
                continue
            vals = list(map(float, parts[1:]))
            pts = [c for i in range(0, 8, 2) for c in (vals[i] * w, vals[i + 1] * h)]
            gts.append({"cls": int(parts[0]), "pts": pts})
    return gts


def compute_ap_from_pr(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def compute_pr_for_class(dets, gts, iou_thr=0.5, device=None):
    """dets: list of {"image_id", "score", "bbox"(8)}; gts: {image_id: [bbox8, ...]} -> (precision, recall, ap, TP, FP, FN)"""
    npos = sum(len(v) for v in gts.values())
    if npos == 0:
        return np.array([0.0]), np.array([0.0]), 0.0, 0, 0, 0
    dets_sorted = sorted(dets, key=lambda x: x["score"], reverse=True)
    if len(dets_sorted) == 0:
        return np.array([0.0]), np.array([0.0]), 0.0, 0, 0, npos
    dev = device or torch.device("cuda", torch.cuda.current_device())
    # one IoU matrix per image (detections of that image x its ground truths)
    by_img = {}
    for i, d in enumerate(dets_sorted):
        by_img.setdefault(d["image_id"], []).append(i)
    iou_rows = {}
    for img, idxs in by_img.items():
        g = gts.get(img, [])
        if not g:
            continue
        a = torch.tensor([list(dets_sorted[i]["bbox"][:8]) for i in idxs], dtype=torch.float64, device=dev)
        b = torch.tensor([list(x) for x in g], dtype=torch.float64, device=dev)
        m = ops.poly_iou_matrix(a, b).cpu().numpy()
        for r, i in enumerate(idxs):
            iou_rows[i] = m[r]
    tp, fp = np.zeros(len(dets_sorted)), np.zeros(len(dets_sorted))
    matched = {img: np.zeros(len(v), dtype=bool) for img, v in gts.items()}
    for i, det in enumerate(dets_sorted):
        img = det["image_id"]
        best_iou, best_j = 0.0, -1
        row = iou_rows.get(i)
        if row is not None:
            for j in range(len(row)):
                if matched[img][j]:
                    continue
                if row[j] > best_iou:
                    best_iou, best_j = row[j], j
        if best_iou >= iou_thr and best_j >= 0:
            tp[i] = 1
            matched[img][best_j] = True
        else:
            fp[i] = 1
    tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
    recall = tp_cum / (npos + 1e-9)
    precision = tp_cum / (tp_cum + fp_cum + 1e-9)
    ap = compute_ap_from_pr(recall, precision)
    return precision, recall, ap, int(tp_cum[-1]), int(fp_cum[-1]), npos - int(tp_cum[-1])


def gather_detections_and_gts(dets_source, gt_source, cls_id):
    dets, gts = [], {}
    for img, gt_boxes in gt_source.items():
        for d in dets_source.get(img, []):
            if int(d[8]) == cls_id and d[9] >= MAP_MIN_SCORE:
                dets.append({"image_id": img, "score": float(d[9]), "bbox": d[:8]})
        gts[img] = [g["pts"] for g in gt_boxes if g["cls"] == cls_id]
    return dets, gts


def evaluate_map(dets_source, gt_source, iou_list=None):
    """dets_source: {image: [11-tuples]}, gt_source: {image: [{"cls", "pts"}]} -> the reference's result dict"""
    if iou_list is None:
        iou_list = [0.5] + [round(0.5 + 0.05 * i, 2) for i in range(1, 10)]
    class_ids = sorted({int(g["cls"]) for gl in gt_source.values() for g in gl})
    per_iou = {}
    for iou in iou_list:
        aps = []
        for cid in class_ids:
            dets, gts = gather_detections_and_gts(dets_source, gt_source, cid)
            aps.append(compute_pr_for_class(dets, gts, iou_thr=iou)[2])
        per_iou[iou] = float(np.mean(aps)) if aps else 0.0
    return {"mAP@0.5": per_iou.get(0.5, 0.0), "mAP@[0.5:0.95]": float(np.mean([per_iou[i] for i in iou_list])) if iou_list else 0.0,
            "per_iou": per_iou}


# ---------------------------------------------------------------------------------------------------------------------------
# Fusion-evaluation metrics beyond mAP.  dets: 11-tuples (x1..y4, cls, conf, angle) in pixels;
# gts: [{"cls": int, "pts": 8 floats}].  All geometry runs on the GPU (one matrix launch per image); the greedy matching loops are
# the reference's own sequential control flow over those matrices.

def prec_rec_f1(tp, fp, fn):
    """Detect_OBB.py:482-486"""
    P = tp / (tp + fp + 1e-9)
    R = tp / (tp + fn + 1e-9)
    return P, R, 2 * P * R / (P + R + 1e-9)


def _dev(device):
    return device or torch.device("cuda", torch.cuda.current_device())


def match_dets_to_gts_pixel(dets, gts, iou_thr=0.5, device=None):
    """Detect_OBB.py:456-480: detections in LIST order (no score sort), strict `iou > best_iou`, then `best_iou >= thr` -> (TP, FP, FN)"""
    if not dets or not gts:
        return 0, len(dets), len(gts)
    dev = _dev(device)
    a = torch.tensor([list(d[:8]) for d in dets], dtype=torch.float64, device=dev)
    b = torch.tensor([list(g["pts"]) for g in gts], dtype=torch.float64, device=dev)
    ca = torch.tensor([int(d[8]) for d in dets], dtype=torch.int32, device=dev)
    cb = torch.tensor([int(g["cls"]) for g in gts], dtype=torch.int32, device=dev)
    iou = ops.poly_iou_matrix(a, b, ca, cb).cpu().numpy()  # 0 where the classes differ
    same = ca.cpu().numpy()[:, None] == cb.cpu().numpy()[None, :]
    used = np.zeros(len(gts), bool)
    tp = 0
    for i in range(len(dets)):
        best_iou, best_j = 0.0, -1
        row = iou[i]
        for j in np.nonzero(same[i] & ~used)[0]:
            if row[j] > best_iou:
                best_iou, best_j = row[j], j
        if best_iou >= iou_thr and best_j >= 0:
            used[best_j] = True
            tp += 1
    return tp, len(dets) - tp, int((~used).sum())


def center_hit_counts(dets, gts, device=None):
    """body of evaluate_center_hit (:609-648) for one image -> (TP, FP, FN)"""
    if not dets or not gts:
        return 0, len(dets), len(gts)
    dev = _dev(device)
    bx = np.array([list(d[:8]) for d in dets], np.float64)
    ctr = np.stack([(bx[:, 0] + bx[:, 2] + bx[:, 4] + bx[:, 6]) / 4.0, (bx[:, 1] + bx[:, 3] + bx[:, 5] + bx[:, 7]) / 4.0], 1)  # :159-165
    inside = ops.points_in_quads(torch.as_tensor(ctr, device=dev), torch.tensor([list(g["pts"]) for g in gts], dtype=torch.float64, device=dev),
                                 torch.tensor([int(d[8]) for d in dets], dtype=torch.int32, device=dev),
                                 torch.tensor([int(g["cls"]) for g in gts], dtype=torch.int32, device=dev)).cpu().numpy().astype(bool)
    used = np.zeros(len(gts), bool)
    tp = fp = 0
    for i in range(len(dets)):
        cand = np.nonzero(inside[i] & ~used)[0]
        if len(cand):  # first unused, same-class, valid GT polygon that contains the centre
            used[cand[0]] = True
            tp += 1
        else:
            fp += 1
    return tp, fp, int((~used).sum())


def evaluate_center_hit(dets_source, gt_source, conf_thr=0.5, device=None):
    """-> (P, R, F1, TP, FP, FN)"""
    tp = fp = fn = 0
    for img, gts in gt_source.items():
        dets = [d for d in dets_source.get(img, []) if d[9] >= conf_thr]
        a, b, c = center_hit_counts(dets, gts, device)
        tp += a; fp += b; fn += c
    return prec_rec_f1(tp, fp, fn) + (tp, fp, fn)


def evaluate_dataset(dets_source, gt_source, conf_thr, iou_thr, device=None):
    """Detect_OBB.py:650-658 -> (P, R, F1)"""
    tp = fp = fn = 0
    for img, gts in gt_source.items():
        filtered = [d for d in dets_source.get(img, []) if d[9] >= conf_thr]
        a, b, c = match_dets_to_gts_pixel(filtered, gts, iou_thr, device)
        tp += a; fp += b; fn += c
    return prec_rec_f1(tp, fp, fn)


def classwise_report(dets_source, gt_source, conf_thr, iou_thr, class_names=None, csv_path=None, device=None):
    """Detect_OBB.py:660-686: rows [cls_id, class, TP, FP, FN, Precision, Recall, F1] for every class that has a detection
    (CSV instead of the reference's xlsx when csv_path is given)."""
    cids = sorted({int(d[8]) for dets in dets_source.values() for d in dets})
    rows = []
    for cid in cids:
        tp = fp = fn = 0
        for img, gts in gt_source.items():
            dets_c = [d for d in dets_source.get(img, []) if int(d[8]) == cid and d[9] >= conf_thr]
            a, b, c = match_dets_to_gts_pixel(dets_c, [g for g in gts if g["cls"] == cid], iou_thr, device)
            tp += a; fp += b; fn += c
        rows.append([cid, (class_names or {}).get(cid, str(cid)), tp, fp, fn, *prec_rec_f1(tp, fp, fn)])
    if csv_path:
        with open(csv_path, "w") as f:
            f.write("cls_id,class,TP,FP,FN,Precision,Recall,F1\n")
            for r in rows:
                f.write(",".join(str(v) for v in r) + "\n")
    return rows


def run_fusion_eval(dets_source, gt_source, iou_thr=0.25, class_names=None, csv_path=None, verbose=True, device=None):
    """Detect_OBB.py:688-741 (the single-scale and the dual-scale branch report the same quantities): the reference uses `iou_thr`
    (:36, 0.25) BOTH as the confidence threshold of the P/R/F1, class-wise and Center-Hit reports and as their IoU threshold.
    dets_source: {image: [11-tuples]} (what process_image stores in all_dets_per_image, :344-345); gt_source: {image: [{"cls", "pts"}]}.
    -> dict with every number the reference prints."""
    thr = float(iou_thr)
    out = {"conf_thr": thr, "iou_thr": float(iou_thr)}
    P, R, F1 = evaluate_dataset(dets_source, gt_source, conf_thr=thr, iou_thr=iou_thr, device=device)
    out["precision"], out["recall"], out["f1"] = P, R, F1
    out["classwise"] = classwise_report(dets_source, gt_source, conf_thr=thr, iou_thr=iou_thr, class_names=class_names, csv_path=csv_path, device=device)
    ch = evaluate_center_hit(dets_source, gt_source, conf_thr=thr, device=device)
    out["center_hit"] = {"P": ch[0], "R": ch[1], "F1": ch[2], "TP": ch[3], "FP": ch[4], "FN": ch[5]}
    maps = evaluate_map(dets_source, gt_source, iou_list=list(np.arange(0.5, 0.96, 0.05)))
    out["mAP@0.5"], out["mAP@[0.5:0.95]"] = maps["mAP@0.5"], maps["mAP@[0.5:0.95]"]
    soft_list = [0.30, 0.40, 0.50, 0.60, 0.70]
    maps_soft = evaluate_map(dets_source, gt_source, iou_list=soft_list)
    out["mAP@0.3"] = maps_soft["per_iou"][0.30]
    out["mAP@[0.3:0.7]"] = float(np.mean([maps_soft["per_iou"][i] for i in soft_list]))
    if verbose:
        print(f"[Report @ {thr:.2f}] Precision={P:.3f} | Recall={R:.3f} | F1={F1:.3f}")
        print(f"[Center-Hit @ conf≥{thr:.2f}] P={ch[0]:.3f} R={ch[1]:.3f} F1={ch[2]:.3f} (TP={ch[3]}, FP={ch[4]}, FN={ch[5]})")
        print(f"mAP@0.5 = {out['mAP@0.5']:.4f}\nmAP@[0.5:0.95] = {out['mAP@[0.5:0.95]']:.4f}")
        print(f"mAP@0.3 = {out['mAP@0.3']:.4f}\nmAP@[0.3:0.7] = {out['mAP@[0.3:0.7]']:.4f}")
    return out
