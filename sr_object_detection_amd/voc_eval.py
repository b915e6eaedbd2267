"""PASCAL-VOC style average precision for the files validate_detector writes.

A Python-3 / numpy restatement of what the reference's scripts/voc_eval.py computes (voc_ap :30-61, the
greedy matching :139-196), reorganised around in-memory arrays instead of XML + pickle caches:

* detections of one class: rows (image_id, score, xmin, ymin, xmax, ymax), read from
  comp4_det_test_<class>.txt (detector.c:201 print_detector_detections);
* ground truth of one class: {image_id: boxes [k][4] (+ optional `difficult` flags)}.

The overlap is the VOC integer-pixel IoU (the +1 on widths/heights), detections are visited by decreasing
score, each ground-truth box can be claimed once, later claims are false positives, "difficult" boxes are
neither.  AP is the area under the monotone precision envelope, or the VOC-2007 11-point average.

`map_equiv` is the use this repo makes of it: the CPU reference's detections above a score threshold
stand in for ground truth, and the GPU engine's detection files are scored against them -- 1.0 means the
two detection sets are interchangeable at IoU 0.5 (BASELINE.json: "mAP-equiv vs CPU ref").
"""
from __future__ import annotations

import os
from collections import defaultdict

import numpy as np


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    rec = np.asarray(rec, dtype=np.float64)
    prec = np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            sel = rec >= t
            ap += (prec[sel].max() if sel.any() else 0.0) / 11.0
        return float(ap)
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]           # precision envelope, right to left
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def read_detection_file(path: str):
    """-> (image_ids list, scores [n], boxes [n][4]) of one comp4_det_test_<class>.txt"""
    ids, rows = [], []
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 6:
                    ids.append(parts[0])
                    rows.append([float(v) for v in parts[1:6]])
    arr = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
    return ids, arr[:, 0], arr[:, 1:5]


def _overlaps(bb: np.ndarray, gt: np.ndarray) -> np.ndarray:
    iw = np.maximum(np.minimum(gt[:, 2], bb[2]) - np.maximum(gt[:, 0], bb[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gt[:, 3], bb[3]) - np.maximum(gt[:, 1], bb[1]) + 1.0, 0.0)
    inter = iw * ih
    union = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gt[:, 2] - gt[:, 0] + 1.0) * (gt[:, 3] - gt[:, 1] + 1.0) - inter
    return inter / union


def evaluate_class(image_ids, scores, boxes, truth: dict, difficult: dict | None = None, ovthresh: float = 0.5,
                   use_07_metric: bool = False):
    """-> (recall [n], precision [n], ap).  truth: {image_id: [k][4] corner boxes}."""
    scores = np.asarray(scores, dtype=np.float64)
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    gts = {k: np.asarray(v, dtype=np.float64).reshape(-1, 4) for k, v in truth.items()}
    hard = {k: np.asarray((difficult or {}).get(k, np.zeros(len(v))), dtype=bool) for k, v in gts.items()}
    claimed = {k: np.zeros(len(v), dtype=bool) for k, v in gts.items()}
    npos = int(sum((~h).sum() for h in hard.values()))
    order = np.argsort(-scores, kind="stable")
    tp = np.zeros(len(order))
    fp = np.zeros(len(order))
    for rank, d in enumerate(order):
        gt = gts.get(image_ids[d])
        best, j = -np.inf, -1
        if gt is not None and len(gt):
            ov = _overlaps(boxes[d], gt)
            j = int(np.argmax(ov))
            best = ov[j]
        if best > ovthresh:
            if not hard[image_ids[d]][j]:
                if not claimed[image_ids[d]][j]:
                    tp[rank] = 1.0
                    claimed[image_ids[d]][j] = True
                else:
                    fp[rank] = 1.0
        else:
            fp[rank] = 1.0
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos) if npos else np.zeros_like(tp)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, (voc_ap(rec, prec, use_07_metric) if npos else float("nan"))


def mean_ap(det_dir: str, names, truth_by_class: dict, prefix: str = "comp4_det_test_", **kw):
    """mAP over the classes that have ground truth; -> (mAP, {class: ap})"""
    aps = {}
    for name in names:
        truth = truth_by_class.get(name)
        if not truth:
            continue
        ids, sc, bx = read_detection_file(os.path.join(det_dir, prefix + name + ".txt"))
        aps[name] = evaluate_class(ids, sc, bx, truth, **kw)[2]
    vals = [v for v in aps.values() if not np.isnan(v)]
    return (float(np.mean(vals)) if vals else float("nan")), aps


def truth_from_detections(det_dir: str, names, min_score: float, prefix: str = "comp4_det_test_") -> dict:
    """ground truth stand-in: every detection of the reference run scoring above min_score"""
    out = {}
    for name in names:
        ids, sc, bx = read_detection_file(os.path.join(det_dir, prefix + name + ".txt"))
        per = defaultdict(list)
        for i, s, b in zip(ids, sc, bx):
            if s > min_score:
                per[i].append(b)
        if per:
            out[name] = {k: np.asarray(v) for k, v in per.items()}
    return out


def map_equiv(candidate_dir: str, reference_dir: str, names, min_score: float = 0.2, **kw):
    """score candidate detection files against the reference run's detections taken as ground truth.
    Candidate detections below min_score are dropped as well, so identical runs give exactly 1.0."""
    truth = truth_from_detections(reference_dir, names, min_score)
    aps = {}
    for name, gt in truth.items():
        ids, sc, bx = read_detection_file(os.path.join(candidate_dir, "comp4_det_test_" + name + ".txt"))
        keep = sc > min_score
        ids = [i for i, k in zip(ids, keep) if k]
        aps[name] = evaluate_class(ids, sc[keep], bx[keep], gt, **kw)[2]
    vals = [v for v in aps.values() if not np.isnan(v)]
    return (float(np.mean(vals)) if vals else float("nan")), aps


def detections_from_dense(boxes: np.ndarray, probs: np.ndarray, thresh: float, scale_w: float, scale_h: float):
    """Detector semantics (yolo_v2_class.cpp:221-238): per box the best class, kept when its score exceeds thresh.
    boxes [total][4] relative centre form, probs [total][classes] after NMS -> rows (class, score, xmin, ymin, xmax, ymax)."""
    rows = []
    if len(boxes):
        cls = probs.argmax(1)
        sc = probs[np.arange(len(boxes)), cls]
        for i in np.nonzero(sc > thresh)[0]:
            x, y, w, h = (float(v) for v in boxes[i])
            rows.append((int(cls[i]), float(sc[i]), (x - w / 2) * scale_w, (y - h / 2) * scale_h,
                         (x + w / 2) * scale_w, (y + h / 2) * scale_h))
    return rows


def map_equiv_rows(candidate_by_image: dict, reference_by_image: dict, **kw):
    """mAP of candidate detections with the reference detections as ground truth.  Both: {image_id: rows as
    returned by detections_from_dense}.  -> (mAP, n_reference_detections)"""
    classes = sorted({r[0] for rows in reference_by_image.values() for r in rows})
    aps = []
    for c in classes:
        truth = {img: np.array([r[2:6] for r in rows if r[0] == c]) for img, rows in reference_by_image.items()
                 if any(r[0] == c for r in rows)}
        ids, sc, bx = [], [], []
        for img, rows in candidate_by_image.items():
            for r in rows:
                if r[0] == c:
                    ids.append(img); sc.append(r[1]); bx.append(r[2:6])
        aps.append(evaluate_class(ids, sc, np.array(bx).reshape(-1, 4), truth, **kw)[2])
    vals = [v for v in aps if not np.isnan(v)]
    n_ref = sum(len(v) for v in reference_by_image.values())
    return (float(np.mean(vals)) if vals else float("nan")), n_ref
