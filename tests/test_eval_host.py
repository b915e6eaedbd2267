"""Evaluation writers and the VOC AP tool, CPU only: text formatting and host logic.

tests/golden/eval_writers.npz holds the exact text the COMPILED REFERENCE wrote (print_detector_detections
and print_imagenet_detections, detector.c:201-243; generator tests/golden/gen_eval_golden.py).  print_cocos is
static in the reference and cannot be called from outside, so its text is checked library-vs-oracle only
(parity unpinned for that one writer; it shares the pinned clipping code)."""
import json
import os

import numpy as np
import pytest

from sr_object_detection_amd import darknet, voc_eval
from tests.helpers import load_golden


def _read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("who", ["library", "oracle"])
def test_voc_and_imagenet_writers_match_reference_text(who, oracle, tmp_path):
    g = load_golden("eval_writers")
    boxes, probs, w, h = g["boxes"], g["probs"], int(g["w"]), int(g["h"])
    classes = probs.shape[1]
    write = darknet.write_detections if who == "library" else oracle.write_detections
    paths = [str(tmp_path / ("c%d.txt" % j)) for j in range(classes)]
    write("voc", paths, str(g["id"]), boxes, probs, w, h)
    for j in range(classes):
        assert _read(paths[j]) == bytes(g["voc_c%d" % j])
    inet = str(tmp_path / "imagenet.txt")
    write("imagenet", [inet], int(g["imagenet_id"]), boxes, probs, w, h)
    assert _read(inet) == bytes(g["imagenet"])
    assert sum(len(bytes(g["voc_c%d" % j]).splitlines()) for j in range(classes)) == int((probs != 0).sum())


def test_coco_writer_library_equals_oracle_and_is_json(oracle, tmp_path):
    g = load_golden("eval_writers")
    boxes, probs, w, h = g["boxes"], g["probs"], int(g["w"]), int(g["h"])
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    image = "/data/coco/val2014/COCO_val2014_000000000139.jpg"
    darknet.write_detections("coco", [a], image, boxes, probs, w, h)
    oracle.write_detections("coco", [b], image, boxes, probs, w, h)
    assert _read(a) == _read(b) and len(_read(a)) > 0
    rows = json.loads("[" + _read(a).decode().rstrip().rstrip(",") + "]")
    assert len(rows) == int((probs != 0).sum())
    assert all(r["image_id"] == 139 for r in rows)
    assert {r["category_id"] for r in rows} <= {1, 2, 3, 4, 5, 6}          # classes 0..5 -> COCO ids 1..6
    # (a box wholly outside the image keeps a negative width: the reference clips xmin at 0 and xmax at w only)
    assert darknet.lib().get_coco_image_id(b"COCO_val2014_000000581929.jpg") == 581929


def test_basecfg():
    L = darknet.lib()
    import ctypes as C
    p = L.basecfg(b"/a/b/2007_000042.jpg")
    assert C.cast(p, C.c_char_p).value == b"2007_000042"


def test_voc_ap_known_answers():
    # perfect ranking: precision 1 at every recall level
    assert voc_eval.voc_ap([0.5, 1.0], [1.0, 1.0]) == pytest.approx(1.0)
    assert voc_eval.voc_ap([0.5, 1.0], [1.0, 1.0], use_07_metric=True) == pytest.approx(1.0)
    # hit, miss, hit over two positives: rec .5,.5,1  prec 1,.5,.667 -> area .5*1 + .5*.667
    rec, prec = np.array([.5, .5, 1.]), np.array([1., .5, 2 / 3])
    assert voc_eval.voc_ap(rec, prec) == pytest.approx(0.5 + 0.5 * 2 / 3)
    # 11-point: thresholds 0..0.5 see max precision 1, 0.6..1.0 see 2/3
    assert voc_eval.voc_ap(rec, prec, True) == pytest.approx((6 * 1.0 + 5 * 2 / 3) / 11)
    # nothing found
    assert voc_eval.voc_ap(np.array([0.]), np.array([0.])) == 0.0


def test_evaluate_class_matching_rules():
    truth = {"a": np.array([[10, 10, 50, 50], [100, 100, 150, 150]]), "b": np.array([[0, 0, 20, 20]])}
    ids = ["a", "a", "a", "b", "c"]
    scores = [0.9, 0.8, 0.7, 0.6, 0.5]
    boxes = [[10, 10, 50, 50],        # TP
             [12, 12, 50, 50],        # same object again -> FP (already claimed)
             [100, 100, 150, 150],    # TP
             [100, 100, 120, 120],    # no overlap -> FP
             [0, 0, 5, 5]]            # image without truth -> FP
    rec, prec, ap = voc_eval.evaluate_class(ids, scores, boxes, truth)
    assert np.allclose(rec, [1 / 3, 1 / 3, 2 / 3, 2 / 3, 2 / 3])
    assert np.allclose(prec, [1, .5, 2 / 3, .5, .4])
    assert ap == pytest.approx(1 / 3 * 1 + 1 / 3 * 2 / 3)
    # a difficult box is neither TP nor FP and does not count as a positive
    rec2, prec2, ap2 = voc_eval.evaluate_class(ids[:1], scores[:1], boxes[:1], {"a": truth["a"][:1]}, {"a": [True]})
    assert np.isnan(ap2) and prec2[0] == 0


def test_map_equiv_of_identical_and_degraded_runs(tmp_path):
    g = load_golden("eval_writers")
    boxes, probs, w, h = g["boxes"], g["probs"], int(g["w"]), int(g["h"])
    names = ["n%d" % j for j in range(probs.shape[1])]
    ref, cand, bad = tmp_path / "ref", tmp_path / "cand", tmp_path / "bad"
    for d in (ref, cand, bad):
        os.makedirs(d)
    for d, bx in ((ref, boxes), (cand, boxes), (bad, boxes + np.array([60, 60, 0, 0], np.float32))):
        darknet.write_detections("voc", [str(d / ("comp4_det_test_%s.txt" % n)) for n in names], "img_1", bx, probs, w, h)
    m, aps = voc_eval.map_equiv(str(cand), str(ref), names)
    assert m == 1.0 and len(aps) > 0
    m_bad, _ = voc_eval.map_equiv(str(bad), str(ref), names)
    assert m_bad < 0.9
