#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own compiled CPU path.

Run in the build container (where /root/reference exists):

    make -C oracle ref && python tests/golden/gen_golden.py

For every case below this script
  1. emits the cfg text (sr_object_detection_amd.zoo) and seeded synthetic
     weights / input (sr_object_detection_amd.synth) into a scratch directory,
  2. runs oracle/_ref/ref_driver (our driver around the reference's
     parse_network_cfg / load_weights / network_predict / get_region_boxes /
     do_nms_sort, compiled from /root/reference/src_yolo2 by oracle/build_ref.sh),
  3. checks the decision margins SURVEY.md 8c asks for (no probability within
     1e-3 of the threshold, no candidate-pair IoU within 1e-3 of the NMS
     threshold, no tied non-zero scores inside a class), and
  4. stores the final tensor, per-layer statistics and the sparse detections as
     a small .npz fixture.  Fixtures hold data only (inputs are re-derived from
     the seed); none of the reference's source text is stored.
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sr_object_detection_amd import synth, zoo  # noqa: E402

REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.dirname(os.path.abspath(__file__))

# name, net, size, batch, weight seed, thresh, nms, head_gain
CASES = [
    ("mini_32_b2", "mini", 32, 2, 11, 0.6, 0.4, 4.0),
    ("mini_64_b3", "mini", 64, 3, 12, 0.7, 0.4, 4.0),
    ("mini_mfma_64_b2", "mini-mfma", 64, 2, 7, 0.5, 0.4, 4.0),
    ("tiny_yolo_voc_416_b1", "tiny-yolo-voc", 416, 1, 21, 0.2, 0.4, 4.0),
    ("tiny_yolo_voc_416_b1_kinect", "tiny-yolo-voc", 416, 1, 21, 0.24, 0.1, 4.0),
    ("yolo_416_b1", "yolo", 416, 1, 31, 0.2, 0.4, 4.0),
    ("yolo_608_b1", "yolo", 608, 1, 31, 0.2, 0.4, 4.0),
    ("darknet19_224_b1", "darknet19", 224, 1, 41, 0.0, 0.0, 1.0),
    ("yolo9000_96_b1", "yolo9000", 96, 1, 51, 0.2, 0.4, 4.0),
    ("yolo9000_96_b1_map", "yolo9000", 96, 1, 51, 0.2, 0.4, 4.0),
    ("mini_res_32_b2", "mini-res", 32, 2, 61, 0.5, 0.4, 4.0),      # [shortcut] + a stride-2 convolution
    ("mini_v1_32_b2", "mini-v1", 32, 2, 71, 0.2, 0.4, 1.0),        # YOLOv1 head: [connected] [dropout] [detection]
    ("tiny_yolo_v1_448_b1", "tiny-yolo-v1", 448, 1, 81, 0.2, 0.4, 1.0),
    ("mini_v1_local_40_b2", "mini-v1-local", 40, 2, 91, 0.2, 0.4, 1.0),
    ("mini_acts_32_b2", "mini-acts", 32, 2, 101, 0.05, 0.4, 4.0),
    ("mini_xnor_32_b2", "mini-xnor", 32, 2, 111, 0.3, 0.4, 4.0),
    ("mini_cls_75_b2", "mini-cls", 75, 2, 121, 0.0, 0.0, 1.0),
    # the BASELINE.json configurations at their full input size (tests/test_gpu_configs.py fills the configs' batch --
    # 8 / 32 / 8 per GPU / 128 -- with these images, so every batch item has a reference answer)
    ("yolo_416_b4", "yolo", 416, 4, 131, 0.2, 0.4, 4.0),
    ("yolo_608_b4", "yolo", 608, 4, 831, 0.2, 0.4, 4.0),
    ("yolo9000_544_b2", "yolo9000", 544, 2, 251, 0.2, 0.4, 4.0),
    ("darknet19_448_b8", "darknet19", 448, 8, 161, 0.0, 0.0, 1.0),
    # the DENSE case: the weights and frames bench.py times (weight seed 31, frames from image seed 0xC0FFEE on): ~280
    # detections per frame at thresh 0.2, an order of magnitude more decisions per frame than the cases above, with the
    # margins relaxed to "no decision within 2e-4" (MARGINS)
    ("yolo_608_dense_b2", "yolo", 608, 2, 31, 0.2, 0.4, 4.0),          # classifier shapes: 7x7/2, 5x5, padded / unpadded pools, strides          # xnor=1 convolutions + standalone [batchnorm]      # the nine activations the target cfgs do not use   # [crop] [batchnorm] [local] in front of the YOLOv1 head
]


# the region tensor of yolo9000 at 544x544 is 32 MB per image: the fixture keeps every OUT_STRIDE-th value (+ the sum,
# the per-layer statistics and the complete decode / NMS results)
OUT_STRIDE = {"yolo9000_544_b2": 61}


# cases whose weight seed is kept (831 gives sparse, well-separated detections at every input size) while the search
# for safe decision margins walks over the IMAGE seeds instead: frame i = synth.image_batch(seed=image_seed)[i]
VARY_IMAGES = {"yolo_608_b4", "yolo9000_544_b2", "yolo_608_dense_b2"}
# (probability vs thresh, IoU vs nms, tie) margins a seed must keep; default 1e-3 / 1e-3 / 2e-5.  The dense case cannot
# have more: ~1000 candidates per frame take hundreds of decisive IoU comparisons, and over thirty image seeds the smallest
# |IoU - nms| among them was 1e-5 .. 1e-4 and the closest scores of two overlapping boxes 6e-8 .. 1.5e-6 apart (measured
# with decisive_margins); the fixture records what the chosen frames have.
MARGINS = {"yolo_608_dense_b2": (2e-5, 1e-5, 5e-7)}
IMAGE_SEED0 = 0xC0FFEE


def materialize(tmp: str, net: str, size: int, batch: int, seed: int, head_gain: float, use_map: bool = False,
                image_seed: int = IMAGE_SEED0):
    """Write cfg, weights, input (and tree/map) into tmp; returns paths + layer table."""
    tree = mp = None
    if net == "yolo9000":
        tree = os.path.join(tmp, "syn9k.tree")
        synth.write_tree(tree, 9418)
        if use_map:
            mp = os.path.join(tmp, "syn9k.map")
            synth.write_map(mp, 200, 9418)
    cfg = os.path.join(tmp, "net.cfg")
    with open(cfg, "w") as f:
        f.write(zoo.cfg_text(net, size, size, batch, tree_path=tree, map_path=mp))
    layers = zoo.resolve(net, size)
    wts = os.path.join(tmp, "net.weights")
    synth.write_weights(wts, layers, seed, head_gain)
    x = synth.image_batch(batch, 3, size, size, seed=image_seed)
    inp = os.path.join(tmp, "input.bin")
    x.tofile(inp)
    return cfg, wts, inp, layers, x


def run_reference(tmp, cfg, wts, inp, thresh, nms, dump=0):
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    subprocess.check_call([REF_DRIVER, "net", cfg, wts, inp, tmp, repr(thresh), repr(nms), str(dump)],
                          env=env, stderr=subprocess.DEVNULL)
    meta = {}
    for line in open(os.path.join(tmp, "meta.txt")):
        k, v = line.split()
        meta[k] = float(v) if "." in v else int(v)
    stats = np.loadtxt(os.path.join(tmp, "layers.txt"), ndmin=2)
    out = np.fromfile(os.path.join(tmp, "out.bin"), dtype=np.float32)
    return meta, stats, out


def iou(a, b):
    def ov(x1, w1, x2, w2):
        return min(x1 + w1 / 2, x2 + w2 / 2) - max(x1 - w1 / 2, x2 - w2 / 2)
    w, h = ov(a[0], a[2], b[0], b[2]), ov(a[1], a[3], b[1], b[3])
    if w < 0 or h < 0:
        return 0.0
    i = w * h
    return i / (a[2] * a[3] + b[2] * b[3] - i)


def margins(boxes, pre, thresh, nms):
    """Smallest distances to a decision boundary (float64 arithmetic)."""
    boxes = boxes.astype(np.float64)
    nz = pre[pre > 0]
    m_thresh = float(np.min(np.abs(nz - thresh))) if nz.size else 1.0
    m_iou, m_tie = 1.0, 1.0
    for k in range(pre.shape[1]):
        idx = np.nonzero(pre[:, k] > 0)[0]
        if idx.size < 2:
            continue
        p = np.sort(pre[idx, k].astype(np.float64))
        m_tie = min(m_tie, float(np.min(np.diff(p))))
        if nms > 0:
            for a in range(idx.size):
                for b in range(a + 1, idx.size):
                    m_iou = min(m_iou, abs(iou(boxes[idx[a]], boxes[idx[b]]) - nms))
    return m_thresh, m_iou, m_tie


def decisive_margins(boxes, pre, nms):
    """Margins of the decisions the reference's do_nms_sort (box.c:249-277) REALLY takes on this frame, in float64: per
    class the boxes are visited in descending score order (ties: ascending index), a live box i zeroes every later box j
    with IoU > nms.  Only comparisons between two live boxes decide anything; a tie matters only between boxes that can
    suppress each other.  (The all-pairs `margins` above is hopeless on ~1000 candidates per frame.)"""
    boxes = boxes.astype(np.float64)
    m_iou, m_tie = 1.0, 1.0
    for k in range(pre.shape[1]):
        idx = np.nonzero(pre[:, k] > 0)[0]
        if idx.size < 2:
            continue
        order = idx[np.argsort(-pre[idx, k].astype(np.float64), kind="stable")]
        alive = {int(i): True for i in order}
        for a_pos, i in enumerate(order):
            if not alive[int(i)]:
                continue
            for j in order[a_pos + 1:]:
                if not alive[int(j)]:
                    continue
                v = iou(boxes[i], boxes[j])
                m_iou = min(m_iou, abs(v - nms))
                if v > nms - 1e-3:
                    m_tie = min(m_tie, abs(float(pre[i, k]) - float(pre[j, k])))
                if v > nms:
                    alive[int(j)] = False
    return m_iou, m_tie


def main():
    if not os.path.exists(REF_DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` where /root/reference exists")
    only = set(sys.argv[1:])
    for name, net, size, batch, seed, thresh, nms, gain in CASES:
        if only and name not in only:
            continue
        use_map = name.endswith("_map")
        for attempt in range(200 if name in MARGINS else 40):
            if name in VARY_IMAGES:
                if generate_case(name, net, size, batch, seed, thresh, nms, gain, use_map, IMAGE_SEED0 + 1000 * attempt):
                    break
                print("  %s: image seed %d rejected (margins), trying next" % (name, IMAGE_SEED0 + 1000 * attempt))
                continue
            if generate_case(name, net, size, batch, seed + 100 * attempt, thresh, nms, gain, use_map):
                break
            print("  %s: seed %d rejected (margins), trying next" % (name, seed + 100 * attempt))
        else:
            sys.exit("no seed with safe margins for " + name)


def generate_case(name, net, size, batch, seed, thresh, nms, gain, use_map, image_seed=IMAGE_SEED0):
    if True:
        with tempfile.TemporaryDirectory() as tmp:
            cfg, wts, inp, layers, x = materialize(tmp, net, size, batch, seed, gain, use_map, image_seed)
            meta, stats, out = run_reference(tmp, cfg, wts, inp, thresh, nms)
            fix = {
                "net": net, "size": size, "batch": batch, "seed": seed, "thresh": thresh, "nms": nms,
                "head_gain": gain, "use_map": int(use_map), "image_seed": image_seed,
                "layer_stats": stats,          # idx type w h c ow oh oc outputs n size stride sum sum2 min max
                "out": out,
                "input_checksum": np.float64(x.astype(np.float64).sum()),
            }
            if name in OUT_STRIDE:
                fix["out"] = out[::OUT_STRIDE[name]].copy()
                fix["out_stride"] = OUT_STRIDE[name]
                fix["out_sum"] = np.float64(out.astype(np.float64).sum())
            if use_map:                        # same forward tensor as the non-map case: keep the fixture small
                fix["out"] = out[:0]
                fix["out_sum"] = np.float64(out.astype(np.float64).sum())
            if meta["last_type"] in (21, 5):  # REGION / DETECTION in the reference's LAYER_TYPE enum (layer.h:33)
                total = meta["lw"] * meta["lh"] * meta["ln"]      # (a detection layer has w = h = side)
                ncls = meta["classes"]
                for b in range(batch):
                    boxes = np.fromfile(os.path.join(tmp, "boxes_%d.bin" % b), dtype=np.float32).reshape(total, 4)
                    pre = np.fromfile(os.path.join(tmp, "probs_pre_%d.bin" % b), dtype=np.float32).reshape(total, ncls)
                    post = np.fromfile(os.path.join(tmp, "probs_post_%d.bin" % b), dtype=np.float32).reshape(total, ncls)
                    if name in MARGINS:
                        mi, mtie = decisive_margins(boxes, pre, nms)
                        mt = 1.0                      # checked below with two more reference runs at thresh -/+ the margin
                    else:
                        mt, mi, mtie = margins(boxes, pre, thresh, nms)
                    print("  %s[b=%d]: pre=%d post=%d margins thresh=%.2e iou=%.2e tie=%.2e" % (
                        name, b, int((pre > 0).sum()), int((post > 0).sum()), mt, mi, mtie))
                    need = MARGINS.get(name, (1e-3, 1e-3, 2e-5))
                    if not (mt > need[0] and mi > need[1] and mtie > need[2]):
                        return False
                    fix["boxes_%d" % b] = boxes
                    r, c = np.nonzero(pre)
                    fix["pre_idx_%d" % b] = np.stack([r, c], 1).astype(np.int32)
                    fix["pre_val_%d" % b] = pre[r, c]
                    r, c = np.nonzero(post)
                    fix["post_idx_%d" % b] = np.stack([r, c], 1).astype(np.int32)
                    fix["post_val_%d" % b] = post[r, c]
                    fix["margins_%d" % b] = np.array([mt, mi, mtie])
                if name in MARGINS:
                    # no probability within the margin of the threshold <=> the reference keeps the same (box, class) set at
                    # thresh - margin and thresh + margin
                    sets = []
                    for t in (thresh - MARGINS[name][0], thresh + MARGINS[name][0]):
                        with tempfile.TemporaryDirectory() as tmp2:
                            run_reference(tmp2, cfg, wts, inp, t, nms)
                            sets.append([np.nonzero(np.fromfile(os.path.join(tmp2, "probs_pre_%d.bin" % b), dtype=np.float32))[0]
                                         for b in range(batch)])
                    if not all(np.array_equal(a, b) for a, b in zip(*sets)):
                        print("  %s: a probability within %.0e of the threshold" % (name, MARGINS[name][0]))
                        return False
                n_pre = sum(len(fix["pre_val_%d" % b]) for b in range(batch))
                n_post = sum(len(fix["post_val_%d" % b]) for b in range(batch))
                if net != "yolo9000" and (n_pre < 8 or (nms > 0 and n_post >= n_pre)):
                    return False              # too few detections / no suppression: a weak NMS test
            path = os.path.join(OUT, name + ".npz")
            np.savez_compressed(path, **fix)
            print("%s: out[%d] sum=%.6f  ref predict %.2fs -> %s (%d KB)" % (
                name, out.size, float(out.astype(np.float64).sum()), meta["predict_s"], os.path.basename(path),
                os.path.getsize(path) // 1024))
    return True


if __name__ == "__main__":
    main()
