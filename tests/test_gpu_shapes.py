"""Shapes the BASELINE configs do not exercise: non-square inputs (the Kinect colour stream is 16:9), sizes
whose pooled maps are odd, one-image batches of a big-batch plan, and frames with no detection at all.
Every case is compared with the CPU oracle on the same seeded data, in the fast path (1e-4) and strict (bitwise)."""
import os

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth, zoo

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _case(workdir, name, w, h, batch, seed, gain=4.0):
    cfg = os.path.join(workdir, "shape_%s_%dx%d_b%d.cfg" % (name.replace("-", "_"), w, h, batch))
    open(cfg, "w").write(zoo.cfg_text(name, w, h, batch))
    wts = os.path.join(workdir, "shape_%s_s%d.weights" % (name.replace("-", "_"), seed))
    if not os.path.exists(wts):
        synth.write_weights(wts, zoo.resolve(name, w), seed, gain)
    x = synth.image_batch(batch, 3, h, w, seed=seed + 1)
    return cfg, wts, x


@pytest.mark.parametrize("name,w,h,batch", [("mini-mfma", 96, 64, 2), ("mini-mfma", 64, 160, 1), ("mini", 48, 32, 3),
                                            ("tiny-yolo-voc", 320, 192, 1), ("tiny-yolo-voc", 416, 224, 2)])
def test_non_square_networks_match_oracle(oracle, workdir, name, w, h, batch):
    cfg, wts, x = _case(workdir, name, w, h, batch, 123)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    out = net.network_predict(x)
    assert out.shape == ref.shape and np.abs(out - ref).max() < TOL
    l = net.last
    assert (l.w, l.h) == (on.layer_info(on.last)["w"], on.layer_info(on.last)["h"]) and l.w != l.h
    # decode + NMS on the non-square grid, every batch item
    for b in range(batch):
        boxes, probs = on.region_boxes(b, 0.1, w=w, h=h)
        post = oracle.do_nms_sort(boxes, probs, 0.4)
        want = [(i, int(np.argmax(post[i]))) for i in range(len(boxes)) if post[i].max() > 0.1]
        dets, counts = net.detect(x, 0.1, 0.4, img_w=w, img_h=h)
        assert int(counts[b]) == len(want)
        for d, (i, c) in zip(dets[b], want):
            assert int(d["obj_id"]) == c and abs(float(d["prob"]) - float(post[i, c])) < TOL
            assert max(abs(float(d[k]) - float(boxes[i, j])) for j, k in enumerate("xywh")) < TOL * max(1.0, float(np.abs(boxes[i]).max()))
    net.set_strict(True)
    assert np.array_equal(net.network_predict(x), ref)
    net.free()
    on.close()


def test_frames_without_detections(oracle, workdir):
    """an all-zero frame and a threshold nothing passes: zero counts, no records, no crash in sort/NMS/compaction"""
    cfg, wts, x = _case(workdir, "mini-mfma", 64, 64, 3, 321)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    x[1] = 0.0
    dets, counts = net.detect(x, 0.999999, 0.4)
    assert list(counts) == [0, 0, 0] and all(len(d) == 0 for d in dets)
    dets, counts = net.detect(x, 0.0, 0.0)          # nms 0 disables suppression, thresh 0 keeps every positive score
    on = oracle.OracleNet(cfg, wts)
    on.predict(x)
    for b in range(3):
        boxes, probs = on.region_boxes(b, 0.0)
        assert int(counts[b]) == int((probs.max(1) > 0.0).sum())
    net.free()
    on.close()


def test_set_batch_to_one_on_a_batched_cfg(oracle, workdir):
    """the Kinect application parses cfgs written for training (batch 64 / subdivisions 8) and calls
    set_batch_network(&net, 1) (KinectUtil.cpp:88): the plan must shrink and results must not change"""
    cfg, wts, x = _case(workdir, "mini-mfma", 64, 64, 8, 555)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    full = net.network_predict(x).reshape(8, -1)
    net.set_batch_network(1)
    for b in (0, 5):
        one = net.network_predict(x[b:b + 1])
        assert np.abs(one - full[b]).max() < 5e-5      # (a different batch may pick another tile / K-split)
    net.free()
