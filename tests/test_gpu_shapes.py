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


@pytest.mark.parametrize("w,h,batch", [(64, 64, 2), (52, 36, 3), (33, 47, 1)])
def test_strided_convolutions_on_the_matrix_cores(oracle, workdir, w, h, batch):
    """3x3 and 1x1 convolutions with stride 2 and 3 (cfg/yolov1/yolo.cfg, resnet50.cfg, strided.cfg) take the implicit-GEMM
    kernel: the GEMM rows enumerate the output grid, the taps are centred on input (oy*stride, ox*stride).  Even and odd
    input sizes (the last tap column / row falls outside for odd ones), every layer against the oracle."""
    spec = [("conv", 16, 3, 1, "leaky"), ("conv", 32, 3, 1, "leaky", 2), ("conv", 32, 1, 1, "leaky", 2), ("conv", 48, 3, 1, "leaky"),
            ("conv", 64, 3, 1, "leaky", 3), ("conv", 30, 1, 0, "linear"),
            ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]})]
    cfg = os.path.join(workdir, "strided_%dx%d_b%d.cfg" % (w, h, batch))
    open(cfg, "w").write(zoo.cfg_text("strided", w, h, batch, spec=spec))
    wts = os.path.join(workdir, "strided_%dx%d.weights" % (w, h))
    synth.write_weights(wts, zoo.resolve(spec, w, h), 99, 4.0)
    x = synth.image_batch(batch, 3, h, w, seed=100)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    out = net.network_predict(x)
    assert [net.layer_kernel(i).startswith("conv_mfma_f32") for i in (1, 2, 4)] == [True, True, True], [net.layer_kernel(i) for i in range(net.n)]
    for i in range(net.n):
        got, want = net.pull_layer_output(i), on.layer_output(i)
        assert got.shape == want.shape and np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, net.layer_kernel(i))
    assert np.abs(out - ref).max() < TOL
    net.set_strict(True)
    assert np.array_equal(net.network_predict(x), ref)
    net.free()
    on.close()


@pytest.mark.parametrize("stem,w,h,batch", [(("conv", 64, 7, 1, "leaky", 2), 64, 48, 2),      # yolov1 / resnet50 / extraction stem
                                            (("conv", 40, 11, 0, "relu", 4, 0), 67, 59, 1),   # alexnet: no padding, bias only
                                            (("conv", 96, 5, 1, "leaky", 1), 33, 21, 3),      # three filter tiles, odd sizes
                                            (("conv", 20, 3, 1, "leaky", 2), 40, 40, 2)])     # 3x3/2 stem (tiny.cfg-like)
def test_few_channel_stem_convolutions(oracle, workdir, stem, w, h, batch):
    """first layers other than 3x3/1 run on the stem kernel (haloed input, weights in LDS) and match the oracle"""
    spec = [stem, ("conv", 32, 3, 1, "leaky"), ("conv", 30, 1, 0, "linear"),
            ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]})]
    tag = "stem_%d_%d_%dx%d" % (stem[2], stem[5], w, h)
    cfg = os.path.join(workdir, tag + ".cfg")
    open(cfg, "w").write(zoo.cfg_text(tag, w, h, batch, spec=spec))
    wts = os.path.join(workdir, tag + ".weights")
    synth.write_weights(wts, zoo.resolve(spec, w, h), 17, 4.0)
    x = synth.image_batch(batch, 3, h, w, seed=18)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    out = net.network_predict(x)
    assert net.layer_kernel(0) == "conv_stem_mfma_f32", net.layer_kernel(0)
    for i in range(net.n):
        got, want = net.pull_layer_output(i), on.layer_output(i)
        assert got.shape == want.shape and np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, net.layer_kernel(i))
    assert np.abs(out - ref).max() < TOL
    net.set_strict(True)
    assert np.array_equal(net.network_predict(x), ref)
    net.set_strict(False)
    assert np.abs(net.network_predict(x) - ref).max() < TOL      # re-planned back onto the stem kernel
    net.free()
    on.close()
