"""GPU parity: the HIP engine, called through the C-ABI library, against the CPU
oracle and against the golden vectors the compiled reference produced.

Tolerances (BASELINE.json north_star): region-layer tensor, box coordinates and
probabilities within 1e-4 absolute of the CPU path; post-NMS detection sets
identical (same (box, class) survivors, so identical counts).  In strict mode
(reference-order VALU convolution) everything must be bit-identical."""
import numpy as np
import pytest

from sr_object_detection_amd import darknet
from tests.helpers import dense_from_sparse, load_golden, materialize

pytestmark = pytest.mark.gpu

TOL = 1e-4


def boxes_close(got, want):
    """centre x,y within 1e-4 absolute; w,h = exp(t)*anchor can be large, so 1e-4 relative above 1."""
    scale = np.maximum(1.0, np.abs(want))
    return bool((np.abs(got - want) < TOL * scale).all())

CASES = ["mini_32_b2", "mini_64_b3", "mini_mfma_64_b2", "mini_res_32_b2", "tiny_yolo_voc_416_b1", "tiny_yolo_voc_416_b1_kinect",
         "yolo_416_b1", "yolo_608_b1"]


def run_case(workdir, name, strict=False):
    g = load_golden(name)
    net_name, size, batch, seed = str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"])
    thresh, nms, gain = float(g["thresh"]), float(g["nms"]), float(g["head_gain"])
    cfg, wts, x = materialize(workdir, net_name, size, batch, seed, gain)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    if strict:
        net.set_strict(True)
    out = net.network_predict(x)
    return g, net, x, out, thresh, nms


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_golden(workdir, name):
    g, net, x, out, thresh, nms = run_case(workdir, name)
    ref = g["out"]
    assert out.shape == ref.shape
    err = np.abs(out - ref)
    # tw/th are raw logits fed to exp(); compare everything absolutely, as the north star states
    assert float(err.max()) < TOL, "max |gpu - reference| = %g at %d" % (err.max(), int(err.argmax()))
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    if name.startswith(("yolo", "tiny", "mini_mfma")):
        assert any(k.startswith("conv_mfma_f32") for k in kernels), kernels
    net.free()


@pytest.mark.parametrize("name", CASES)
def test_decode_and_nms_match_reference_golden(workdir, name):
    g, net, x, out, thresh, nms = run_case(workdir, name)
    batch = int(g["batch"])
    l = net.last
    total, classes = l.w * l.h * l.n, l.classes
    dets, counts = net.detect_resident(thresh, nms)
    for b in range(batch):
        boxes, probs = net.get_region_boxes(1, 1, thresh, batch_item=b)
        assert boxes_close(boxes, g["boxes_%d" % b])
        pre = dense_from_sparse(g["pre_idx_%d" % b], g["pre_val_%d" % b], total, classes)
        assert np.array_equal(probs > 0, pre > 0), "different set of (box, class) pairs above thresh"
        assert np.abs(probs - pre).max() < TOL
        post = darknet.do_nms_sort(boxes, probs, nms)
        gpost = dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes)
        assert np.array_equal(post > 0, gpost > 0), "NMS kept a different set"
        assert int((post > 0).sum()) == len(g["post_val_%d" % b])
        assert np.abs(post - gpost).max() < TOL
        # fused resident path: same survivors, in ascending box order
        keep = np.nonzero(gpost.max(axis=1) > thresh)[0]
        assert int(counts[b]) == keep.size
        d = dets[b]
        assert np.array_equal(d["obj_id"], gpost[keep].argmax(axis=1))
        if keep.size:
            assert np.abs(d["prob"] - gpost[keep].max(axis=1)).max() < TOL
            got = np.stack([d["x"], d["y"], d["w"], d["h"]], 1)
            assert boxes_close(got, g["boxes_%d" % b][keep])
    net.free()


@pytest.mark.parametrize("name", ["mini_32_b2", "mini_mfma_64_b2", "mini_res_32_b2", "tiny_yolo_voc_416_b1"])
def test_strict_mode_is_bit_identical_to_reference(workdir, name):
    """With the reference-order VALU convolution every layer, the decode and the NMS are bit-exact."""
    g, net, x, out, thresh, nms = run_case(workdir, name, strict=True)
    assert all(net.layer_kernel(i) == "conv_direct_f32" for i in range(net.n)
               if darknet.LAYER_TYPES[net.layer(i).type] == "CONVOLUTIONAL")
    assert np.array_equal(out, g["out"])
    l = net.last
    total, classes = l.w * l.h * l.n, l.classes
    for b in range(int(g["batch"])):
        boxes, probs = net.get_region_boxes(1, 1, thresh, batch_item=b)
        assert np.array_equal(boxes, g["boxes_%d" % b])
        assert np.array_equal(probs, dense_from_sparse(g["pre_idx_%d" % b], g["pre_val_%d" % b], total, classes))
        post = darknet.do_nms_sort(boxes, probs, nms)
        assert np.array_equal(post, dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes))
    net.free()


@pytest.mark.parametrize("name,size,batch", [("mini-mfma", 64, 2), ("mini-mfma", 96, 3), ("tiny-yolo-voc", 416, 2)])
def test_every_layer_against_oracle(oracle, workdir, name, size, batch):
    """Layer-by-layer comparison with the CPU oracle on the same seeded inputs (NCHW on both sides)."""
    cfg, wts, x = materialize(workdir, name, size, batch, 77)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    fused = net.network_predict(x)
    assert any("+maxpool2" in net.layer_kernel(i) for i in range(net.n)), "conv+maxpool fusion did not engage"
    with pytest.raises(darknet.Y2Error, match="fused"):
        net.pull_layer_output(0)
    net.set_fusion(False)              # keep every layer's full-resolution output for the comparison
    out = net.network_predict(x)
    # (not bitwise: an unfused conv on a small grid may be K-split, which re-associates the sum)
    assert np.abs(out - fused).max() < 5e-5
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    for i in range(net.n):
        got = net.pull_layer_output(i)
        want = on.layer_output(i)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() < TOL * scale, "layer %d (%s) differs" % (i, net.layer_kernel(i))
    assert np.abs(out - ref).max() < TOL
    net.free()
    on.close()


def test_fused_maxpool_is_bit_identical_to_conv_then_maxpool(workdir, monkeypatch):
    """Pooling in the conv epilogue only changes WHERE the max is taken; with the same conv kernel
    (K-splitting pinned off so both plans pick identical tiles) every value must be identical."""
    monkeypatch.setenv("Y2_CONV_KSPLIT", "1")
    for name, size, batch in (("tiny-yolo-voc", 416, 2), ("mini-mfma", 64, 3)):
        cfg, wts, x = materialize(workdir, name, size, batch, 78)
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        fused = net.network_predict(x)
        assert any("+maxpool2" in net.layer_kernel(i) for i in range(net.n))
        net.set_fusion(False)
        plain = net.network_predict(x)
        assert not any("+maxpool2" in net.layer_kernel(i) for i in range(net.n))
        assert np.array_equal(fused, plain)
        net.free()


def test_batch_items_are_independent_and_batch_can_change(oracle, workdir, monkeypatch):
    """Frame sharding relies on this: an image's result does not depend on its batch mates,
    and set_batch_network may shrink or grow the batch (re-planned lazily).  Tile shape and K-split are
    pinned so that every batch size runs the same kernels and the comparison can be bitwise."""
    monkeypatch.setenv("Y2_CONV_KSPLIT", "1")
    monkeypatch.setenv("Y2_CONV_TILE", "64x64")
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 4, 5)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    full = net.network_predict(x).reshape(4, -1)
    net.set_batch_network(1)
    for b in (0, 3):
        one = net.network_predict(x[b])
        assert np.array_equal(one, full[b])
    net.set_batch_network(6)                         # larger than the cfg's batch: the reference would overflow
    x6 = np.concatenate([x, x[:2]], 0)
    six = net.network_predict(x6).reshape(6, -1)
    assert np.array_equal(six[:4], full) and np.array_equal(six[4:], full[:2])
    net.free()


def test_resize_network_matches_oracle(oracle, workdir):
    cfg416, wts, _ = materialize(workdir, "tiny-yolo-voc", 416, 1, 9)
    cfg288, _, x = materialize(workdir, "tiny-yolo-voc", 288, 1, 9)
    net = darknet.Network.parse_network_cfg(cfg416)
    net.load_weights(wts)
    net.resize_network(288, 288)
    out = net.network_predict(x)
    on = oracle.OracleNet(cfg288, wts)
    assert np.abs(out - on.predict(x)).max() < TOL
    net.free()
    on.close()


def test_darknet19_classifier(oracle, workdir):
    g = load_golden("darknet19_224_b1")
    cfg, wts, x = materialize(workdir, "darknet19", 224, 1, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    assert out.shape == (1000,)
    assert np.abs(out - g["out"]).max() < TOL
    assert list(oracle.top_k(out, 5)) == list(oracle.top_k(g["out"], 5))
    net.free()


@pytest.mark.parametrize("name", ["yolo9000_96_b1", "yolo9000_96_b1_map"])
def test_yolo9000_tree_head(workdir, name):
    g = load_golden(name)
    use_map = bool(int(g["use_map"]))
    cfg, wts, x = materialize(workdir, "yolo9000", int(g["size"]), 1, int(g["seed"]), float(g["head_gain"]), use_map)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    if not use_map:
        assert np.abs(out - g["out"]).max() < TOL
    thresh, nms = float(g["thresh"]), float(g["nms"])
    l = net.last
    total, classes = l.w * l.h * l.n, l.classes
    boxes, probs = net.get_region_boxes(1, 1, thresh, use_map=use_map)
    assert boxes_close(boxes, g["boxes_0"])
    pre = dense_from_sparse(g["pre_idx_0"], g["pre_val_0"], total, classes)
    assert np.array_equal(probs > 0, pre > 0)
    assert np.abs(probs - pre).max() < TOL
    ncls = 200 if use_map else classes
    post = probs.copy()
    post[:, :ncls] = darknet.do_nms_sort(boxes, np.ascontiguousarray(probs[:, :ncls]), nms)
    gpost = dense_from_sparse(g["post_idx_0"], g["post_val_0"], total, classes)
    assert np.array_equal(post > 0, gpost > 0)
    net.free()


def test_detector_hand_off_matches_oracle(oracle, workdir):
    """test_detector_img (detector.c:558: resize -> predict -> decode -> nms 0.1 -> objects) end to end,
    including the 4-plane BGRA quirk (only the first 3 planes are read) and a resize."""
    g = load_golden("tiny_yolo_voc_416_b1_kinect")
    cfg, wts, x = materialize(workdir, "tiny-yolo-voc", 416, 1, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    thresh = float(g["thresh"])
    im4 = np.concatenate([x[0], np.full((1, 416, 416), 0.5, np.float32)], 0)     # B,G,R,A planes
    objs = net.test_detector_img(im4, thresh)
    total, classes = net.last.w * net.last.h * net.last.n, net.last.classes
    gpost = dense_from_sparse(g["post_idx_0"], g["post_val_0"], total, classes)
    want = oracle.test_detector_objects(g["boxes_0"], gpost, thresh)
    assert len(objs) == len(want) > 0
    for o, w in zip(objs, want):
        assert o["objClass"] == int(w[5])
        assert abs(o["prob"] - w[4]) < TOL and max(abs(o[k] - w[j]) for j, k in enumerate("xywh")) < TOL
        assert np.allclose(o["boxRGB"], w[6:9], atol=0)
    # a frame of another size goes through the device resize (image.c:1950)
    small = oracle.resize_image(x[0], 320, 240)
    sized = oracle.resize_image(small, 416, 416)
    on = oracle.OracleNet(cfg, wts)
    on.predict(sized)
    boxes, probs = on.region_boxes(0, thresh)
    post = oracle.do_nms_sort(boxes, probs, 0.1)
    want2 = oracle.test_detector_objects(boxes, post, thresh)
    objs2 = net.test_detector_img(small, thresh)
    assert len(objs2) == len(want2)
    net.free()
    on.close()


def test_denormalized_network_predicts_the_same(workdir):
    """y2_denormalize_network (darknet.c:309) folds BN into the weights: the engine must re-pack them and the
    region tensor must stay put up to the reference's own constant mismatch (sqrt(var + 1e-5) in
    convolutional_layer.c:325 vs sqrt(var) + 1e-6 in the forward pass)."""
    g = load_golden("mini_mfma_64_b2")
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 2, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    before = net.network_predict(x).copy()
    net.denormalize()
    after = net.network_predict(x)
    assert all(net.layer(i).batch_normalize == 0 for i in range(net.n))
    assert np.abs(after - before).max() < 2e-3 and not np.array_equal(after, before)
    net.free()


def test_three_frame_mean_matches_oracle(oracle, workdir):
    """Detector use_mean (yolo_v2_class.cpp:208-213): region outputs of the last three frames averaged in slot order
    (zero slots first, utils.c:420 mean_arrays), then decode + NMS on the average -- on the device, checked against
    the same sequence through the oracle (strict mode: bit-identical forward, so identical detections)."""
    import ctypes as C
    g = load_golden("mini_mfma_64_b2")
    cfg, wts, _ = materialize(workdir, "mini-mfma", 64, 1, int(g["seed"]), float(g["head_gain"]))
    from sr_object_detection_amd import synth
    frames = [synth.image_batch(1, 3, 64, 64, seed=900 + k) for k in range(4)]
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_strict(True)
    on = oracle.OracleNet(cfg, wts)
    last = on.last
    n = on.layer_info(last)["outputs"]
    slots = [np.zeros(n, np.float32) for _ in range(3)]
    thresh, nms = 0.1, 0.4
    for k, x in enumerate(frames):
        net.network_predict(x)
        dets, count = net.detect_mean(thresh, nms)
        slots[k % 3] = on.predict(x).copy()
        avg = np.zeros(n, np.float32)
        for s in slots:
            avg = (avg + s).astype(np.float32)
        avg = (avg / np.float32(3)).astype(np.float32)
        C.memmove(on.L.orc_layer_output(on.h, last), avg.ctypes.data, avg.nbytes)      # decode the averaged tensor
        boxes, probs = on.region_boxes(0, thresh)
        post = oracle.do_nms_sort(boxes, probs, nms)
        want = [(i, int(np.argmax(post[i]))) for i in range(len(boxes)) if post[i].max() > thresh]
        assert count == len(want) and (k == 0 or count > 0)
        for d, (i, c) in zip(dets, want):
            assert int(d["obj_id"]) == c and float(d["prob"]) == float(post[i, c])
            assert [float(d[key]) for key in "xywh"] == [float(v) for v in boxes[i]]
    net.free()
    on.close()


def test_autotuned_plan_matches_reference_and_is_remembered(workdir):
    """y2_set_autotune: every fp32 matrix-core convolution times its tile shapes at plan time.  The tile never changes a
    result bit; a measured K-split may differ from the modelled one, so the comparison with the reference golden keeps
    the 1e-4 bar.  A second network of the same shapes re-uses the measured choices."""
    g = load_golden("yolo_416_b1")
    cfg, wts, x = materialize(workdir, "yolo", 416, 1, int(g["seed"]), float(g["head_gain"]))
    nets = []
    for k in range(2):
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_autotune(True)
        out = net.network_predict(x)
        assert np.abs(out - g["out"]).max() < TOL
        nets.append([net.layer_kernel(i) for i in range(net.n)])
        dets, counts = net.detect_resident(float(g["thresh"]), float(g["nms"]))
        total, classes = net.last.w * net.last.h * net.last.n, net.last.classes
        gpost = dense_from_sparse(g["post_idx_0"], g["post_val_0"], total, classes)
        assert int(counts[0]) == int((gpost.max(axis=1) > float(g["thresh"])).sum())
        net.free()
    assert nets[0] == nets[1]                      # remembered per shape, not measured again
    assert all(n.startswith(("conv_mfma_f32_", "conv_first_")) for n in nets[0] if n.startswith("conv"))


def test_batch1_pooled_layers_are_k_split(workdir):
    """batch-1 inference (the Kinect caller): a conv + maxpool pair on a grid too small for 256 CUs is cut along K like
    any other layer, and still equals the unfused plan"""
    g = load_golden("tiny_yolo_voc_416_b1")
    cfg, wts, x = materialize(workdir, "tiny-yolo-voc", 416, 1, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    assert np.abs(out - g["out"]).max() < TOL
    net.free()
