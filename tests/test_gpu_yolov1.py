"""YOLOv1 family: [connected] [dropout] [detection] + get_detection_boxes on the GPU against the golden
vectors the compiled reference produced (tests/golden/mini_v1_32_b2.npz, tiny_yolo_v1_448_b1.npz), and the older layer
types of cfg/yolov1/yolo.cfg, yolo-small.cfg and xyolo.test.cfg -- [crop], [local], standalone [batchnorm]
(mini_v1_local_40_b2.npz).

The dense layers run as 1x1 convolutions on the matrix cores with their weights re-ordered for the NHWC producer;
strict mode uses the reference-order kernel and must be bit-identical, decode and NMS included."""
import numpy as np
import pytest

from sr_object_detection_amd import darknet
from tests.helpers import dense_from_sparse, load_golden, materialize

pytestmark = pytest.mark.gpu

TOL = 1e-4
CASES = ["mini_v1_32_b2", "tiny_yolo_v1_448_b1", "mini_v1_local_40_b2"]


def _open(workdir, name):
    g = load_golden(name)
    cfg, wts, x = materialize(workdir, str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    return g, net, x


@pytest.mark.parametrize("name", CASES)
def test_forward_decode_nms_match_reference_golden(workdir, name):
    g, net, x = _open(workdir, name)
    out = net.network_predict(x)
    assert out.shape == g["out"].shape and np.abs(out - g["out"]).max() < TOL
    kinds = [net.layer_kernel(i) for i in range(net.n)]
    if name == "mini_v1_local_40_b2":
        assert {"crop", "batchnorm", "local"} <= set(kinds), kinds
    else:
        assert any(k.startswith("conv_mfma_f32") and darknet.LAYER_TYPES[net.layer(i).type] == "CONNECTED"
                   for i, k in enumerate(kinds)), kinds
    thresh, nms = float(g["thresh"]), float(g["nms"])
    l = net.last
    total, classes = l.side * l.side * l.n, l.classes
    for b in range(int(g["batch"])):
        boxes, probs = net.get_detection_boxes(1, 1, thresh, batch_item=b)
        scale = np.maximum(1.0, np.abs(g["boxes_%d" % b]))
        assert (np.abs(boxes - g["boxes_%d" % b]) < TOL * scale).all()
        pre = dense_from_sparse(g["pre_idx_%d" % b], g["pre_val_%d" % b], total, classes)
        assert np.array_equal(probs > 0, pre > 0) and np.abs(probs - pre).max() < TOL
        post = darknet.do_nms_sort(boxes, probs, nms)
        gpost = dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes)
        assert np.array_equal(post > 0, gpost > 0)
    # the fused HBM-resident chain (decode + NMS + compaction) reports the same best-class detections
    dets, counts = net.detect(x, thresh, nms)
    for b in range(int(g["batch"])):
        gpost = dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes)
        want = [(i, int(np.argmax(gpost[i]))) for i in range(total) if gpost[i].max() > thresh]
        assert int(counts[b]) == len(want) > 0
        for d, (i, c) in zip(dets[b], want):
            assert int(d["obj_id"]) == c and abs(float(d["prob"]) - float(gpost[i, c])) < TOL
    net.free()


@pytest.mark.parametrize("name", ["mini_v1_32_b2", "mini_v1_local_40_b2"])
def test_strict_mode_is_bit_identical(workdir, name):
    g, net, x = _open(workdir, name)
    net.set_strict(True)
    out = net.network_predict(x)
    assert np.array_equal(out, g["out"])
    assert any(net.layer_kernel(i) == "connected_ref" for i in range(net.n))
    if name == "mini_v1_local_40_b2":
        assert any(net.layer_kernel(i) == "local_ref" for i in range(net.n))
    thresh, nms = float(g["thresh"]), float(g["nms"])
    l = net.last
    total, classes = l.side * l.side * l.n, l.classes
    for b in range(int(g["batch"])):
        boxes, probs = net.get_detection_boxes(1, 1, thresh, batch_item=b)
        assert np.array_equal(boxes, g["boxes_%d" % b])
        assert np.array_equal(probs, dense_from_sparse(g["pre_idx_%d" % b], g["pre_val_%d" % b], total, classes))
        post = darknet.do_nms_sort(boxes, probs, nms)
        assert np.array_equal(post, dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes))
    net.free()


@pytest.mark.parametrize("name", ["mini_v1_32_b2", "mini_v1_local_40_b2"])
def test_every_layer_of_mini_v1_against_oracle(oracle, workdir, name):
    g, net, x = _open(workdir, name)
    net.set_fusion(False)
    net.network_predict(x)
    cfg, wts, _ = materialize(workdir, str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"]), float(g["head_gain"]))
    on = oracle.OracleNet(cfg, wts)
    on.predict(x)
    for i in range(net.n):
        got, want = net.pull_layer_output(i), on.layer_output(i)
        assert np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, net.layer_kernel(i))
    net.free()
    on.close()


@pytest.mark.parametrize("batch", [1, 3, 5, 8, 11])
def test_local_layer_batches_and_unaligned_channels(oracle, workdir, batch):
    """[local] with c % 4 != 0 (scalar tap loads), a filter count that is not a multiple of 4, stride 2, and batches on
    either side of the four- and eight-images-per-pass blockings."""
    from sr_object_detection_amd import synth, zoo
    import os
    spec = [("conv", 6, 3, 1, "leaky"), ("local", 7, 3, 2, 1, "leaky"), ("local", 5, 2, 1, 0, "relu"), ("connected", 36, 0, "linear"),
            ("detection", {"classes": 4, "num": 1, "side": 2, "softmax": 0, "sqrt": 0})]
    cfg = os.path.join(workdir, "loc.cfg")
    with open(cfg, "w") as f:
        f.write(zoo.cfg_text("loc", 12, 12, batch, spec=spec))
    wts = os.path.join(workdir, "loc.weights")
    synth.write_weights(wts, zoo.resolve(spec, 12), 5, 1.0)
    x = synth.image_batch(batch, 3, 12, 12, seed=77)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    on.predict(x)
    for strict in (False, True):
        net.set_strict(strict)
        net.network_predict(x)
        for i in range(net.n):
            got, want = net.pull_layer_output(i), on.layer_output(i)
            if strict:
                assert np.array_equal(got, want), (i, net.layer_kernel(i))
            else:
                assert np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, net.layer_kernel(i))
    net.free()
    on.close()
