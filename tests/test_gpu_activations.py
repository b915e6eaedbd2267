"""The nine activations of activations.h:21-54 the target cfgs do not use (loggy, relie, ramp, tanh, plse, elu, stair,
hardtan, lhtan; cfg/strided.cfg is all `ramp`).  The producing kernel runs with a linear epilogue and y2h_activate_array
applies the function as a pass of its own -- the reference's own order (activations.c:95).  Checked against the golden
vector the compiled reference produced (tests/golden/mini_acts_32_b2.npz) and, layer by layer, against the oracle."""
import numpy as np
import pytest

from sr_object_detection_amd import darknet
from tests.helpers import dense_from_sparse, load_golden, materialize

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _open(workdir, name="mini_acts_32_b2"):
    g = load_golden(name)
    cfg, wts, x = materialize(workdir, str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    return g, net, x, cfg, wts


def test_forward_and_detections_match_reference_golden(workdir):
    g, net, x, _, _ = _open(workdir)
    out = net.network_predict(x)
    assert out.shape == g["out"].shape and np.abs(out - g["out"]).max() < TOL
    assert sum(net.layer_kernel(i).startswith("conv_mfma_f32") for i in range(net.n)) >= 7, [net.layer_kernel(i) for i in range(net.n)]
    l = net.last
    total, classes = l.w * l.h * l.n, l.classes
    for b in range(int(g["batch"])):
        boxes, probs = net.get_region_boxes(1, 1, float(g["thresh"]), batch_item=b)
        post = darknet.do_nms_sort(boxes, probs, float(g["nms"]))
        gpost = dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes)
        assert np.array_equal(post > 0, gpost > 0) and np.abs(post - gpost).max() < TOL
    net.free()


@pytest.mark.parametrize("name", ["mini_acts_32_b2", "mini_xnor_32_b2"])
def test_every_layer_against_oracle_and_strict_is_bit_identical(oracle, workdir, name):
    """(mini_xnor_32_b2: xnor=1 convolutions -- weights binarized to +-mean|w| per filter at upload, inputs to +-1 by
    binarize_kernel, convolutional_layer.c:443-447 -- behind standalone [batchnorm] layers, as in cfg/yolov1/xyolo.test.cfg)"""
    g, net, x, cfg, wts = _open(workdir, name)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.array_equal(ref, g["out"])
    for strict in (False, True):
        net.set_strict(strict)
        net.set_fusion(not strict)
        out = net.network_predict(x)
        for i in range(net.n):
            if net.layer_kernel(i).endswith("+maxpool2"):
                continue                                   # checked through the maxpool layer's output
            got, want = net.pull_layer_output(i), on.layer_output(i)
            if strict:
                assert np.array_equal(got, want), (i, net.layer_kernel(i))
            else:
                assert np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, net.layer_kernel(i))
        assert np.array_equal(out, ref) if strict else np.abs(out - ref).max() < TOL
    if name == "mini_xnor_32_b2":
        net.set_strict(False)
        net.network_predict(x)
        assert sum(net.layer_kernel(i).startswith("conv_mfma_f32") for i in range(net.n)) >= 3
        net.set_half(True)
        with pytest.raises(darknet.Y2Error):
            net.network_predict(x)
    net.free()
    on.close()
