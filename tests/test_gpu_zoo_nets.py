"""The reference's other cfg files -- classifiers (resnet50, densenet201, extraction, darknet, tiny, alexnet, vgg-16,
strided) and the rest of the YOLOv1 family -- run end to end on the GPU and match the CPU oracle layer by layer.
The network structures are the zoo restatements that tests/test_capi_host.py ties to the reference's files; they run
here at reduced input sizes so that the oracle finishes in seconds (the layer types, strides, paddings, activations,
shortcut / route wiring and flattening orders are what is being checked, not the spatial size)."""
import os

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth, zoo

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _crop_to(spec, size):
    return [("crop", size, size, e[3]) if e[0] == "crop" else e for e in spec]


CASES = [
    ("resnet50", None, 64, 2), ("densenet201", None, 64, 1), ("extraction", None, 64, 2), ("darknet-ref", None, 64, 2),
    ("tiny", None, 64, 3), ("alexnet", None, 67, 2), ("vgg-16", 32, 40, 2), ("strided", 64, 72, 1),
    ("yolo-v1-small", 128, 136, 1), ("yolo-v1", None, 128, 2),
]


@pytest.mark.parametrize("name,crop,size,batch", CASES)
def test_reference_networks_match_oracle(oracle, workdir, name, crop, size, batch):
    spec = zoo.SPECS[name] if crop is None else _crop_to(zoo.SPECS[name], crop)
    tag = name.replace("-", "_")
    cfg = os.path.join(workdir, "zoo_%s.cfg" % tag)
    open(cfg, "w").write(zoo.cfg_text(name, size, size, batch, spec=spec))
    wts = os.path.join(workdir, "zoo_%s.weights" % tag)
    synth.write_weights(wts, zoo.resolve(spec, size), 23, 1.0)
    x = synth.image_batch(batch, 3, size, size, seed=29)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    out = net.network_predict(x)
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    for i in range(net.n):
        if kernels[i].endswith("+maxpool2") or darknet.LAYER_TYPES[net.layer(i).type] == "COST":
            continue                      # (a conv fused with its maxpool is checked through the maxpool layer's output;
                                          #  a [cost] layer has no inference output: network.c:173 skips it)
        got, want = net.pull_layer_output(i), on.layer_output(i)
        assert got.shape == want.shape, (i, kernels[i])
        assert np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, kernels[i])
    assert out.shape == ref.shape and np.abs(out - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
    # the heavy layers are on the matrix cores, not on the direct kernel
    direct = [i for i, k in enumerate(kernels) if k == "conv_direct_f32"]
    assert len(direct) <= (1 if name == "alexnet" else 0), (direct, kernels)        # alexnet's 5x5 convolution
    net.free()
    on.close()


def test_classifier_shapes_against_reference_golden(workdir):
    """tests/golden/mini_cls_75_b2.npz, produced by the compiled reference: 7x7/2 stem, 3x3/2 pool without padding, 5x5
    convolution, 2x2/2 pool with padding=1, strided 3x3 and 1x1 convolutions, a shortcut, two dense layers, softmax --
    the matrix-core path within 1e-4, the strict path bit for bit"""
    from tests.helpers import load_golden, materialize
    g = load_golden("mini_cls_75_b2")
    cfg, wts, x = materialize(workdir, str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    assert kernels[0] == "conv_stem_mfma_f32" and kernels[2].endswith("_k5") and "conv_direct_f32" not in kernels, kernels
    assert out.shape == g["out"].shape and np.abs(out - g["out"]).max() < TOL
    assert np.array_equal(np.argsort(-out.reshape(2, -1), axis=1)[:, :3], np.argsort(-g["out"].reshape(2, -1), axis=1)[:, :3])
    net.set_strict(True)
    assert np.array_equal(net.network_predict(x), g["out"])
    net.free()
