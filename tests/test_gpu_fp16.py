"""fp16 storage mode (y2_set_half; BASELINE configs[4], darknet19_448 classifier on the fp16 matrix cores).

The reference has no half path, so parity is against the fp32 CPU path with the relaxed tolerance SURVEY 8(d)
sets for this config: identical top-5 classes, probabilities within 1e-2.  The kernels themselves are checked
exactly: on integer-valued data every product and partial sum is exact, so the half kernels must reproduce
the oracle's fp32 result rounded once to half (the only rounding they perform), bit for bit."""
import numpy as np
import pytest

from sr_object_detection_amd import darknet
from tests.helpers import load_golden, materialize
from tests.test_gpu_kernels import _small_int_conv_case

pytestmark = pytest.mark.gpu


def _as_half(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("cin,filters,ksize,size,batch", [(32, 128, 3, 19, 2), (64, 256, 3, 13, 1), (32, 64, 1, 16, 3),
                                                         (64, 425, 1, 13, 2), (32, 96, 3, 38, 1), (64, 300, 3, 20, 2),
                                                         (128, 512, 3, 26, 1)])
def test_f16_conv_is_exact_on_integer_data(oracle, workdir, cin, filters, ksize, size, batch):
    """conv(3->cin, first-layer kernel writing half) -> conv(cin->filters, fp16 MFMA, linear): taps, padding,
    tile edges and the K order of the implicit GEMM, bit for bit"""
    spec = [("conv", cin, 3, 0, "linear"), ("conv", filters, ksize, 0, "linear")]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9000 + cin + filters + ksize + size)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    assert net.layer_kernel(1).startswith("conv_mfma_f16"), net.layer_kernel(1)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(on.layer_output(0)).max() <= 2048 and np.abs(ref).max() < 60000      # layer 0 exact in half, no overflow
    assert np.array_equal(out, _as_half(ref))
    net.free()
    on.close()


@pytest.mark.parametrize("cin,filters,size,batch", [(32, 64, 24, 2), (64, 160, 20, 1)])
def test_f16_conv_with_fused_maxpool_is_exact(oracle, workdir, cin, filters, size, batch):
    spec = [("conv", cin, 3, 0, "linear"), ("conv", filters, 3, 0, "linear"), ("max", 2, 2)]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9500 + cin + filters + size)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    for fuse in (True, False):
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_half(True)
        net.set_fusion(fuse)
        out = net.network_predict(x)
        assert ("+maxpool2" in net.layer_kernel(1)) == fuse, net.layer_kernel(1)
        assert np.array_equal(out, _as_half(ref))          # rounding to half commutes with max
        net.free()
    on.close()


@pytest.mark.parametrize("filters,size,batch,act", [(64, 32, 2, "linear"), (64, 48, 3, "leaky"), (48, 16, 1, "linear"), (24, 32, 2, "linear"),
                                                    (64, 64, 5, "linear")])
def test_f16_c32_weights_stationary_kernel_is_exact(oracle, workdir, filters, size, batch, act):
    """the 32-channel 3x3 kernel (weights in registers, 18x18 input patches through LDS, 2x16-pixel MFMA strips in
    pool-major order): with and without the fused maxpool, one and two filter tiles, partial filter tiles, image
    borders on every side of a 16x16 tile, more tiles than workgroups per image row"""
    spec = [("conv", 32, 3, 0, "linear"), ("conv", filters, 3, 0, act), ("max", 2, 2)]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9700 + filters + size)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(on.layer_output(0)).max() <= 2048 and np.abs(on.layer_output(1)).max() < 60000
    for fuse in (True, False):
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_half(True)
        net.set_fusion(fuse)
        out = net.network_predict(x)
        assert net.layer_kernel(1) == "conv_c32_f16_16x16" + ("+maxpool2" if fuse else ""), net.layer_kernel(1)
        if act == "linear":
            assert np.array_equal(out, _as_half(ref))
            if not fuse:
                assert np.array_equal(net.pull_layer_output(1), _as_half(on.layer_output(1)))
        else:       # leaky: .1*x is evaluated in fp32 here and in double by the reference -- one half ulp at most
            assert np.abs(out - ref).max() <= np.abs(ref).max() * 2.0 ** -10
        net.free()
    on.close()


def test_f16_darknet19_classifier_top5(workdir):
    g = load_golden("darknet19_224_b1")
    cfg, wts, x = materialize(workdir, "darknet19", 224, 1, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    assert sum(k.startswith(("conv_mfma_f16", "conv_c32_f16")) for k in kernels) == 18, kernels
    want = g["out"]
    err = float(np.abs(out - want).max())
    print("darknet19 fp16 vs fp32 CPU reference: max |dp| = %.3e" % err)
    assert err < 1e-2
    assert set(np.argsort(-out)[:5]) == set(np.argsort(-want)[:5])
    assert abs(float(out.sum()) - 1.0) < 1e-4
    # switching back restores the fp32 engine exactly
    net.set_half(False)
    out32 = net.network_predict(x)
    assert np.abs(out32 - want).max() < 1e-4
    net.free()


def test_f16_detector_with_route_and_reorg(workdir):
    """mini-mfma: Cin=16 layers fall back to the direct kernel (half in/out), the concat is placed in half, the conv
    in front of the region head writes fp32"""
    g = load_golden("mini_mfma_64_b2")
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 2, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    err = float(np.abs(out - g["out"]).max())
    print("mini-mfma fp16 region tensor: max abs err %.3e" % err)
    assert err < 5e-2
    names = [net.layer_kernel(i) for i in range(net.n)]
    assert any(k.startswith("conv_mfma_f16") for k in names) and any(k == "conv_direct_f16" for k in names), names
    net.free()


def test_f16_yolo_region_tensor_close(workdir):
    g = load_golden("yolo_416_b1")
    cfg, wts, x = materialize(workdir, "yolo", 416, 1, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    err = np.abs(out - g["out"])
    print("yolo 416 fp16 region tensor: max abs err %.3e, mean %.3e" % (float(err.max()), float(err.mean())))
    assert float(err.mean()) < 2e-3
    # the confident detections survive with the same class
    thresh = float(g["thresh"])
    boxes, probs = net.get_region_boxes(1, 1, thresh)
    total, classes = probs.shape
    from tests.helpers import dense_from_sparse
    pre = dense_from_sparse(g["pre_idx_0"], g["pre_val_0"], total, classes)
    strong = pre > thresh + 0.1
    assert strong.sum() > 0 and (probs[strong] > thresh).all()
    net.free()


def test_f16_mode_errors(workdir):
    cfg, wts, x = materialize(workdir, "mini", 32, 1, 1)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    net.set_strict(True)             # strict wins: reference-order fp32
    out = net.network_predict(x)
    assert net.layer_kernel(0) == "conv_direct_f32"
    net.free()


@pytest.mark.parametrize("filters,size,batch", [(64, 14, 3), (1000, 7, 2), (72, 5, 1), (40, 2, 2), (36, 14, 2)])
def test_f16_avgpool_is_exact_on_integer_data(oracle, workdir, monkeypatch, filters, size, batch):
    """avgpool_layer.c:40-54 on half activations: the 16-byte-load kernel (8 channels per lane, the pixels in four
    contiguous quarters) and the scalar one both equal the oracle on integer data (every partial sum exact); 36 channels
    take the scalar kernel (not a multiple of 8), 1000 are darknet19's"""
    spec = [("conv", 16, 3, 0, "linear"), ("conv", filters, 1, 0, "linear"), ("avg",)]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9900 + filters + size)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(on.layer_output(1)).max() <= 2048          # the pooled tensor is exact in half
    on.close()
    for scalar in (False, True):
        if scalar:
            monkeypatch.setenv("Y2_AVGPOOL_SCALAR", "1")
        else:
            monkeypatch.delenv("Y2_AVGPOOL_SCALAR", raising=False)
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_half(True)
        out = net.network_predict(x)
        net.free()
        assert out.shape == ref.shape and np.array_equal(out, ref)


@pytest.mark.parametrize("pool", [False, True], ids=["nopool", "pool"])
@pytest.mark.parametrize("filters,size,batch", [(128, 48, 3), (96, 32, 5), (72, 64, 2)])
def test_f16_c64_weights_stationary_kernel_is_exact(oracle, workdir, monkeypatch, filters, size, batch, pool):
    """conv_c64_f16_kernel (3x3, 64 input channels, 65..128 filters, weights in registers, 18x18 input patch in LDS, 2x16
    pixel strips in pool-major order): exact against the oracle on integer data with and without the fused 2x2 maxpool,
    full and partial second filter half (96, 72), several tiles per workgroup (grid capped), image borders inside the
    patch halo (convolutional_layer.c:435-474, maxpool_layer.c:79-114); with batch-norm + leaky the same bits as the
    generic tile"""
    monkeypatch.setenv("Y2_C64_MIN_TILES", "1")
    monkeypatch.setenv("Y2_CONV_GRID", "4")
    spec = [("conv", 64, 3, 0, "linear"), ("conv", filters, 3, 0, "linear")] + ([("max", 2, 2)] if pool else [])
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9950 + filters + size + pool)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    assert net.layer_kernel(1) == "conv_c64_f16_16x16" + ("+maxpool2" if pool else ""), net.layer_kernel(1)
    net.free()
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(on.layer_output(0)).max() <= 2048 and np.abs(ref).max() < 60000
    on.close()
    assert np.array_equal(out, _as_half(ref))
    # batch-norm + leaky (the compiled-in epilogue): equal to the generic kernel's result
    spec = [("conv", 64, 3, 0, "linear"), ("conv", filters, 3, 1, "leaky")] + ([("max", 2, 2)] if pool else [])
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 9980 + filters + size + pool, neg_scale=True)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("Y2_NO_C64", "1")
        else:
            monkeypatch.delenv("Y2_NO_C64", raising=False)
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_half(True)
        outs.append(net.network_predict(x).copy())
        assert net.layer_kernel(1).startswith("conv_c64") != off
        net.free()
    assert np.array_equal(outs[0], outs[1])
