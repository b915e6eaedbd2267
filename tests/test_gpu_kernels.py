"""GPU unit tests of the individual HIP kernels against the oracle's primitives:
bit-exact for copies / comparisons / the region, decode and NMS arithmetic, and
for the convolution on integer-valued data (where fp32 sums are exact in any
order); edge cases follow SURVEY.md 8a/8c (2x2 stride-1 maxpool reading past the
edge, the reorg quirk, NMS ties, large-magnitude softmax, empty inputs)."""
import ctypes as C

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth, zoo
from tests.helpers import materialize

pytestmark = pytest.mark.gpu


class Dev:
    """Tiny device-buffer helper over the C-ABI (y2h_malloc / memcpy)."""

    def __init__(self):
        self.L = darknet.lib()
        self.bufs = []

    def put(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.L.y2h_malloc(C.byref(p), max(a.nbytes, 16)) == 0
        assert self.L.y2h_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes, None) == 0
        self.bufs.append(p)
        return p

    def empty(self, nbytes):
        p = C.c_void_p()
        assert self.L.y2h_malloc(C.byref(p), max(nbytes, 16)) == 0
        self.bufs.append(p)
        return p

    def get(self, p, shape, dtype=np.float32):
        out = np.zeros(shape, dtype=dtype)
        assert self.L.y2h_device_sync() == 0
        assert self.L.y2h_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, out.nbytes, None) == 0
        assert self.L.y2h_device_sync() == 0
        return out

    def close(self):
        for p in self.bufs:
            self.L.y2h_free(p)


@pytest.fixture()
def dev():
    d = Dev()
    d.L.y2h_set_device(0)
    yield d
    d.close()


def to_nhwc(x):   # [n,c,h,w] -> [n,h,w,c]
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def to_nchw(x):
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


@pytest.mark.parametrize("n,c,h,w", [(1, 3, 5, 7), (2, 32, 19, 19), (3, 425, 13, 13), (1, 1, 1, 1), (2, 70, 33, 65)])
def test_layout_round_trip(dev, n, c, h, w):
    L = dev.L
    x = synth.uniform(1, n * c * h * w, -1, 1).reshape(n, c, h, w)
    dx = dev.put(x)
    dy = dev.empty(x.nbytes)
    dz = dev.empty(x.nbytes)
    L.y2h_nchw_to_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.y2h_nhwc_to_nchw.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    assert L.y2h_nchw_to_nhwc(dx, dy, n, c, h, w, c, None) == 0
    assert np.array_equal(dev.get(dy, (n, h, w, c)), to_nhwc(x))
    assert L.y2h_nhwc_to_nchw(dy, c, dz, n, c, h, w, None) == 0
    assert np.array_equal(dev.get(dz, (n, c, h, w)), x)


@pytest.mark.parametrize("h,w,c,size,stride,pad", [(8, 8, 16, 2, 2, 0), (13, 13, 512, 2, 1, 0), (7, 9, 3, 2, 2, 0),
                                                   (9, 9, 8, 3, 2, 1), (5, 5, 4, 3, 1, 1), (6, 6, 5, 2, 1, 0)])
def test_maxpool_matches_oracle_bitwise(dev, oracle, h, w, c, size, stride, pad):
    L = dev.L
    n = 2
    x = synth.uniform(3, n * c * h * w, -2, 2).reshape(n, c, h, w)
    x[0, 0, 0, 0] = np.float32(-3.0e38)
    oh, ow = (h + 2 * pad) // stride, (w + 2 * pad) // stride
    want = oracle.maxpool(x, n, h, w, c, size, stride, pad).reshape(n, c, oh, ow)
    dx = dev.put(to_nhwc(x))
    dy = dev.empty(n * oh * ow * c * 4)
    L.y2h_maxpool.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
    assert L.y2h_maxpool(dx, c, dy, c, n, h, w, c, size, stride, pad, oh, ow, None) == 0
    assert np.array_equal(to_nchw(dev.get(dy, (n, oh, ow, c))), want)


@pytest.mark.parametrize("w,h,c,s", [(2, 2, 4, 2), (2, 2, 8, 2), (4, 2, 4, 2), (38, 38, 64, 2), (6, 6, 9, 3)])
@pytest.mark.parametrize("reverse", [0, 1])
def test_reorg_quirk_matches_oracle_bitwise(dev, oracle, w, h, c, s, reverse):
    L = dev.L
    n = 2
    if reverse:
        h, w = max(h // s, 1), max(w // s, 1)
    x = np.arange(n * c * h * w, dtype=np.float32).reshape(n, c, h, w)
    want = oracle.reorg(x, w, h, c, n, s, forward=reverse)
    oc, oh, ow = (c // (s * s), h * s, w * s) if reverse else (c * s * s, h // s, w // s)
    dx = dev.put(to_nhwc(x))
    dy = dev.empty(x.nbytes)
    L.y2h_reorg.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    assert L.y2h_reorg(dx, c, dy, oc, n, h, w, c, s, reverse, None) == 0
    got = to_nchw(dev.get(dy, (n, oh, ow, oc))).reshape(-1)
    assert np.array_equal(got, want.reshape(-1))
    if (w, h, c, s, reverse) == (2, 2, 4, 2, 0):      # SURVEY.md appendix C known answer (batch item 0)
        assert got[:16].tolist() == [0, 2, 8, 10, 1, 3, 9, 11, 4, 6, 12, 14, 5, 7, 13, 15]


def test_softmax_rows_matches_oracle(dev, oracle):
    L = dev.L
    rows = np.stack([np.array([1, 2, 3, 4], np.float32), np.array([1e4, -1e4, 0, 1e4], np.float32),
                     np.array([-50, -50.5, -49, -80], np.float32), np.zeros(4, np.float32)])
    dx = dev.put(rows)
    dy = dev.empty(rows.nbytes)
    L.y2h_softmax_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]
    for temp in (1.0, 2.5):
        assert L.y2h_softmax_rows(dx, dy, 4, 4, temp, None) == 0
        got = dev.get(dy, (4, 4))
        want = np.stack([oracle.softmax(r, temp) for r in rows])
        assert np.abs(got - want).max() <= 1.2e-7     # exp() differs from glibc's by at most an ulp
    assert L.y2h_softmax_rows(dx, dy, 4, 4, 1.0, None) == 0
    assert np.allclose(dev.get(dy, (4, 4))[0], [0.0320586041, 0.0871443227, 0.236882836, 0.643914282], atol=1e-9)


def _nms_case(total, classes, seed, ties=False):
    b = synth.uniform(seed, total * 4, 0.05, 0.95).reshape(total, 4)
    b[:, 2:] = b[:, 2:] * 0.5 + 0.05
    p = synth.uniform(seed + 1, total * classes, 0, 1).reshape(total, classes)
    p[p < 0.6] = 0
    if ties:
        p[p > 0] = np.round(p[p > 0] * 8) / 8        # many exactly equal scores
        b[1] = b[0]                                   # identical boxes: IoU exactly 1
    return b.astype(np.float32), p.astype(np.float32)


@pytest.mark.parametrize("total,classes,seed,ties", [(845, 20, 1, False), (1805, 80, 2, False), (64, 3, 3, True),
                                                    (300, 7, 4, True), (1, 1, 5, False), (5, 2, 6, False)])
def test_do_nms_sort_matches_oracle_bitwise(oracle, total, classes, seed, ties):
    boxes, probs = _nms_case(total, classes, seed, ties)
    for thresh in (0.1, 0.4):
        got = darknet.do_nms_sort(boxes, probs, thresh)
        want = oracle.do_nms_sort(boxes, probs, thresh)
        # also with tied scores: the kernel reproduces the order the reference's repeated stable sort gives
        assert np.array_equal(got, want)
    allzero = darknet.do_nms_sort(boxes, np.zeros_like(probs), 0.4)
    assert not allzero.any()


@pytest.mark.parametrize("total,classes,seed", [(200, 5, 11), (845, 20, 12)])
def test_do_nms_matches_oracle_bitwise(oracle, total, classes, seed):
    boxes, probs = _nms_case(total, classes, seed)
    got = darknet.do_nms(boxes, probs, 0.4)
    assert np.array_equal(got, oracle.do_nms(boxes, probs, 0.4))


def test_box_iou_known_answer(oracle):
    assert darknet.box_iou((.5, .5, .4, .4), (.6, .6, .4, .4)) == oracle.box_iou((.5, .5, .4, .4), (.6, .6, .4, .4))
    assert abs(darknet.box_iou((.5, .5, .4, .4), (.6, .6, .4, .4)) - 0.391304165) < 1e-8


def test_resize_image_matches_oracle_bitwise(oracle):
    x = synth.image_batch(1, 3, 53, 71)[0]
    for (w, h) in [(416, 416), (32, 17), (71, 53), (1 + 70, 2)]:
        assert np.array_equal(darknet.resize_image(x, w, h), oracle.resize_image(x, w, h))


def _small_int_conv_case(workdir, spec, size, batch, seed, neg_scale=False):
    """cfg + integer-valued weights/inputs so that every fp32 partial sum is exact in any order.
    neg_scale: batch-norm scales alternate +1, -1.5 (a negative scale turns the epilogue into a DEcreasing function)."""
    import os
    import struct
    layers = zoo.resolve(spec, size)
    cfg = os.path.join(workdir, "intconv_%d.cfg" % seed)
    open(cfg, "w").write(zoo.cfg_text("x", size, size, batch, spec=spec))
    wts = os.path.join(workdir, "intconv_%d.weights" % seed)
    with open(wts, "wb") as f:
        f.write(struct.pack("<iiii", 0, 1, 0, 0))
        s = seed
        for l in layers:
            if l["type"] != "convolutional":
                continue
            n, K = l["filters"], l["c"] * l["size"] ** 2
            s += 1
            f.write(np.round(synth.uniform(s, n, -2, 2)).astype(np.float32).tobytes())          # biases
            if l["batch_normalize"]:
                sc = np.full(n, 1, np.float32)
                if neg_scale:
                    sc[1::2] = -1.5
                f.write(sc.tobytes())                                                             # scales
                f.write(np.round(synth.uniform(s + 100, n, -2, 2)).astype(np.float32).tobytes())  # mean
                f.write(np.full(n, 1, np.float32).tobytes())                                      # variance
            f.write(np.round(synth.uniform(s + 200, n * K, -1.49, 1.49)).astype(np.float32).tobytes())
    x = np.round(synth.uniform(seed + 999, batch * 3 * size * size, -2, 2)).astype(np.float32).reshape(batch, 3, size, size)
    return cfg, wts, x


@pytest.mark.parametrize("filters,ksize,size,batch", [(128, 3, 19, 2), (64, 3, 24, 1), (32, 1, 16, 3), (160, 3, 13, 1),
                                                     (425, 1, 13, 2), (96, 3, 38, 1)])
def test_mfma_conv_is_exact_on_integer_data(oracle, workdir, filters, ksize, size, batch):
    """conv(3->32, direct) -> conv(32->filters, MFMA, linear, no BN): integer data makes the GEMM exact, so the
    implicit-GEMM indexing (taps, padding, tile edges, channel offsets) is checked bit for bit."""
    spec = [("conv", 32, 3, 0, "linear"), ("conv", filters, ksize, 0, "linear")]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 1000 + filters + ksize + size)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    assert net.layer_kernel(1).startswith("conv_mfma_f32"), net.layer_kernel(1)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(ref).max() < 2 ** 22            # stays in the exactly-representable integer range
    assert np.array_equal(out, ref)
    net.free()
    on.close()


@pytest.mark.parametrize("cin,filters,ksize,size,batch,force", [(64, 192, 3, 13, 1, 0), (64, 1024, 3, 13, 2, 0),
                                                                (64, 128, 3, 19, 1, 3), (64, 425, 1, 13, 1, 2),
                                                                (48, 160, 3, 26, 1, 5)])
def test_split_k_conv_is_exact_on_integer_data(oracle, workdir, monkeypatch, cin, filters, ksize, size, batch, force):
    """Small grids are cut along K (partial sums through an fp32 workspace + reduce kernel).  With integer data
    every partial sum is exact, so the split must reproduce the oracle bit for bit; `force` pins the number of
    K ranges (Y2_CONV_KSPLIT) to also cover uneven splits, 0 lets the cost model choose."""
    if force:
        monkeypatch.setenv("Y2_CONV_KSPLIT", str(force))
    spec = [("conv", cin, 3, 0, "linear"), ("conv", filters, ksize, 0, "linear")]
    cfg, wts, x = _small_int_conv_case(workdir, spec, size, batch, 7000 + cin + filters + ksize + size + force)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    assert net.layer_kernel(1).startswith("conv_mfma_f32"), net.layer_kernel(1)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    assert np.abs(ref).max() < 2 ** 22
    assert np.array_equal(out, ref)
    net.free()
    on.close()


def test_region_kernel_matches_oracle(oracle, workdir):
    cfg, wts, x = materialize(workdir, "mini", 32, 2, 3)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_strict(True)
    out = net.network_predict(x)
    on = oracle.OracleNet(cfg, wts)
    assert np.array_equal(out, on.predict(x))
    net.free()
    on.close()
