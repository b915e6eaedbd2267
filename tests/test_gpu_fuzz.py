"""Randomised network structures against the CPU oracle: the planner's decisions (zero-copy route placement, conv+maxpool
fusion, haloed first layers, split-K, tile choice, strided / 5x5 / 1x1 shapes, shortcut wiring, separate activation
passes) are exercised in combinations no hand-written case covers.  Seeds are fixed, so a failure reproduces."""
import os

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth, zoo

pytestmark = pytest.mark.gpu

TOL = 1e-4
ACTS = ["leaky", "leaky", "leaky", "linear", "relu", "logistic", "ramp", "elu", "hardtan"]


def random_spec(rng, size):
    """a list of zoo entries ending in a region head, valid for a size x size x 3 input"""
    spec, shapes = [], []           # shapes[i] = (w, h, c) of layer i's output
    w = h = size
    c = 3

    def conv(filters, k, stride=1, act=None, bn=None):
        nonlocal w, h, c
        spec.append(("conv", filters, k, int(rng.integers(0, 2)) if bn is None else bn, act or ACTS[int(rng.integers(len(ACTS)))], stride))
        pad = k // 2
        w, h, c = (w + 2 * pad - k) // stride + 1, (h + 2 * pad - k) // stride + 1, filters
        shapes.append((w, h, c))

    first = int(rng.integers(0, 4))
    if first == 0:
        conv(int(rng.choice([16, 32, 48])), 3)                       # first-layer kernel
    elif first == 1:
        conv(int(rng.choice([16, 64])), 7, 2)                        # stem kernel
    elif first == 2:
        conv(int(rng.choice([16, 32])), 5, 1)                        # stem kernel, 5x5
    else:
        conv(int(rng.choice([8, 20])), 3, int(rng.choice([1, 2])))   # odd filter counts: direct kernel downstream
    for _ in range(int(rng.integers(4, 10))):
        kind = rng.choice(["conv3", "conv1", "conv5", "convs2", "max", "max1", "route1", "route2", "shortcut", "reorg"],
                          p=[0.28, 0.16, 0.04, 0.08, 0.12, 0.04, 0.06, 0.1, 0.08, 0.04])
        if kind == "conv3":
            conv(int(rng.choice([16, 32, 64, 96, 160])), 3)
        elif kind == "conv1":
            conv(int(rng.choice([16, 32, 64, 40])), 1)
        elif kind == "conv5":
            conv(int(rng.choice([16, 32])), 5)
        elif kind == "convs2" and min(w, h) >= 4:
            conv(int(rng.choice([32, 64])), int(rng.choice([1, 3])), 2)
        elif kind == "max" and min(w, h) >= 4:
            spec.append(("max", 2, 2)); w, h = w // 2, h // 2; shapes.append((w, h, c))
        elif kind == "max1":
            spec.append(("max", 2, 1)); shapes.append((w, h, c))
        elif kind == "route1" and len(shapes) >= 2:
            j = int(rng.integers(0, len(shapes) - 1))
            spec.append(("route", [j])); w, h, c = shapes[j]; shapes.append((w, h, c))
        elif kind == "route2":
            same = [j for j, s in enumerate(shapes[:-1]) if s[:2] == (w, h)]
            if same:
                j = int(rng.choice(same))
                spec.append(("route", [-1, j])); c = c + shapes[j][2]; shapes.append((w, h, c))
        elif kind == "shortcut" and len(shapes) >= 2:
            j = int(rng.integers(0, len(shapes) - 1))
            sw, sh, _ = shapes[j]
            if sw and (sw // w if sw >= w else w // sw) == (sh // h if sh >= h else h // sh) and ((sw >= w) == (sh >= h)):
                spec.append(("shortcut", j, ACTS[int(rng.integers(len(ACTS)))])); shapes.append((w, h, c))
        elif kind == "reorg" and w % 2 == 0 and h % 2 == 0 and c % 4 == 0 and min(w, h) >= 4:
            spec.append(("reorg", 2)); w, h, c = w // 2, h // 2, c * 4; shapes.append((w, h, c))
    conv(int(rng.choice([32, 64])), 3, 1, "leaky", 1)
    conv(30, 1, 1, "linear", 0)
    spec.append(("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}))
    return spec


# Y2_FUZZ_SEEDS=N widens the sweep (a one-off soak run; the default 24 keep the suite short)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("Y2_FUZZ_SEEDS", "24")))))
def test_random_networks_match_oracle(oracle, workdir, seed):
    rng = np.random.default_rng(1000 + seed)
    size = int(rng.choice([32, 48, 64]))
    batch = int(rng.choice([1, 2, 3, 5]))
    spec = random_spec(rng, size)
    cfg = os.path.join(workdir, "fuzz_%d.cfg" % seed)
    open(cfg, "w").write(zoo.cfg_text("fuzz", size, size, batch, spec=spec))
    wts = os.path.join(workdir, "fuzz_%d.weights" % seed)
    synth.write_weights(wts, zoo.resolve(spec, size), seed, 2.0)
    x = synth.image_batch(batch, 3, size, size, seed=seed + 7)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    out = net.network_predict(x)
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    scale = max(1.0, float(np.abs(ref).max()))
    assert out.shape == ref.shape and np.abs(out - ref).max() < TOL * scale, (spec, kernels)
    net.set_fusion(False)
    net.network_predict(x)
    for i in range(net.n):
        got, want = net.pull_layer_output(i), on.layer_output(i)
        assert got.shape == want.shape and np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), (i, kernels[i], spec)
    net.set_strict(True)
    assert np.array_equal(net.network_predict(x), ref), (spec,)
    net.free()
    on.close()


def random_spec_f16(rng, size):
    """structures the fp16 mode takes: 3-channel first layer, then 3x3 / 1x1 convolutions with 32-multiple channels,
    2x2 maxpools (fused and not), routes, reorg; leaky / linear / relu / logistic"""
    spec, shapes = [], []
    w = h = size
    c = 3
    acts = ["leaky", "leaky", "linear", "relu", "logistic"]

    def conv(filters, k, act=None, bn=None):
        nonlocal c
        spec.append(("conv", filters, k, int(rng.integers(0, 2)) if bn is None else bn, act or acts[int(rng.integers(len(acts)))]))
        c = filters
        shapes.append((w, h, c))

    conv(32, 3, "leaky")
    for _ in range(int(rng.integers(3, 9))):
        kind = rng.choice(["conv3", "conv1", "max", "max1", "route1", "route2", "reorg"], p=[0.4, 0.2, 0.15, 0.05, 0.05, 0.1, 0.05])
        if kind == "conv3":
            conv(int(rng.choice([32, 64, 96, 128, 160])), 3)
        elif kind == "conv1":
            conv(int(rng.choice([32, 64, 128])), 1)
        elif kind == "max" and min(w, h) >= 8 and w % 2 == 0 and h % 2 == 0:
            spec.append(("max", 2, 2)); w, h = w // 2, h // 2; shapes.append((w, h, c))
        elif kind == "max1":
            spec.append(("max", 2, 1)); shapes.append((w, h, c))
        elif kind == "route1" and len(shapes) >= 2:
            j = int(rng.integers(0, len(shapes) - 1))
            spec.append(("route", [j])); w, h, c = shapes[j]; shapes.append((w, h, c))
        elif kind == "route2":
            same = [j for j, s in enumerate(shapes[:-1]) if s[:2] == (w, h)]
            if same:
                j = int(rng.choice(same))
                spec.append(("route", [-1, j])); c = c + shapes[j][2]; shapes.append((w, h, c))
        elif kind == "reorg" and w % 2 == 0 and h % 2 == 0 and min(w, h) >= 8:
            spec.append(("reorg", 2)); w, h, c = w // 2, h // 2, c * 4; shapes.append((w, h, c))
    if c % 32:
        conv(64, 1, "leaky", 1)
    conv(64, 3, "leaky", 1)
    conv(30, 1, "linear", 0)
    spec.append(("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}))
    return spec


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("Y2_FUZZ_SEEDS", "16")))))
def test_random_networks_fp16_mode(oracle, workdir, seed):
    """the same idea in fp16 storage mode (no reference counterpart: the bar is the fp32 oracle within half-precision
    noise): the weights-stationary 32-channel kernel, generic fp16 tiles, fused pools, half routes / reorg, fp32 head"""
    rng = np.random.default_rng(5000 + seed)
    size = int(rng.choice([32, 48, 64]))
    batch = int(rng.choice([1, 2, 3]))
    spec = random_spec_f16(rng, size)
    cfg = os.path.join(workdir, "fuzz16_%d.cfg" % seed)
    open(cfg, "w").write(zoo.cfg_text("fuzz16", size, size, batch, spec=spec))
    wts = os.path.join(workdir, "fuzz16_%d.weights" % seed)
    synth.write_weights(wts, zoo.resolve(spec, size), seed, 2.0)
    x = synth.image_batch(batch, 3, size, size, seed=seed + 3)
    on = oracle.OracleNet(cfg, wts)
    ref = on.predict(x)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(True)
    out = net.network_predict(x)
    kernels = [net.layer_kernel(i) for i in range(net.n)]
    assert not any(k.startswith("conv_direct") for k in kernels), kernels
    # region output: raw box terms scale with the activations, objectness / class scores live in [0, 1]
    scale = max(1.0, float(np.abs(ref).max()))
    assert out.shape == ref.shape and np.abs(out - ref).max() < 3e-2 * scale, (float(np.abs(out - ref).max()), scale, spec, kernels)
    net.set_fusion(False)
    out2 = net.network_predict(x)
    assert np.array_equal(out, out2), (spec, kernels)            # pooling in the epilogue == pooling afterwards, bit for bit
    net.free()
    on.close()
