#!/usr/bin/env python3
"""Per-layer kernel, time and TFLOP/s of one workload (HIP events around every layer).
usage: python tools/layer_profile.py [workload] [steps]      (workloads: see bench.py)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sr_object_detection_amd import darknet, synth, zoo  # noqa: E402


def main():
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "yolo608_b32"]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    name, size, batch = wl["net"], wl["size"], wl["batch"]
    tmp = tempfile.mkdtemp()
    cfg = bench.write_cfg(tmp, name, size, batch, "n.cfg")
    wts = os.path.join(tmp, "n.weights")
    synth.write_weights(wts, zoo.resolve(name, size), 31)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.set_half(bool(wl.get("half")))
    x = synth.image_batch(batch, 3, size, size)
    net.network_predict(x)
    net.set_timing(True)
    acc = np.zeros(net.n)
    for _ in range(steps):
        net.network_predict(x)
        acc += net.layer_times_ms()
    acc /= steps
    flops = bench.conv_layer_flops(net)
    print("%3s %-34s %18s %9s %8s" % ("L", "kernel", "in -> filters", "ms", "TFLOP/s"))
    for i in range(net.n):
        l = net.layer(i)
        tf = "%8.1f" % (flops[i] * batch / (acc[i] * 1e-3) / 1e12) if i in flops and acc[i] > 0 else ""
        print("%3d %-34s %4dx%-4dx%-5d->%-5d %9.3f %s" % (i, net.layer_kernel(i), l.w, l.h, l.c, l.out_c, acc[i], tf))
    print("total %.3f ms, conv %.3f ms, %.1f TFLOP/s over convs" % (
        acc.sum(), sum(acc[i] for i in flops), sum(flops.values()) * batch / (sum(acc[i] for i in flops) * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
