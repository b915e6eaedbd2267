#!/usr/bin/env python3
"""Probe (not the benchmark): does a second engine on its own stream fill the CUs the first leaves idle in its last tile
round?  R networks of batch B/R each (own weights, own stream), every step = forward + detect on each, against one network
of batch B.   tools/dual_stream_probe.py [net] [size] [batch] [steps]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sr_object_detection_amd import darknet, synth, zoo


def build(name, size, batch, tmp):
    cfg = bench.write_cfg(tmp, name, size, batch, fname="net_b%d.cfg" % batch)
    wts = os.path.join(tmp, "w_%s_%d.weights" % (name, size))
    if not os.path.exists(wts):
        synth.write_weights(wts, zoo.resolve(name, size), 831, 6.0)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    x = torch.from_numpy(synth.image_batch(batch, 3, size, size, seed=0xC0FFEE)).cuda()
    return net, x


def run(nets, steps):
    for net, x in nets:                      # warm-up / plan
        net.forward_device(x.data_ptr()); net.detect_resident(0.2, 0.4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for net, x in nets:
        net.forward_device(x.data_ptr()); net.detect_enqueue(0.2, 0.4)
    for _ in range(steps - 1):
        for net, x in nets:
            net.forward_device(x.data_ptr())
            net.detect_fetch()
            net.detect_enqueue(0.2, 0.4)
    for net, x in nets:
        net.detect_fetch()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolo"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 608
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    tmp = tempfile.mkdtemp(prefix="y2dual_")
    for r in (1, 2, 1, 2, 4):
        nets = [build(name, size, batch // r, tmp) for _ in range(r)]
        el = run(nets, steps)
        print("%d engine(s) x batch %d: %.1f images/s (%.3f ms per %d frames)" % (r, batch // r, steps * batch / el, el / steps * 1e3, batch), flush=True)
        for net, _ in nets:
            net.free()


if __name__ == "__main__":
    main()
