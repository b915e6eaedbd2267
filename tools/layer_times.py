#!/usr/bin/env python3
"""Per-layer device time of one zoo network (engine timing events, median over iterations).
usage: layer_times.py <net> [size] [batch] [iters=20] [top=0]      top > 0: only the `top` slowest layers"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sr_object_detection_amd import darknet, synth, zoo  # noqa: E402


def main():
    name = sys.argv[1]
    size = int(sys.argv[2]) if len(sys.argv) > 2 else zoo.DEFAULT_SIZE.get(name, 416)
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    tmp = tempfile.mkdtemp()
    cfg = os.path.join(tmp, "n.cfg")
    open(cfg, "w").write(zoo.cfg_text(name, size, size, batch))
    layers = zoo.resolve(name, size)
    wts = os.path.join(tmp, "n.weights")
    synth.write_weights(wts, layers, 7)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    x = synth.image_batch(batch, 3, size, size)
    net.set_timing(True)
    ts = []
    for _ in range(iters + 2):
        net.network_predict(x)
        ts.append(net.layer_times_ms())
    t = np.median(np.array(ts[2:]), axis=0)
    flops = zoo.conv_flops(layers) * batch
    print("%s %dx%d batch %d: %.3f ms device time per forward (median of %d) = %.0f images/s, %.1f TFLOP/s over the convolutions" %
          (name, size, size, batch, float(t.sum()), iters, batch / float(t.sum()) * 1e3, flops / float(t.sum()) / 1e9))
    order = sorted(range(len(layers)), key=lambda i: -t[i])[:top] if top else range(len(layers))
    for i in order:
        l = layers[i]
        fl = 2.0 * l["filters"] * l["size"] ** 2 * l["c"] * l["out_h"] * l["out_w"] * batch if l["type"] == "convolutional" else 0.0
        print("  %3d %-14s %-40s %8.3f ms %s" % (i, l["type"], net.layer_kernel(i), t[i], ("%7.1f TF" % (fl / t[i] / 1e9)) if fl and t[i] > 0 else ""))


if __name__ == "__main__":
    main()
