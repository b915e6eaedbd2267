#!/usr/bin/env python3
"""Time ONE convolution shape through the engine: a two-layer net [3->Cin 3x3 (first-layer kernel)] ->
[Cin->Cout kxk] at HxH, batch B; prints kernel, ms and TFLOP/s of the second layer.
usage: conv_probe.py H Cin Cout k [bn=1] [act=leaky] [batch=32] [steps=5]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sr_object_detection_amd import darknet, synth, zoo  # noqa: E402


def main():
    H, cin, cout, k = [int(v) for v in sys.argv[1:5]]
    bn = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    act = sys.argv[6] if len(sys.argv) > 6 else "leaky"
    batch = int(sys.argv[7]) if len(sys.argv) > 7 else 32
    steps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
    spec = [("conv", cin, 3, 1, "leaky"), ("conv", cout, k, bn, act)]
    tmp = tempfile.mkdtemp()
    cfg = os.path.join(tmp, "p.cfg")
    open(cfg, "w").write(zoo.cfg_text("probe", H, H, batch, spec=spec))
    wts = os.path.join(tmp, "p.weights")
    synth.write_weights(wts, zoo.resolve(spec, H), 5)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    if os.environ.get("Y2_PROBE_HALF"):
        net.set_half(True)
    x = synth.image_batch(batch, 3, H, H)
    net.network_predict(x)
    net.set_timing(True)
    ms = []
    for _ in range(steps):
        net.network_predict(x)
        ms.append(float(net.layer_times_ms()[1]))
    t = float(np.median(ms))
    fl = 2.0 * cout * k * k * cin * H * H * batch
    print("%dx%d b%d %d->%d k%d bn=%d %s: %s %.3f ms %.1f TFLOP/s" % (H, H, batch, cin, cout, k, bn, act,
                                                                  net.layer_kernel(1), t, fl / t / 1e9))


if __name__ == "__main__":
    main()
