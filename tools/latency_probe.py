#!/usr/bin/env python3
"""Batch-1 latency of the caller-facing entry points (the Kinect application's mode: one frame at a time).
usage: latency_probe.py [net=tiny-yolo-voc] [size=416] [iters=50]
Prints median wall-clock ms per call of: network_predict (host floats in, host tensor out), test_detector_img with a
network-sized frame, test_detector_img with a 640x480 4-plane frame (device resize), y2_detect_u8 with a 640x480
BGRA byte frame, y2_forward_device + y2_detect_resident (frame already in HBM)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sr_object_detection_amd import darknet, synth, zoo  # noqa: E402


def med_ms(fn, iters):
    fn(); fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny-yolo-voc"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    tmp = tempfile.mkdtemp()
    cfg = os.path.join(tmp, "n.cfg")
    open(cfg, "w").write(zoo.cfg_text(name, size, size, 1))
    wts = os.path.join(tmp, "n.weights")
    synth.write_weights(wts, zoo.resolve(name, size), 7)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    x = synth.image_batch(1, 3, size, size)
    cam = synth.image_batch(1, 4, 480, 640, seed=3)[0]
    cam_u8 = np.ascontiguousarray((cam.transpose(1, 2, 0) * 255).astype(np.uint8))[None]   # C-contiguous, like a capture buffer
    import torch
    d_x = torch.from_numpy(x).cuda()
    res = {
        "network_predict": med_ms(lambda: net.network_predict(x), iters),
        "test_detector_img(net-sized)": med_ms(lambda: net.test_detector_img(x[0], 0.24), iters),
        "test_detector_img(640x480x4)": med_ms(lambda: net.test_detector_img(cam, 0.24), iters),
        "y2_detect_u8(640x480 BGRA)": med_ms(lambda: net.detect_u8(cam_u8, 0.24, 0.1), iters),
        "forward_device+detect_resident": med_ms(lambda: (net.forward_device(d_x.data_ptr()), net.detect_resident(0.24, 0.1)), iters),
    }
    net.set_graph(True)
    res_g = {
        "network_predict": med_ms(lambda: net.network_predict(x), iters),
        "test_detector_img(net-sized)": med_ms(lambda: net.test_detector_img(x[0], 0.24), iters),
        "test_detector_img(640x480x4)": med_ms(lambda: net.test_detector_img(cam, 0.24), iters),
        "y2_detect_u8(640x480 BGRA)": med_ms(lambda: net.detect_u8(cam_u8, 0.24, 0.1), iters),
        "forward_device+detect_resident": med_ms(lambda: (net.forward_device(d_x.data_ptr()), net.detect_resident(0.24, 0.1)), iters),
    }
    print("%s %dx%d batch 1 (median of %d calls, ms; second column: forward replayed from a hipGraph):" % (name, size, size, iters))
    for k, v in res.items():
        print("  %-34s %8.3f %8.3f" % (k, v, res_g[k]))


if __name__ == "__main__":
    main()
