"""Frame ingest on the device: u8 -> planes, BGR->RGB, resize / letterbox, and the fused
camera-frame entry y2_detect_u8, checked against the oracle's restatement of yolo_v2_class.hpp:94-141
and image.c:1087,1601-1645,1950-1992.

Parity note: the reference's image.c does not compile without OpenCV, so these
oracle functions are a line-by-line restatement that no run of the reference pins ("parity unpinned" for
the ingest helpers; the network/decode/NMS behind them are pinned by tests/golden)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth
from tests.helpers import load_golden, materialize

pytestmark = pytest.mark.gpu


def _frames(b, h, w, c, seed=5):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(b, h, w, c), dtype=np.uint8)


def _u8_to_planes_dev(frames, planes, swap, step=None):
    """call y2h_u8_to_planes directly (thin C-ABI layer), optionally with a padded row pitch"""
    L = darknet.lib()
    b, h, w, c = frames.shape
    step = step or w * c
    raw = np.zeros((b, h, step), np.uint8)
    raw[:, :, :w * c] = frames.reshape(b, h, w * c)
    d_src, d_dst = C.c_void_p(), C.c_void_p()
    assert L.y2h_malloc(C.byref(d_src), raw.nbytes) == 0 and L.y2h_malloc(C.byref(d_dst), b * planes * h * w * 4) == 0
    out = np.zeros((b, planes, h, w), np.float32)
    assert L.y2h_memcpy_h2d(d_src, raw.ctypes.data, raw.nbytes, None) == 0
    assert L.y2h_u8_to_planes(d_src, b, h, w, c, step, step * h, planes, int(swap), d_dst, None) == 0
    assert L.y2h_memcpy_d2h(out.ctypes.data, d_dst, out.nbytes, None) == 0
    L.y2h_free(d_src); L.y2h_free(d_dst)
    return out


@pytest.mark.parametrize("c,planes,swap,pad", [(3, 3, True, 0), (3, 3, False, 5), (4, 3, True, 8), (4, 4, True, 0), (1, 1, True, 3)])
def test_u8_to_planes_bitwise(oracle, c, planes, swap, pad):
    fr = _frames(2, 37, 53, c)
    got = _u8_to_planes_dev(fr, planes, swap, step=53 * c + pad)
    for b in range(2):
        assert np.array_equal(got[b], oracle.u8_to_planes(fr[b], planes, swap))
    # every byte value maps to the reference's double division rounded once to fp32
    allv = np.arange(256, dtype=np.uint8).reshape(1, 1, 256, 1)
    assert np.array_equal(_u8_to_planes_dev(allv, 1, False)[0, 0, 0], (np.arange(256) / 255.0).astype(np.float32))


@pytest.mark.parametrize("ih,iw,h,w", [(48, 64, 96, 96), (64, 48, 96, 96), (33, 97, 64, 128), (50, 50, 32, 32), (17, 200, 416, 416)])
def test_letterbox_image_bitwise(oracle, ih, iw, h, w):
    x = synth.image_batch(1, 3, ih, iw, seed=77)[0]
    want = oracle.letterbox_image(x, w, h)
    got = darknet.letterbox_image(x, w, h)
    assert np.array_equal(got, want)
    assert (got == .5).any() or (ih * w == iw * h)          # the bars are there unless the aspect already matches
    box = synth.image_batch(1, 3, h, w, seed=78)[0]
    assert np.array_equal(darknet.letterbox_image(x, w, h, into=box), oracle.letterbox_image(x, w, h, into=box))


def test_letterbox_errors():
    with pytest.raises(darknet.Y2Error):
        darknet.letterbox_image(np.zeros((3, 1, 900), np.float32), 4, 4)     # (ih*w)/iw == 0 -> degenerate


@pytest.mark.parametrize("letterbox,fh,fw,c", [(False, 64, 64, 3), (False, 48, 80, 4), (True, 48, 80, 3), (True, 90, 40, 4)])
def test_detect_u8_equals_float_path(oracle, workdir, letterbox, fh, fw, c):
    """bytes in -> detections out equals: oracle ingest (u8->planes, swap, resize/letterbox) + the float entry"""
    g = load_golden("mini_64_b3")
    cfg, wts, _ = materialize(workdir, "mini", 64, 3, int(g["seed"]), float(g["head_gain"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    fr = _frames(3, fh, fw, c, seed=11)
    planes = [oracle.u8_to_planes(f, 3, True) for f in fr]
    if (fh, fw) != (64, 64):
        planes = [oracle.letterbox_image(p, 64, 64) if letterbox else oracle.resize_image(p, 64, 64) for p in planes]
    x = np.stack(planes)
    thresh, nms = 0.05, 0.4
    want, wc = net.detect(x, thresh, nms)
    got, gc = net.detect_u8(fr, thresh, nms, swap_rb=True, letterbox=letterbox)
    assert np.array_equal(wc, gc) and int(gc.sum()) > 0
    for a, b in zip(want, got):
        assert a.tobytes() == b.tobytes()
    # and the network input it built is bit-identical to the oracle's
    net.ingest_u8(fr, swap_rb=True, letterbox=letterbox)
    L = darknet.lib()
    L.y2_forward_device(net.net, None)
    ref = darknet.Network.parse_network_cfg(cfg)
    ref.load_weights(wts)
    ref.network_predict(x)
    assert np.array_equal(net.pull_layer_output(net.n - 2), ref.pull_layer_output(net.n - 2))
    ref.free()
    net.free()


def test_detect_u8_rejects_bad_geometry(workdir):
    cfg, wts, _ = materialize(workdir, "mini", 32, 2, 1)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    with pytest.raises(darknet.Y2Error):
        net.detect_u8(_frames(3, 32, 32, 3), 0.2, 0.4)          # batch mismatch
    with pytest.raises(darknet.Y2Error):
        net.detect_u8(_frames(2, 32, 32, 2), 0.2, 0.4)          # fewer channels than the network reads
    net.free()


def test_graph_replay_matches_direct_launches(workdir):
    """y2_set_graph: the forward pass recorded into a hipGraph and replayed gives the same bits as launching the kernels
    one by one -- across repeated calls, a second input pointer (re-record), a batch change (new plan) and back"""
    import torch
    from sr_object_detection_amd import synth, zoo
    import os
    cfg = os.path.join(workdir, "graph.cfg")
    open(cfg, "w").write(zoo.cfg_text("mini-mfma", 64, 64, 2))
    wts = os.path.join(workdir, "graph.weights")
    synth.write_weights(wts, zoo.resolve("mini-mfma", 64), 5)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    xa, xb = synth.image_batch(2, 3, 64, 64, seed=1), synth.image_batch(2, 3, 64, 64, seed=2)
    want_a, want_b = net.network_predict(xa).copy(), net.network_predict(xb).copy()
    net.set_graph(True)
    for _ in range(3):
        assert np.array_equal(net.network_predict(xa), want_a)
        assert np.array_equal(net.network_predict(xb), want_b)          # same staging buffer: replay, new contents
    da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
    for d, want in ((da, want_a), (db, want_b), (da, want_a)):           # a different device pointer re-records
        assert np.array_equal(net.predict_device(d.data_ptr()).reshape(want.shape), want)
    net.set_batch_network(1)                                             # new plan: the graph is dropped and re-recorded
    one = net.network_predict(xa[:1])
    assert np.array_equal(one.reshape(-1), want_a.reshape(2, -1)[0])
    net.set_graph(False)
    assert np.array_equal(net.network_predict(xa[:1]), one)
    net.free()


def test_detect_enqueue_fetch_overlaps_the_next_forward(workdir):
    """y2_detect_enqueue / y2_detect_fetch: the detections of batch i fetched after batch i+1's forward pass has been
    enqueued are those of batch i (the fetch waits for its own event, the next forward does not disturb the record
    buffers), identical to the synchronous y2_detect_resident"""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    cfg = os.path.join(workdir, "pipe.cfg")
    open(cfg, "w").write(zoo.cfg_text("mini-mfma", 64, 64, 3))
    wts = os.path.join(workdir, "pipe.weights")
    synth.write_weights(wts, zoo.resolve("mini-mfma", 64), 5)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    frames = [torch.from_numpy(synth.image_batch(3, 3, 64, 64, seed=10 + i)).cuda() for i in range(4)]
    want = []
    for f in frames:
        net.forward_device(f.data_ptr())
        want.append(net.detect_resident(0.3, 0.4))
    assert sum(int(c.sum()) for _, c in want) > 10 and len({int(c.sum()) for _, c in want}) > 1
    got = []
    net.forward_device(frames[0].data_ptr())
    net.detect_enqueue(0.3, 0.4)
    for i in range(1, 4):
        net.forward_device(frames[i].data_ptr())
        got.append(net.detect_fetch())
        net.detect_enqueue(0.3, 0.4)
    got.append(net.detect_fetch())
    for (gd, gc), (wd, wc) in zip(got, want):
        assert np.array_equal(gc, wc)
        for a, b in zip(gd, wd):
            assert np.array_equal(a, b)
    with pytest.raises(darknet.Y2Error):
        net.detect_fetch()                      # nothing outstanding
    net.free()


@pytest.mark.parametrize("netname,size,batch", [("mini-mfma", 64, 3), ("yolo", 160, 4)])
def test_detect_overlap_mode_gives_the_same_detections(workdir, netname, size, batch):
    """y2_set_detect_overlap: decode / NMS / compaction of batch i on their own stream while batch i+1's forward pass
    runs -- the next forward must not overwrite the region tensor before the chain has read it (event wait in front of
    the region layer), and every batch's detections must equal the one-stream run.  Eight different batches through the
    pipelined loop the benchmark uses, then the plain synchronous call and get_region_boxes with the mode still on."""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    cfg = os.path.join(workdir, "ovl_%s.cfg" % netname)
    open(cfg, "w").write(zoo.cfg_text(netname, size, size, batch))
    wts = os.path.join(workdir, "ovl_%s.weights" % netname)
    synth.write_weights(wts, zoo.resolve(netname, size), 5 if netname == "mini-mfma" else 831)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    frames = [torch.from_numpy(synth.image_batch(batch, 3, size, size, seed=40 + i)).cuda() for i in range(8)]
    thresh = 0.3 if netname == "mini-mfma" else 0.2
    want = []
    for f in frames:
        net.forward_device(f.data_ptr())
        want.append(net.detect_resident(thresh, 0.4))
    assert sum(int(c.sum()) for _, c in want) > 10 and len({int(c.sum()) for _, c in want}) > 1
    net.set_detect_overlap(True)
    for rep in range(3):
        got = []
        net.forward_device(frames[0].data_ptr())
        net.detect_enqueue(thresh, 0.4)
        for i in range(1, 8):
            net.forward_device(frames[i].data_ptr())
            got.append(net.detect_fetch())
            net.detect_enqueue(thresh, 0.4)
        got.append(net.detect_fetch())
        for (gd, gc), (wd, wc) in zip(got, want):
            assert np.array_equal(gc, wc)
            for a, b in zip(gd, wd):
                assert np.array_equal(a, b)
    # the synchronous forms with the mode on
    net.forward_device(frames[2].data_ptr())
    gd, gc = net.detect_resident(thresh, 0.4)
    assert np.array_equal(gc, want[2][1])
    net.forward_device(frames[3].data_ptr())
    net.detect_enqueue(thresh, 0.4)
    out = net.network_predict(frames[3].cpu().numpy())               # a forward + output copy while a chain is pending
    boxes, probs = net.get_region_boxes(1, 1, thresh, batch_item=0)  # waits for the pending chain before using its scratch
    gd, gc = net.detect_fetch()
    assert np.array_equal(gc, want[3][1])
    assert int((probs.max(axis=1) > thresh).sum()) >= int(gc[0])
    net.set_detect_overlap(False)
    net.forward_device(frames[5].data_ptr())
    gd, gc = net.detect_resident(thresh, 0.4)
    assert np.array_equal(gc, want[5][1])
    net.free()


def test_detect_overlap_together_with_graph_replay(workdir):
    """y2_set_detect_overlap + y2_set_graph: the wait that protects the region tensor from the next forward cannot be part
    of the captured graph; it is issued in front of every replay.  The benchmark's pipelined loop over eight batches that
    alternate between two device buffers (replay, re-record, replay ...) and then over ONE buffer whose contents change
    (pure replay) must give every batch's own detections -- also for the tree head, whose decode rewrites the region
    tensor in place on the detect stream."""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    for netname, size, batch, thresh in (("mini-mfma", 64, 3, 0.3), ("yolo", 160, 4, 0.2)):
        cfg = os.path.join(workdir, "ovg_%s.cfg" % netname)
        open(cfg, "w").write(zoo.cfg_text(netname, size, size, batch))
        wts = os.path.join(workdir, "ovg_%s.weights" % netname)
        synth.write_weights(wts, zoo.resolve(netname, size), 5 if netname == "mini-mfma" else 831)
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        host = [synth.image_batch(batch, 3, size, size, seed=70 + i) for i in range(8)]
        frames = [torch.from_numpy(h).cuda() for h in host]
        want = []
        for f in frames:
            net.forward_device(f.data_ptr())
            want.append(net.detect_resident(thresh, 0.4))
        assert len({int(c.sum()) for _, c in want}) > 1
        net.set_detect_overlap(True)
        net.set_graph(True)
        one = torch.empty_like(frames[0])
        for mode in ("two buffers", "one buffer"):
            def feed(i):
                if mode == "one buffer":
                    torch.cuda.synchronize()                 # (the test's own copy must not race the previous forward)
                    one.copy_(frames[i])
                    torch.cuda.synchronize()
                    return one.data_ptr()
                return frames[i].data_ptr()
            got = []
            net.forward_device(feed(0))
            net.detect_enqueue(thresh, 0.4)
            for i in range(1, 8):
                net.forward_device(feed(i))
                got.append(net.detect_fetch())
                net.detect_enqueue(thresh, 0.4)
            got.append(net.detect_fetch())
            for k, ((gd, gc), (wd, wc)) in enumerate(zip(got, want)):
                assert np.array_equal(gc, wc), "%s, %s: batch %d" % (netname, mode, k)
                for a, b in zip(gd, wd):
                    assert np.array_equal(a, b)
        net.free()


def test_misaligned_device_input_is_staged_not_refused(workdir):
    """a device input pointer that is not 16-byte aligned (frame slice of an odd-sized batch: 75x75x3 floats = 12 mod 16
    bytes): the fp32 first-layer kernel reads it with dword loads as it is, the fp16 one gets it through the engine's own
    aligned input slot -- same bits as the aligned call in both modes"""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    spec = [("conv", 32, 3, 1, "leaky"), ("max", 2, 2), ("conv", 64, 3, 1, "leaky"), ("conv", 30, 1, 0, "linear")]
    size = 76
    cfg = os.path.join(workdir, "misal.cfg")
    open(cfg, "w").write(zoo.cfg_text("misal", size, size, 1, spec=spec))
    wts = os.path.join(workdir, "misal.weights")
    synth.write_weights(wts, zoo.resolve(spec, size), 5)
    x = synth.image_batch(1, 3, size, size, seed=3)
    buf = torch.zeros(x.size + 8, dtype=torch.float32, device="cuda")
    for half in (False, True):
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        net.set_half(half)
        want = net.network_predict(x).copy()
        assert "nchw" in net.layer_kernel(0), net.layer_kernel(0)
        for off in (1, 3, 4):
            view = buf[off:off + x.size]
            view.copy_(torch.from_numpy(x.reshape(-1)))
            torch.cuda.synchronize()
            assert (view.data_ptr() % 16 != 0) == (off % 4 != 0)
            got = net.predict_device(view.data_ptr())
            assert np.array_equal(got.reshape(want.shape), want), "half=%s offset %d floats" % (half, off)
        net.free()


def test_output_enqueue_fetch_matches_predict_device(workdir):
    """y2_output_enqueue / y2_output_fetch (the classifier's overlapped host copy): batch i's scores fetched after batch
    i+1's forward was enqueued equal y2_network_predict_device's"""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    cfg = os.path.join(workdir, "pipe_cls.cfg")
    open(cfg, "w").write(zoo.cfg_text("darknet-ref", 64, 64, 2))
    wts = os.path.join(workdir, "pipe_cls.weights")
    synth.write_weights(wts, zoo.resolve("darknet-ref", 64), 6, 1.0)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    frames = [torch.from_numpy(synth.image_batch(2, 3, 64, 64, seed=20 + i)).cuda() for i in range(3)]
    want = [net.predict_device(f.data_ptr()) for f in frames]
    assert not np.array_equal(want[0], want[1])
    got = []
    net.forward_device(frames[0].data_ptr())
    net.output_enqueue()
    for i in range(1, 3):
        net.forward_device(frames[i].data_ptr())
        got.append(net.output_fetch())
        net.output_enqueue()
    got.append(net.output_fetch())
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    with pytest.raises(darknet.Y2Error):
        net.output_fetch()
    net.free()


@pytest.mark.parametrize("netname,size,batch,seed", [("tiny-yolo-voc", 416, 2, 21), ("yolo", 160, 4, 831), ("mini-mfma", 64, 3, 5)])
def test_three_launch_detect_chain_equals_separate_kernels(workdir, monkeypatch, netname, size, batch, seed):
    """y2h_detect_chain (decode_all + nms_sort + best_collect) against the eight separate launches it replaces
    (Y2_DETECT_SEPARATE=1): identical records and counts, frame after frame (the per-class candidate counts must come back
    to zero by themselves), with and without NMS, at two thresholds (region_layer.c:328-379, box.c:249-277,
    yolo_v2_class.cpp:221-238)"""
    import os
    import torch
    from sr_object_detection_amd import synth, zoo
    cfg = os.path.join(workdir, "chain_%s.cfg" % netname)
    open(cfg, "w").write(zoo.cfg_text(netname, size, size, batch))
    wts = os.path.join(workdir, "chain_%s.weights" % netname)
    synth.write_weights(wts, zoo.resolve(netname, size), seed)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    frames = [torch.from_numpy(synth.image_batch(batch, 3, size, size, seed=90 + i)).cuda() for i in range(4)]
    results = {}
    for mode in ("separate", "chain"):
        if mode == "separate":
            monkeypatch.setenv("Y2_DETECT_SEPARATE", "1")
        else:
            monkeypatch.delenv("Y2_DETECT_SEPARATE", raising=False)
        got = []
        for thresh, nms in ((0.2, 0.4), (0.05, 0.4), (0.2, 0.0), (0.1, 0.1)):
            for f in frames:
                net.forward_device(f.data_ptr())
                got.append(net.detect_resident(thresh, nms))
        results[mode] = got
    total = 0
    for (da, ca), (db, cb) in zip(results["separate"], results["chain"]):
        assert np.array_equal(ca, cb)
        total += int(ca.sum())
        for a, b in zip(da, db):
            assert np.array_equal(a, b)
    assert total > 50
    net.free()
