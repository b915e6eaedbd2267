"""CPU-only checks of the product library's host side: it loads, exports every
symbol the public headers declare, lays its structs out as the Python mirror
expects, parses cfgs to the same layer table as the oracle (and as the
reference's own cfg files), reads/writes .weights, and fails loudly -- never
falls back -- when no GPU is present.  No compute entry point is called."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sr_object_detection_amd import darknet, zoo
from tests.conftest import REFERENCE_ROOT, has_gpu
from tests.helpers import load_golden, materialize

INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?[\s\*]([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", text, flags=re.M):
        names.add(m.group(1))
    return names - {"defined", "__attribute__"}


@pytest.mark.parametrize("header", ["sr_yolo2.h", "y2_hip.h"])
def test_library_exports_every_declared_symbol(header):
    L = darknet.lib()
    names = declared_functions(header)
    assert len(names) > 20
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)
    C.c_int.in_dll(L, "gpu_index")


def test_struct_layout_matches_python_mirror():
    L = darknet.lib()
    for f in ("y2_sizeof_layer", "y2_sizeof_network", "y2_offsetof_layer_dev"):
        getattr(L, f).restype = C.c_size_t
    assert L.y2_sizeof_layer() == C.sizeof(darknet.Layer)
    assert L.y2_sizeof_network() == C.sizeof(darknet.CNetwork)
    assert L.y2_offsetof_layer_dev() == darknet.Layer.dev.offset


KEYS = ["type", "w", "h", "c", "out_w", "out_h", "out_c", "outputs", "n", "size", "stride", "pad", "batch_normalize"]


def table_of(net):
    rows = []
    for r in net.layer_table():
        if r["type"] in ("route", "region", "softmax", "cost", "avgpool"):
            r = dict(r, size=0, stride=0, pad=0)       # fields the reference leaves unset for these types
        if r["type"] == "route":
            r = dict(r, w=0, h=0, c=0)
        rows.append({k: r[k] for k in KEYS})
    return rows


@pytest.mark.parametrize("name,size,batch", [("mini", 32, 2), ("mini-mfma", 64, 2), ("mini-res", 32, 2), ("mini-v1", 32, 2), ("tiny-yolo-v1", 448, 1), ("tiny-yolo-voc", 416, 1),
                                             ("yolo", 608, 4), ("darknet19", 448, 2), ("yolo9000", 544, 1)])
def test_cfg_parse_matches_oracle_and_zoo(oracle, workdir, name, size, batch):
    cfg, _, _ = materialize(workdir, name, size, batch, 1) if name in ("mini", "mini-mfma") else (None, None, None)
    if cfg is None:
        tree = mp = None
        if name == "yolo9000":
            from sr_object_detection_amd import synth
            tree = os.path.join(workdir, "t9k.tree")
            synth.write_tree(tree, 9418)
        cfg = os.path.join(workdir, "%s_%d_b%d_parse.cfg" % (name, size, batch))
        open(cfg, "w").write(zoo.cfg_text(name, size, size, batch, tree_path=tree))
    net = darknet.Network.parse_network_cfg(cfg)
    on = oracle.OracleNet(cfg)
    assert net.n == on.n and net.batch == on.batch == batch
    resolved = zoo.resolve(name, size)
    for i, row in enumerate(table_of(net)):
        o = on.layer_info(i)
        assert row["type"] == o["type"]
        for k in ("out_w", "out_h", "out_c", "outputs"):
            assert row[k] == o[k] == resolved[i][k], (i, k)
        if row["type"] in ("convolutional", "maxpool"):
            for k in ("w", "h", "c", "size", "stride", "pad"):
                assert row[k] == o[k], (i, k)
    assert net.output_size == on.layer_info(on.last)["outputs"]
    net.free()
    on.close()


REF_CFGS = {"yolo": ("yolo.cfg", 416), "tiny-yolo-voc": ("tiny-yolo-voc.cfg", 416), "darknet19": ("darknet19_448.cfg", 448),
            # the reference's other cfg files, restated in zoo.py (classifiers, the YOLOv1 family)
            "resnet50": ("resnet50.cfg", 256), "densenet201": ("densenet201.cfg", 256), "extraction": ("extraction.cfg", 224),
            "darknet-ref": ("darknet.cfg", 224), "tiny": ("tiny.cfg", 224), "alexnet": ("alexnet.cfg", 227), "vgg-16": ("vgg-16.cfg", 256),
            "strided": ("strided.cfg", 256), "yolo-v1": ("yolov1/yolo.cfg", 448), "yolo-v1-small": ("yolov1/yolo-small.cfg", 448),
            "tiny-yolo-v1": ("yolov1/tiny-yolo.cfg", 448)}


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("name", sorted(REF_CFGS))
def test_reference_cfg_files_parse_to_the_zoo_tables(workdir, name):
    """The reference's own cfg text (training keys, comments, blanks, batch/subdivisions) is accepted
    verbatim and yields the same layers as the cfg this repo emits for that network."""
    fname, size = REF_CFGS[name]
    ref_net = darknet.Network.parse_network_cfg(os.path.join(REFERENCE_ROOT, "cfg", fname))
    ours = os.path.join(workdir, name + "_ours.cfg")
    open(ours, "w").write(zoo.cfg_text(name, size, size, ref_net.batch))
    our_net = darknet.Network.parse_network_cfg(ours)
    assert table_of(ref_net) == table_of(our_net)
    assert [(ref_net.layer(i).activation, ref_net.layer(i).noadjust) for i in range(ref_net.n)] == \
           [(our_net.layer(i).activation, our_net.layer(i).noadjust) for i in range(our_net.n)]
    if name not in ("yolo", "tiny-yolo-voc", "darknet19"):
        ref_net.free()
        our_net.free()
        return
    if name == "tiny-yolo-voc":
        assert ref_net.batch == 8          # batch=64 / subdivisions=8 (cfg/tiny-yolo-voc.cfg:2-3)
    a = np.ctypeslib.as_array(ref_net.last.biases, shape=(2 * ref_net.last.n,)) if name != "darknet19" else None
    b = np.ctypeslib.as_array(our_net.last.biases, shape=(2 * our_net.last.n,)) if name != "darknet19" else None
    if a is not None:
        assert np.array_equal(a, b)        # anchors
    ref_net.free()
    our_net.free()


def test_load_weights_fills_host_arrays_like_the_oracle(oracle, workdir):
    cfg, wts, _ = materialize(workdir, "mini-mfma", 64, 1, 3)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    on = oracle.OracleNet(cfg, wts)
    for i in range(net.n):
        l = net.layer(i)
        if darknet.LAYER_TYPES[l.type] != "CONVOLUTIONAL":
            continue
        nw = l.n * l.c * l.size * l.size
        for which, (ptr, cnt) in enumerate([(l.weights, nw), (l.biases, l.n), (l.scales, l.n),
                                            (l.rolling_mean, l.n), (l.rolling_variance, l.n)]):
            if not ptr:
                continue
            ours = np.ctypeslib.as_array(ptr, shape=(cnt,))
            theirs = np.ctypeslib.as_array(oracle.lib().orc_layer_param(on.h, i, which), shape=(cnt,))
            assert np.array_equal(ours, theirs), (i, which)
    # writer round trip (parser.c:822): the bytes we write are the bytes we read
    out = os.path.join(workdir, "roundtrip.weights")
    net.save_weights(out)
    assert open(out, "rb").read() == open(wts, "rb").read()
    net.free()
    on.close()


def test_weights_header_version_02_is_read_with_a_64bit_seen(workdir):
    from sr_object_detection_amd import synth
    cfg, _, _ = materialize(workdir, "mini", 32, 1, 5)
    layers = zoo.resolve("mini", 32)
    a = os.path.join(workdir, "v01.weights")
    b = os.path.join(workdir, "v02.weights")
    synth.write_weights(a, layers, 5, version=(0, 1, 0))
    synth.write_weights(b, layers, 5, version=(0, 2, 0))
    assert os.path.getsize(b) == os.path.getsize(a) + 4
    na = darknet.Network.parse_network_cfg(cfg); na.load_weights(a)
    nb = darknet.Network.parse_network_cfg(cfg); nb.load_weights(b)
    la, lb = na.layer(0), nb.layer(0)
    n = la.n * la.c * la.size * la.size
    assert np.array_equal(np.ctypeslib.as_array(la.weights, shape=(n,)), np.ctypeslib.as_array(lb.weights, shape=(n,)))
    na.free(); nb.free()


def test_errors_are_loud(workdir):
    with pytest.raises(darknet.Y2Error):
        darknet.Network.parse_network_cfg(os.path.join(workdir, "does_not_exist.cfg"))
    bad = os.path.join(workdir, "bad.cfg")
    open(bad, "w").write("[net]\nbatch=1\nwidth=32\nheight=32\nchannels=3\n\n[gru]\nfilters=10\n")
    with pytest.raises(darknet.Y2Error, match="outside"):
        darknet.Network.parse_network_cfg(bad)
    cfg, wts, x = materialize(workdir, "mini", 32, 1, 5)
    net = darknet.Network.parse_network_cfg(cfg)
    with pytest.raises(darknet.Y2Error):
        net.load_weights(os.path.join(workdir, "nope.weights"))
    net.free()


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_failure_not_fallback(workdir):
    """Without a HIP device every compute entry point must fail; nothing is computed on the CPU."""
    cfg, wts, x = materialize(workdir, "mini", 32, 1, 5)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    with pytest.raises(darknet.Y2Error, match="no HIP device|cannot run"):
        net.network_predict(x)
    with pytest.raises(darknet.Y2Error):
        darknet.do_nms_sort(np.zeros((4, 4), np.float32), np.ones((4, 2), np.float32), 0.4)
    with pytest.raises(darknet.Y2Error):
        darknet.resize_image(np.zeros((3, 8, 8), np.float32), 4, 4)
    net.free()


def test_gpu_index_negative_is_rejected(workdir):
    cfg, _, x = materialize(workdir, "mini", 32, 1, 5)
    net = darknet.Network.parse_network_cfg(cfg, gpu=-1)
    with pytest.raises(darknet.Y2Error, match="no CPU compute path|no HIP device|cannot run"):
        net.network_predict(x)
    net.free()


def test_denormalize_net_writes_the_reference_weight_file(workdir):
    """darknet.c:309 denormalize_net + convolutional_layer.c:321: the weight file saved after folding batch-norm
    must equal, byte for byte, the one the compiled reference saved (tests/golden/denorm_mini.npz)."""
    from sr_object_detection_amd import synth
    g = load_golden("denorm_mini")
    cfg = os.path.join(workdir, "denorm_mini.cfg")
    open(cfg, "w").write(zoo.cfg_text("mini", int(g["size"]), int(g["size"]), 1))
    wts = os.path.join(workdir, "denorm_mini_in.weights")
    synth.write_weights(wts, zoo.resolve("mini", int(g["size"])), int(g["seed"]))
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    net.denormalize()
    assert all(net.layer(i).batch_normalize == 0 for i in range(net.n))
    out = os.path.join(workdir, "denorm_mini_out.weights")
    net.save_weights(out)
    assert open(out, "rb").read() == bytes(g["weights"])
    net.free()


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="reference checkout not present (GPU box)")
def test_reference_resnet50_cfg_parses(oracle):
    """cfg/resnet50.cfg ([shortcut] + strided convolutions) is accepted verbatim; shapes agree with the oracle"""
    path = os.path.join(REFERENCE_ROOT, "cfg", "resnet50.cfg")
    net = darknet.Network.parse_network_cfg(path)
    on = oracle.OracleNet(path)
    assert net.n == on.n == 70 and net.output_size == 1000
    rows = table_of(net)
    assert sum(r["type"] == "shortcut" for r in rows) == 16
    for i, row in enumerate(rows):
        o = on.layer_info(i)
        assert row["type"] == o["type"] and row["outputs"] == o["outputs"], i
    net.free()
    on.close()
