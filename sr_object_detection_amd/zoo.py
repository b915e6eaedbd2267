"""Network definitions for the BASELINE.json configurations, as Darknet cfg text.

The engine parses ordinary Darknet .cfg files (the grammar of
src_yolo2/parser.c:702-735).  The reference's cfg/*.cfg files do not travel to
the GPU box, so the benchmark/test networks are emitted here from compact
specs; tests/test_capi_host.py checks, when /root/reference is present,
that each emitted text parses to the same layer table as the reference's own
cfg file (cfg/yolo.cfg, cfg/tiny-yolo-voc.cfg, cfg/yolo9000.cfg,
cfg/darknet19_448.cfg).

`resolve()` is a small shape-inference pass over a spec (the rules of
parser.c:139-170, :359-374, :343-357, :450-489, :236-285) used only to size
synthetic weights and to report FLOPs; the engine does its own parse in C.
"""
from __future__ import annotations

# spec entries:
#   ("conv", filters, size, bn, activation[, stride[, pad flag[, xnor]]])      default stride 1, pad=1 (= size/2 pixels)
#   ("max", size, stride[, padding]) | ("route", [idx...]) | ("reorg", stride)
#   ("region", dict) | ("avg",) | ("softmax",) | ("cost",)
#   ("crop", width, height, noadjust) | ("batchnorm",) | ("local", filters, size, stride, pad, activation)

_D19_TRUNK = [
    ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 64, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 128, 3, 1, "leaky"), ("conv", 64, 1, 1, "leaky"), ("conv", 128, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 256, 3, 1, "leaky"), ("conv", 128, 1, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 512, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 1024, 3, 1, "leaky"), ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
    ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
]

COCO_ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]
VOC_TINY_ANCHORS = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]
Y9K_ANCHORS = [0.77871, 1.14074, 3.00525, 4.31277, 9.22725, 9.61974]

SPECS = {
    # cfg/yolo.cfg (YOLOv2-COCO): Darknet-19 trunk + passthrough (route/reorg) head
    "yolo": _D19_TRUNK + [
        ("conv", 1024, 3, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
        ("route", [-9]), ("conv", 64, 1, 1, "leaky"), ("reorg", 2), ("route", [-1, -4]),
        ("conv", 1024, 3, 1, "leaky"), ("conv", 425, 1, 0, "linear"),
        ("region", {"classes": 80, "num": 5, "anchors": COCO_ANCHORS}),
    ],
    # cfg/tiny-yolo-voc.cfg
    "tiny-yolo-voc": [
        ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
        ("conv", 64, 3, 1, "leaky"), ("max", 2, 2), ("conv", 128, 3, 1, "leaky"), ("max", 2, 2),
        ("conv", 256, 3, 1, "leaky"), ("max", 2, 2), ("conv", 512, 3, 1, "leaky"), ("max", 2, 1),
        ("conv", 1024, 3, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"), ("conv", 125, 1, 0, "linear"),
        ("region", {"classes": 20, "num": 5, "anchors": VOC_TINY_ANCHORS}),
    ],
    # cfg/yolo9000.cfg: trunk + one 1x1 conv with 3*(9418+5) filters, tree softmax head
    "yolo9000": _D19_TRUNK + [
        ("conv", 28269, 1, 0, "linear"),
        ("region", {"classes": 9418, "num": 3, "anchors": Y9K_ANCHORS, "tree": True, "map": True}),
    ],
    # cfg/darknet19_448.cfg classifier
    "darknet19": _D19_TRUNK + [("conv", 1000, 1, 0, "linear"), ("avg",), ("softmax",), ("cost",)],
    # small all-layer-types net for fast tests (not a reference cfg): 3x3 + 1x1 convs, 2/2 and 2/1 maxpools,
    # single- and multi-input routes, reorg, linear head, region
    "mini": [
        ("conv", 8, 3, 1, "leaky"), ("max", 2, 2), ("conv", 16, 3, 1, "leaky"), ("conv", 8, 1, 1, "leaky"),
        ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 1),
        ("conv", 32, 3, 1, "leaky"), ("route", [-5]), ("conv", 4, 1, 1, "leaky"), ("reorg", 2),
        ("route", [-1, -4]), ("conv", 32, 3, 1, "leaky"), ("conv", 30, 1, 0, "linear"),
        ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}),
    ],
}

# same topology with channel counts the matrix-core kernels take (Cin % 16 == 0 after the first layer):
# exercises BK=16 and BK=32 slices, 128/64/32-filter tiles, a placed (zero-copy) concat and the edge tiles
SPECS["mini-mfma"] = [
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("conv", 16, 1, 1, "leaky"),
    ("conv", 64, 3, 1, "leaky"), ("max", 2, 2), ("conv", 128, 3, 1, "leaky"), ("max", 2, 1),
    ("conv", 160, 3, 1, "leaky"), ("route", [-5]), ("conv", 16, 1, 1, "leaky"), ("reorg", 2),
    ("route", [-1, -4]), ("conv", 96, 3, 1, "leaky"), ("conv", 30, 1, 0, "linear"),
    ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}),
]

# residual blocks (shortcut_layer.c): an identity shortcut, a strided conv, and a shortcut whose
# source is twice as large and half as deep (the stride / min-channel branch of blas.c:57-81 shortcut_cpu)
SPECS["mini-res"] = [
    ("conv", 16, 3, 1, "leaky"), ("conv", 16, 1, 1, "leaky"), ("conv", 16, 3, 1, "linear"), ("shortcut", -3, "leaky"),
    ("conv", 32, 3, 1, "leaky", 2), ("conv", 32, 1, 1, "linear"), ("shortcut", -3, "linear"),
    ("conv", 32, 3, 1, "leaky"), ("shortcut", -2, "leaky"), ("conv", 30, 1, 0, "linear"),
    ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}),
]

# YOLOv1 family: cfg/yolov1/tiny-yolo.cfg -- convolutions, one dense layer, the [detection] head
SPECS["tiny-yolo-v1"] = [
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 64, 3, 1, "leaky"), ("max", 2, 2), ("conv", 128, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 256, 3, 1, "leaky"), ("max", 2, 2), ("conv", 512, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 1024, 3, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("connected", 1470, 0, "linear"),
    ("detection", {"classes": 20, "num": 2, "side": 7, "softmax": 0, "sqrt": 1}),
]
# small version for tests: a dense layer on an image (weights re-ordered for NHWC), dropout, a dense layer with
# batch-norm on a vector, softmax classes
SPECS["mini-v1"] = [
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
    ("connected", 96, 1, "leaky"), ("dropout", 0.5), ("connected", 240, 0, "linear"),
    ("detection", {"classes": 5, "num": 2, "side": 4, "softmax": 1, "sqrt": 1}),
]

# the remaining YOLOv1-era layer types (cfg/yolov1/yolo.cfg, yolo-small.cfg, xyolo.test.cfg): [crop] (centred window,
# 2x-1), a standalone [batchnorm], a locally connected layer with and without padding, then the dense head
SPECS["mini-v1-local"] = [
    ("crop", 32, 32, 0), ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("batchnorm",), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
    ("local", 24, 3, 1, 1, "leaky"), ("local", 8, 3, 1, 0, "logistic"), ("dropout", 0.5), ("connected", 240, 0, "linear"),
    ("detection", {"classes": 5, "num": 2, "side": 4, "softmax": 1, "sqrt": 1}),
]

# ---- the reference's other cfg files (classifiers and the YOLOv1 family), restated from their structure.  tests/
# test_capi_host.py checks, where /root/reference exists, that each text parses to the same layer table as the file.
_CLS_HEAD = [("avg",), ("softmax",), ("cost",)]


def _resnet50():                      # cfg/resnet50.cfg: 7x7/2 stem, 3-4-6-3 bottleneck blocks joined by [shortcut]
    s = [("conv", 64, 7, 1, "leaky", 2), ("max", 2, 2)]
    for width, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(blocks):
            s += [("conv", width, 1, 1, "leaky"), ("conv", width, 3, 1, "leaky", stride if b == 0 else 1),
                  ("conv", 4 * width, 1, 1, "linear"), ("shortcut", -4, "leaky")]
    return s + [("conv", 1000, 1, 0, "linear")] + _CLS_HEAD


def _densenet201():                   # cfg/densenet201.cfg: dense blocks of 6-12-48-32 (1x1 128, 3x3 32, route -1,-3)
    s = [("conv", 64, 7, 1, "leaky", 2), ("max", 2, 2)]
    for blocks, trans in ((6, 128), (12, 256), (48, 512), (32, None)):
        for _ in range(blocks):
            s += [("conv", 128, 1, 1, "leaky"), ("conv", 32, 3, 1, "leaky"), ("route", [-1, -3])]
        if trans:
            s += [("conv", trans, 1, 1, "leaky"), ("max", 2, 2)]
    return s + [("conv", 1000, 1, 0, "linear")] + _CLS_HEAD


_EXTRACTION_TRUNK = [
    ("conv", 64, 7, 1, "leaky", 2), ("max", 2, 2), ("conv", 192, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 128, 1, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"), ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
]
SPECS["resnet50"] = _resnet50()
SPECS["densenet201"] = _densenet201()
SPECS["extraction"] = _EXTRACTION_TRUNK + [("conv", 1000, 1, 0, "leaky")] + _CLS_HEAD          # cfg/extraction.cfg
SPECS["darknet-ref"] = [                                                                         # cfg/darknet.cfg
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2), ("conv", 64, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 128, 3, 1, "leaky"), ("max", 2, 2), ("conv", 256, 3, 1, "leaky"), ("max", 2, 2), ("conv", 512, 3, 1, "leaky"),
    ("max", 2, 2, 1), ("conv", 1024, 3, 1, "leaky"), ("conv", 1000, 1, 0, "leaky")] + _CLS_HEAD
SPECS["tiny"] = [                                                                                # cfg/tiny.cfg
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("conv", 32, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 16, 1, 1, "leaky"), ("conv", 128, 3, 1, "leaky"), ("conv", 16, 1, 1, "leaky"), ("conv", 128, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 32, 1, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("conv", 32, 1, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 64, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("conv", 64, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 128, 1, 1, "leaky"), ("conv", 1000, 1, 0, "linear")] + _CLS_HEAD
SPECS["alexnet"] = [                                                                             # cfg/alexnet.cfg (227x227)
    ("conv", 96, 11, 0, "relu", 4, 0), ("max", 3, 2, 0), ("conv", 256, 5, 0, "relu"), ("max", 3, 2, 0),
    ("conv", 384, 3, 0, "relu"), ("conv", 384, 3, 0, "relu"), ("conv", 256, 3, 0, "relu"), ("max", 3, 2, 0),
    ("connected", 4096, 0, "relu"), ("dropout", 0.5), ("connected", 4096, 0, "relu"), ("dropout", 0.5),
    ("connected", 1000, 0, "linear"), ("softmax",), ("cost",)]
_VGG_CONV = ([("conv", 64, 3, 0, "relu")] * 2 + [("max", 2, 2)] + [("conv", 128, 3, 0, "relu")] * 2 + [("max", 2, 2)] +
             [("conv", 256, 3, 0, "relu")] * 3 + [("max", 2, 2)] + [("conv", 512, 3, 0, "relu")] * 3 + [("max", 2, 2)] +
             [("conv", 512, 3, 0, "relu")] * 3 + [("max", 2, 2)])
SPECS["vgg-16"] = [("crop", 224, 224, 0)] + _VGG_CONV + [                                        # cfg/vgg-16.cfg (256x256 in)
    ("connected", 4096, 0, "relu"), ("dropout", 0.5), ("connected", 4096, 0, "relu"), ("dropout", 0.5),
    ("connected", 1000, 0, "linear"), ("softmax",), ("cost",)]
SPECS["strided"] = [("crop", 224, 224, 0), ("conv", 64, 7, 0, "ramp", 2), ("conv", 192, 3, 0, "ramp", 2),      # cfg/strided.cfg
                    ("conv", 128, 1, 0, "ramp"), ("conv", 256, 3, 0, "ramp", 2), ("conv", 128, 1, 0, "ramp"), ("conv", 256, 3, 0, "ramp"),
                    ("conv", 128, 1, 0, "ramp"), ("conv", 512, 3, 0, "ramp", 2)] + \
                   [("conv", 256, 1, 0, "ramp"), ("conv", 512, 3, 0, "ramp")] * 4 + \
                   [("conv", 256, 1, 0, "ramp"), ("conv", 1024, 3, 0, "ramp", 2), ("conv", 512, 1, 0, "ramp"), ("conv", 1024, 3, 0, "ramp"),
                    ("max", 3, 2), ("connected", 4096, 0, "ramp"), ("dropout", 0.5), ("connected", 1000, 0, "ramp"), ("softmax",), ("cost",)]
# cfg/yolov1/yolo-small.cfg: [crop] in front, no batch-norm, three dense layers
SPECS["yolo-v1-small"] = [("crop", 448, 448, 0)] + [(e[0], e[1], e[2], 0) + tuple(e[4:]) if e[0] == "conv" else e for e in _EXTRACTION_TRUNK] + [
    ("conv", 1024, 3, 0, "leaky"), ("conv", 1024, 3, 0, "leaky", 2), ("conv", 1024, 3, 0, "leaky"), ("conv", 1024, 3, 0, "leaky"),
    ("connected", 512, 0, "leaky"), ("connected", 4096, 0, "leaky"), ("dropout", 0.5), ("connected", 1470, 0, "linear"),
    ("detection", {"classes": 20, "num": 2, "side": 7, "softmax": 0, "sqrt": 1})]

# cfg/yolov1/xyolo.test.cfg in small: xnor=1 convolutions (binarized weights and inputs, convolutional_layer.c:443-447)
# behind standalone [batchnorm] layers
SPECS["mini-xnor"] = [
    ("conv", 16, 3, 1, "leaky"), ("max", 2, 2), ("batchnorm",), ("conv", 32, 3, 1, "leaky", 1, 1, 1), ("max", 2, 2), ("batchnorm",),
    ("conv", 64, 3, 1, "leaky", 1, 1, 1), ("batchnorm",), ("conv", 32, 1, 1, "leaky", 1, 1, 1), ("conv", 30, 1, 0, "linear"),
    ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}),
]

# the shapes of the reference's classifier cfgs in one small net: 7x7/2 stem (resnet50 / extraction), 3x3/2 maxpool without
# padding (alexnet), 5x5 convolution (alexnet), 2x2/2 maxpool with padding=1 (darknet.cfg), strided 3x3 and 1x1 convolutions
# (resnet50 / strided.cfg), a residual shortcut across a stride, dense head with relu, softmax
SPECS["mini-cls"] = [
    ("conv", 16, 7, 1, "leaky", 2), ("max", 3, 2, 0), ("conv", 32, 5, 0, "relu"), ("max", 2, 2, 1),
    ("conv", 32, 3, 1, "leaky", 2), ("conv", 64, 1, 1, "linear", 2), ("conv", 64, 3, 1, "leaky"), ("shortcut", -2, "leaky"),
    ("connected", 48, 0, "relu"), ("dropout", 0.5), ("connected", 10, 0, "linear"), ("softmax",), ("cost",),
]

# every activation of activations.h:21-54 outside the four the target cfgs use (strided.cfg is all `ramp`): on
# matrix-core convolutions, a strided one, a shortcut, and behind a placed (zero-copy) route source
SPECS["mini-acts"] = [
    ("conv", 16, 3, 1, "loggy"), ("conv", 16, 3, 1, "relie"), ("conv", 32, 3, 1, "ramp", 2), ("conv", 16, 1, 1, "tanh"),
    ("conv", 32, 3, 0, "plse"), ("conv", 32, 3, 1, "elu"), ("shortcut", -2, "stair"), ("conv", 16, 1, 1, "hardtan"),
    ("route", [-1, -3]), ("conv", 32, 3, 1, "lhtan"), ("conv", 30, 1, 0, "linear"),
    ("region", {"classes": 5, "num": 3, "anchors": [1.0, 1.2, 2.5, 2.0, 4.0, 3.5]}),
]

# cfg/yolov1/yolo.cfg: the full YOLOv1 -- 7x7/2 stem, 24 convolutions, a 3x3/2 convolution, a locally connected layer
# (49 locations x 256 filters x 9216 taps = 462 MB of weights), dropout, one dense layer, the [detection] head
SPECS["yolo-v1"] = [
    ("conv", 64, 7, 1, "leaky", 2), ("max", 2, 2), ("conv", 192, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 128, 1, 1, "leaky"), ("conv", 256, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"), ("conv", 256, 1, 1, "leaky"), ("conv", 512, 3, 1, "leaky"),
    ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"), ("max", 2, 2),
    ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"), ("conv", 512, 1, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
    ("conv", 1024, 3, 1, "leaky"), ("conv", 1024, 3, 1, "leaky", 2), ("conv", 1024, 3, 1, "leaky"), ("conv", 1024, 3, 1, "leaky"),
    ("local", 256, 3, 1, 1, "leaky"), ("dropout", 0.5), ("connected", 1715, 0, "linear"),
    ("detection", {"classes": 20, "num": 3, "side": 7, "softmax": 0, "sqrt": 1}),
]

DEFAULT_SIZE = {"mini-cls": 75, "mini-xnor": 32, "resnet50": 256, "densenet201": 256, "extraction": 224, "darknet-ref": 224, "tiny": 224, "alexnet": 227, "vgg-16": 256,
                "strided": 256, "yolo-v1-small": 448, "mini-acts": 32, "yolo-v1": 448, "mini-v1-local": 40, "yolo": 416, "tiny-yolo-voc": 416, "yolo9000": 544, "darknet19": 448, "mini": 32, "mini-mfma": 64, "mini-res": 32, "tiny-yolo-v1": 448, "mini-v1": 32}


def cfg_text(name: str, width: int | None = None, height: int | None = None, batch: int = 1,
             tree_path: str | None = None, map_path: str | None = None, spec=None) -> str:
    """Darknet cfg text for one of SPECS (or an explicit spec list)."""
    spec = SPECS[name] if spec is None else spec
    width = width or DEFAULT_SIZE.get(name, 416)
    height = height or width
    out = ["[net]", "batch=%d" % batch, "subdivisions=1", "width=%d" % width, "height=%d" % height, "channels=3", ""]
    for e in spec:
        kind = e[0]
        if kind == "conv":
            _, filters, size, bn, act = e[:5]
            stride = e[5] if len(e) > 5 else 1
            out += ["[convolutional]", "filters=%d" % filters, "size=%d" % size, "stride=%d" % stride, "pad=%d" % (e[6] if len(e) > 6 else 1)]
            if bn:
                out.append("batch_normalize=1")
            if len(e) > 7 and e[7]:
                out.append("xnor=1")
            out += ["activation=%s" % act, ""]
        elif kind == "max":
            out += ["[maxpool]", "size=%d" % e[1], "stride=%d" % e[2]] + (["padding=%d" % e[3]] if len(e) > 3 else []) + [""]
        elif kind == "route":
            out += ["[route]", "layers=" + ",".join(str(i) for i in e[1]), ""]
        elif kind == "reorg":
            out += ["[reorg]", "stride=%d" % e[1], ""]
        elif kind == "region":
            r = e[1]
            out += ["[region]", "anchors = " + ", ".join(repr(a) for a in r["anchors"]),
                    "classes=%d" % r["classes"], "coords=4", "num=%d" % r["num"], "softmax=1"]
            if r.get("tree"):
                if not tree_path:
                    raise ValueError("%s needs tree_path" % name)
                out.append("tree=%s" % tree_path)
            if r.get("map") and map_path:
                out.append("map=%s" % map_path)
            out.append("")
        elif kind == "shortcut":
            out += ["[shortcut]", "from=%d" % e[1], "activation=%s" % e[2], ""]
        elif kind == "connected":
            out += ["[connected]", "output=%d" % e[1]] + (["batch_normalize=1"] if e[2] else []) + ["activation=%s" % e[3], ""]
        elif kind == "dropout":
            out += ["[dropout]", "probability=%g" % e[1], ""]
        elif kind == "detection":
            d = e[1]
            out += ["[detection]", "classes=%d" % d["classes"], "coords=4", "rescore=1", "side=%d" % d["side"],
                    "num=%d" % d["num"], "softmax=%d" % d.get("softmax", 0), "sqrt=%d" % d.get("sqrt", 1), "jitter=.2", ""]
        elif kind == "crop":
            out += ["[crop]", "crop_width=%d" % e[1], "crop_height=%d" % e[2], "flip=0", "angle=0", "saturation=1", "exposure=1",
                    "noadjust=%d" % e[3], ""]
        elif kind == "batchnorm":
            out += ["[batchnorm]", ""]
        elif kind == "local":
            out += ["[local]", "filters=%d" % e[1], "size=%d" % e[2], "stride=%d" % e[3], "pad=%d" % e[4], "activation=%s" % e[5], ""]
        elif kind == "avg":
            out += ["[avgpool]", ""]
        elif kind == "softmax":
            out += ["[softmax]", "groups=1", ""]
        elif kind == "cost":
            out += ["[cost]", "type=sse", ""]
        else:
            raise ValueError(kind)
    return "\n".join(out)


def resolve(name_or_spec, width: int, height: int | None = None, channels: int = 3):
    """Layer table with shapes: list of dicts (type, w,h,c, out_w,out_h,out_c, outputs, ...)."""
    spec = SPECS[name_or_spec] if isinstance(name_or_spec, str) else name_or_spec
    h = height or width
    w, c = width, channels
    inputs = w * h * c
    layers = []
    for i, e in enumerate(spec):
        kind = e[0]
        L = {"w": w, "h": h, "c": c, "inputs": inputs}
        if kind == "conv":
            _, filters, size, bn, act = e[:5]
            stride = e[5] if len(e) > 5 else 1
            pad = size // 2 if (e[6] if len(e) > 6 else 1) else 0
            L.update(type="convolutional", filters=filters, size=size, stride=stride, pad=pad, batch_normalize=bn,
                     activation=act, out_w=(w + 2 * pad - size) // stride + 1, out_h=(h + 2 * pad - size) // stride + 1,
                     out_c=filters)
        elif kind == "max":
            size, stride = e[1], e[2]
            pad = e[3] if len(e) > 3 else (size - 1) // 2
            L.update(type="maxpool", size=size, stride=stride, pad=pad,
                     out_w=(w + 2 * pad) // stride, out_h=(h + 2 * pad) // stride, out_c=c)
        elif kind == "route":
            idx = [j if j >= 0 else i + j for j in e[1]]
            first = layers[idx[0]]
            oc = sum(layers[j]["out_c"] for j in idx)
            L.update(type="route", layers=idx, out_w=first["out_w"], out_h=first["out_h"], out_c=oc)
        elif kind == "reorg":
            s = e[1]
            L.update(type="reorg", stride=s, out_w=w // s, out_h=h // s, out_c=c * s * s)
        elif kind == "region":
            r = e[1]
            L.update(type="region", classes=r["classes"], coords=4, num=r["num"], anchors=r["anchors"],
                     out_w=0, out_h=0, out_c=0, outputs=w * h * r["num"] * (r["classes"] + 5))
        elif kind == "shortcut":
            L.update(type="shortcut", index=i + e[1] if e[1] < 0 else e[1], activation=e[2], out_w=w, out_h=h, out_c=c)
        elif kind == "connected":
            L.update(type="connected", outputs=e[1], batch_normalize=e[2], activation=e[3], out_w=1, out_h=1, out_c=e[1])
        elif kind == "dropout":
            L.update(type="dropout", out_w=w, out_h=h, out_c=c, outputs=inputs)
        elif kind == "detection":
            d = e[1]
            L.update(type="detection", classes=d["classes"], num=d["num"], side=d["side"], softmax=d.get("softmax", 0),
                     sqrt=d.get("sqrt", 1), out_w=0, out_h=0, out_c=0, outputs=inputs)
        elif kind == "crop":
            L.update(type="crop", noadjust=e[3], out_w=e[1], out_h=e[2], out_c=c)
        elif kind == "batchnorm":
            L.update(type="batchnorm", out_w=w, out_h=h, out_c=c)
        elif kind == "local":
            _, filters, size, stride, pad, act = e
            L.update(type="local", filters=filters, size=size, stride=stride, pad=pad, activation=act,
                     out_w=(w - (1 if pad else size)) // stride + 1, out_h=(h - (1 if pad else size)) // stride + 1, out_c=filters)
        elif kind == "avg":
            L.update(type="avgpool", out_w=1, out_h=1, out_c=c)
        elif kind in ("softmax", "cost"):
            L.update(type=kind, out_w=0, out_h=0, out_c=0, outputs=inputs)
        L.setdefault("outputs", L["out_w"] * L["out_h"] * L["out_c"])
        layers.append(L)
        w, h, c, inputs = L["out_w"], L["out_h"], L["out_c"], L["outputs"]
    return layers


def conv_flops(layers) -> float:
    """The reference's own FLOP count, 2*M*N*K summed over convs (src_yolo2/darknet.c:115-131)."""
    return float(sum(2.0 * l["filters"] * l["size"] ** 2 * l["c"] * l["out_h"] * l["out_w"]
                     for l in layers if l["type"] == "convolutional"))
