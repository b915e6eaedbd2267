"""Synthetic, seed-reproducible inputs for tests and benchmarks.

There are no .weights files, images or label trees in the reference checkout
(SURVEY.md section 0), so every tensor used for parity and timing is generated
here from a portable counter-based PRNG (splitmix64) -- never libc rand() --
so that the CPU oracle, the compiled reference and the HIP engine all read the
same bytes.  The .weights container follows the reference's layout
(src_yolo2/parser.c:1009-1082 load_weights_upto, :963-1006
load_convolutional_weights; header written as version 0.1.0 + int32 `seen`,
the form save_weights_upto emits, parser.c:833-839).
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit outputs of splitmix64 started at `seed`, skipping `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int) -> np.ndarray:
    """float32 uniform on [0,1): the top 24 bits of each splitmix64 output."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform(seed: int, n: int, lo: float, hi: float) -> np.ndarray:
    u = uniform01(seed, n)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def image_batch(batch: int, c: int, h: int, w: int, seed: int = 0xC0FFEE) -> np.ndarray:
    """[batch][c][h][w] float32 in [0,1); image i uses seed + i."""
    out = np.empty((batch, c, h, w), dtype=np.float32)
    for i in range(batch):
        out[i] = uniform01(seed + i, c * h * w).reshape(c, h, w)
    return out


def conv_params(layers, seed: int, head_gain: float = 4.0, obj_gain: float = 2.0, obj_bias: float = -4.0):
    """Yield per-conv parameter dicts for a resolved layer table (zoo.resolve).

    Recipe: weights U(-a,a) with a = sqrt(6/K) (variance 2/K keeps leaky-ReLU
    activations O(1) through the trunk), scales in [0.8,1.2], rolling_mean in
    [-0.1,0.1], rolling_variance in [0.5,1.5], biases in [-0.1,0.1].  When the
    last conv feeds a [region] layer its class rows are scaled by `head_gain`
    (peaky softmax) and its objectness rows by `obj_gain` with bias `obj_bias`
    (few confident boxes, like a trained detector) so that detections are
    sparse and spread over (0,1) instead of hugging one value; the tx,ty,tw,th
    rows keep gain 1 so exp(tw) stays tame.
    """
    convs = [l for l in layers if l["type"] == "convolutional"]
    region = next((l for l in layers if l["type"] == "region"), None)
    for ci, l in enumerate(convs):
        n, c, k = l["filters"], l["c"], l["size"]
        K = c * k * k
        s = seed * 1000003 + ci * 7919
        p = {"biases": uniform(s + 1, n, -0.1, 0.1)}
        if l["batch_normalize"]:
            p["scales"] = uniform(s + 2, n, 0.8, 1.2)
            p["rolling_mean"] = uniform(s + 3, n, -0.1, 0.1)
            p["rolling_variance"] = uniform(s + 4, n, 0.5, 1.5)
        a = math.sqrt(6.0 / K)
        wts = uniform(s + 5, n * K, -a, a).reshape(n, K)
        if region is not None and ci == len(convs) - 1 and head_gain != 1.0:
            per = region["classes"] + region["coords"] + 1
            rows = np.arange(n)
            cls = (rows % per) > region["coords"]
            obj = (rows % per) == region["coords"]
            wts[cls] *= np.float32(head_gain)
            p["biases"][cls] *= np.float32(head_gain)
            wts[obj] *= np.float32(obj_gain)
            p["biases"][obj] = np.float32(obj_bias)
        p["weights"] = wts.reshape(-1)
        yield p


def write_weights(path: str, layers, seed: int, head_gain: float = 4.0, version=(0, 1, 0)) -> int:
    """Write a Darknet .weights file; returns the byte count."""
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", *version))
        if version[0] * 10 + version[1] >= 2:
            f.write(struct.pack("<Q", 0))
        else:
            f.write(struct.pack("<i", 0))
        convs = conv_params(layers, seed, head_gain)
        for li, l in enumerate(layers):
            if l["type"] == "convolutional":
                p = next(convs)
                f.write(p["biases"].tobytes())
                if "scales" in p:
                    f.write(p["scales"].tobytes())
                    f.write(p["rolling_mean"].tobytes())
                    f.write(p["rolling_variance"].tobytes())
                f.write(p["weights"].tobytes())
            elif l["type"] == "connected":
                # parser.c:806-820 order: biases, weights [outputs][inputs], then scales / mean / variance
                n, K = l["outputs"], l["inputs"]
                s = seed * 1000003 + 500000 + li * 7919
                head = li + 1 < len(layers) and layers[li + 1]["type"] == "detection"
                a = math.sqrt(6.0 / K) * (0.5 if head else 1.0)
                f.write((uniform(s + 1, n, 0.2, 0.8) if head else uniform(s + 1, n, -0.1, 0.1)).tobytes())
                f.write(uniform(s + 5, n * K, -a, a).tobytes())
                if l.get("batch_normalize"):
                    f.write(uniform(s + 2, n, 0.8, 1.2).tobytes())
                    f.write(uniform(s + 3, n, -0.1, 0.1).tobytes())
                    f.write(uniform(s + 4, n, 0.5, 1.5).tobytes())
            elif l["type"] == "batchnorm":
                # parser.c:794-804 order: scales, rolling_mean, rolling_variance (no biases)
                n = l["c"]
                s = seed * 1000003 + 600000 + li * 7919
                f.write(uniform(s + 2, n, 0.8, 1.2).tobytes())
                f.write(uniform(s + 3, n, -0.1, 0.1).tobytes())
                f.write(uniform(s + 4, n, 0.5, 1.5).tobytes())
            elif l["type"] == "local":
                # parser.c:865-875 order: biases [outputs], weights [locations][filters][c*size*size]
                K = l["size"] * l["size"] * l["c"]
                s = seed * 1000003 + 700000 + li * 7919
                a = math.sqrt(6.0 / K)
                f.write(uniform(s + 1, l["outputs"], -0.1, 0.1).tobytes())
                f.write(uniform(s + 5, l["out_w"] * l["out_h"] * l["filters"] * K, -a, a).tobytes())
        return f.tell()


def write_tree(path: str, n_nodes: int, n_roots: int = 10, seed: int = 9000) -> dict:
    """A valid synthetic label tree in the reference's "name parent" text form
    (src_yolo2/tree.c:53).  The checkout's cfg/9k.tree is corrupt,
    so the yolo9000 configuration runs on this one.  Nodes are laid out so that
    siblings are contiguous (the grouping rule of read_tree): roots first, then
    each parent's children in turn, fan-out 2..12 chosen by the PRNG."""
    parents = [-1] * n_roots
    fan = (splitmix64(seed, n_nodes) % np.uint64(11)).astype(np.int64) + 2
    p = 0
    while len(parents) < n_nodes:
        k = int(min(fan[p], n_nodes - len(parents)))
        parents.extend([p] * k)
        p += 1
    with open(path, "w") as f:
        for i, par in enumerate(parents):
            f.write("n%08d %d\n" % (i, par))
    return {"parents": parents}


def write_map(path: str, n_entries: int, n_nodes: int, seed: int = 9001) -> list:
    """coco9k.map-style file: one node index per line (src_yolo2/utils.c:17)."""
    idx = (splitmix64(seed, n_entries) % np.uint64(n_nodes)).astype(np.int64).tolist()
    with open(path, "w") as f:
        for v in idx:
            f.write("%d\n" % v)
    return idx
