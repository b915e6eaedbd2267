"""Shared helpers: materialise a synthetic case (cfg + weights + input) on disk."""
from __future__ import annotations

import os

import numpy as np

from sr_object_detection_amd import synth, zoo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def materialize(tmp: str, net: str, size: int, batch: int, seed: int, head_gain: float = 4.0,
                use_map: bool = False, tag: str = ""):
    """Write cfg/weights (and tree/map) for a zoo net; returns (cfg, weights, x[b,3,h,w])."""
    os.makedirs(tmp, exist_ok=True)
    stem = "%s_%d_b%d_s%d%s" % (net.replace("-", "_"), size, batch, seed, tag)
    tree = mp = None
    if net == "yolo9000":
        tree = os.path.join(tmp, "syn9k.tree")
        if not os.path.exists(tree):
            synth.write_tree(tree, 9418)
        if use_map:
            mp = os.path.join(tmp, "syn9k.map")
            if not os.path.exists(mp):
                synth.write_map(mp, 200, 9418)
    cfg = os.path.join(tmp, stem + ("_map" if use_map else "") + ".cfg")
    with open(cfg, "w") as f:
        f.write(zoo.cfg_text(net, size, size, batch, tree_path=tree, map_path=mp))
    wts = os.path.join(tmp, "%s_s%d_g%g.weights" % (net.replace("-", "_"), seed, head_gain))
    if not os.path.exists(wts):
        synth.write_weights(wts, zoo.resolve(net, size), seed, head_gain)
    x = synth.image_batch(batch, 3, size, size)
    return cfg, wts, x


def load_golden(name: str):
    path = os.path.join(GOLDEN, name + ".npz")
    return dict(np.load(path, allow_pickle=False))


def dense_from_sparse(idx, val, total, classes):
    out = np.zeros((total, classes), dtype=np.float32)
    if len(idx):
        out[idx[:, 0], idx[:, 1]] = val
    return out
