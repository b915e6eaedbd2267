"""The tile plan of the fp32 matrix-core convolutions is host arithmetic (pick_variant, y2_conv.hip): it can be checked
without a GPU.  These pins are the plans the round-3 measurements were taken with (profiles/r03_notes.md section 11); a
change of the cost model that moves the headline plan shows up here first, before any GPU time is spent."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "plan_dump.py")] + [str(a) for a in args],
                         check=True, capture_output=True, text=True,
                         env={k: v for k, v in os.environ.items() if k not in ("Y2_MODEL_R1", "Y2_CONV_TILE", "Y2_CONV_KSPLIT", "Y2_CONV_GRID")}).stdout
    plan = {}
    for line in out.splitlines():
        f = line.split()
        if f and f[0].startswith("L") and f[0][1:].isdigit():
            plan[int(f[0][1:])] = [w for w in f if w.startswith("conv_")][0]
    return plan


def test_headline_plan_yolo_608_b32():
    p = _plan("yolo", 608, 32)
    big = [8, 10, 12, 14, 16, 18, 20, 22, 23, 24, 29]            # the 3x3 layers from 76x76 down: the dominant kernel
    assert all(p[i] == "conv_mfma_f32_192x256x32_k3" for i in big), p
    assert p[2] == "conv_c32_f32_16x16"                           # 304x304 32->64 + pool: weights stationary
    assert p[4] == p[6] == "conv_mfma_f32_128x128x32_k3"
    assert p[13] == p[15] == "conv_mfma_f32_192x256x32_k1"


def test_small_grid_plans_follow_the_measured_choices():
    p = _plan("yolo", 416, 8)
    assert p[8] == p[10] == "conv_mfma_f32_128x64x32_k3"         # 52x52 128->256: unsplit 128x64, not 192x256 with a K-split
    assert p[5] == "conv_mfma_f32_64x64x32_k1"                    # 104x104 1x1: 64x64, not 128x32
    assert p[12] == "conv_mfma_f32_64x64x32_k3"
    q = _plan("yolo9000", 544, 8)
    assert q[8] == q[10] == "conv_mfma_f32_128x64x32_k3"
    assert q[12] == q[14] == q[16] == "conv_mfma_f32_64x64x32_k3"  # 34x34 256->512: 64x64, not 192x256 split 2
    assert q[4] == q[6] == "conv_mfma_f32_128x128x32_k3"
