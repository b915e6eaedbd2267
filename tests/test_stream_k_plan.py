"""Host arithmetic of the stream-K plan (y2h_p8_stream_k_plan, y2_conv_f16.hip) -- no GPU needed.

The kernel and the fix-up launch both derive their pieces from share boundaries floor(w * I / G); this test restates that
rule and checks, for the plans the host really makes, that the pieces partition the tail tiles' K loops exactly once, that
no share exceeds one tile (at most two pieces per workgroup: the slot scheme 2*wg + piece relies on it) and that the
BASELINE fp16 shapes (darknet19_448 b128: 392 / 784 / 1568 tiles on 256 workgroups) are split as DESIGN.md says."""
import ctypes as C

import pytest

from sr_object_detection_amd import darknet


def _plan(ntiles, nk, grid):
    t, w = C.c_int(0), C.c_int(0)
    r = darknet.lib().y2h_p8_stream_k_plan(ntiles, nk, grid, C.byref(t), C.byref(w))
    return r, t.value, w.value


def _pieces(sk_tiles, sk_wgs, nk):
    """(wg, piece, tile, kb, ke) as conv_p8_f16_kernel derives them"""
    total = sk_tiles * nk
    out = []
    for w in range(sk_wgs):
        lo, hi = w * total // sk_wgs, (w + 1) * total // sk_wgs
        if hi <= lo:
            continue
        t0, k0, ln = lo // nk, lo % nk, hi - lo
        assert ln <= nk
        out.append((w, 0, t0, k0, min(nk, k0 + ln)))
        if k0 + ln > nk:
            out.append((w, 1, t0 + 1, 0, k0 + ln - nk))
    return out


@pytest.mark.parametrize("ntiles,nk,grid", [(392, 72, 256), (784, 36, 256), (1568, 18, 256), (392, 8, 256), (136, 72, 256),
                                            (12, 18, 7), (300, 9, 256), (257, 144, 256), (511, 4, 256), (1000, 1, 256)])
def test_plan_pieces_partition_the_tail(monkeypatch, ntiles, nk, grid):
    for k in ("Y2_SK", "Y2_SK_TILES", "Y2_SK_WGS", "Y2_SK_MARGIN", "Y2_SK_MINK"):
        monkeypatch.delenv(k, raising=False)
    r, t, w = _plan(ntiles, nk, grid)
    if not r:
        assert t == 0 and w == 0
        return
    assert 0 < t <= w <= grid and t <= ntiles
    assert t == ntiles % grid                       # the plan splits exactly the last, partial round
    cover = [[0] * nk for _ in range(t)]
    slots = set()
    for (wg, piece, tile, kb, ke) in _pieces(t, w, nk):
        assert 0 <= tile < t and 0 <= kb < ke <= nk
        assert (wg, piece) not in slots
        slots.add((wg, piece))
        for k in range(kb, ke):
            cover[tile][k] += 1
    assert all(c == 1 for row in cover for c in row)


def test_baseline_fp16_shapes_are_split(monkeypatch):
    for k in ("Y2_SK", "Y2_SK_TILES", "Y2_SK_WGS", "Y2_SK_MARGIN", "Y2_SK_MINK"):
        monkeypatch.delenv(k, raising=False)
    # darknet19_448 b128, 3x3 layers on the 256x256 tile: 28x28 256->512 (784 tiles, 36 K-tiles), 14x14 512->1024 (392, 72)
    assert _plan(784, 36, 256)[0] == 1
    assert _plan(392, 72, 256)[0] == 1
    # a full last round is left alone, and so is everything when switched off
    assert _plan(512, 36, 256) == (0, 0, 0)
    monkeypatch.setenv("Y2_SK", "0")
    assert _plan(784, 36, 256) == (0, 0, 0)


def test_forced_plan_keeps_a_share_within_one_tile(monkeypatch):
    monkeypatch.setenv("Y2_SK_TILES", "9")
    monkeypatch.setenv("Y2_SK_WGS", "4")            # fewer workgroups than tiles: raised to the tile count
    assert _plan(12, 18, 16) == (1, 9, 9)
    monkeypatch.setenv("Y2_SK_WGS", "40")           # more than the grid: clamped
    assert _plan(12, 18, 16) == (1, 9, 16)
    monkeypatch.setenv("Y2_SK_TILES", "30")         # more than there are tiles
    assert _plan(12, 18, 16) == (1, 12, 16)
